# SQ instruction / activity counters of the plane two-level scan (one launch, 10 M points x 4096 hypotheses),
# four separate rocprofv3 --pmc passes (counters only, with --kernel-trace); run on the GPU box from the repo
# root:  bash tools/pmc_sq.sh > gpurun_out/sq.txt
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"; do
  d=gpurun_out/pmc_$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace -d $d --output-format csv -- python3 tools/cells_once.py plane 10000000 4096 0:0:0 > /dev/null 2>&1 || exit 1
  f=$(find $d -name '*counter_collection.csv' | head -1)
  python3 - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_scan_cells' in r['Kernel_Name']:
        acc[r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in acc.items(): print(k, v)
PY
done
