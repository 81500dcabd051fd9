# Collect the per-round evidence under gpurun_out/<tag>/ (then copy what is to be judged into profiles/):
#   bash tools/collect_profiles.sh r02
# per workload: (1) the SQ / HBM counters of the scan kernels (tools/collect_counters.py: separate rocprofv3 --pmc
# passes), copied into profiles/ so that the bench lines below quote them; (2) the plain bench line (with the CPU
# leg; steps alternate over 4 streams, per-kernel figures from its single-stream pass); (3) a rocprofv3
# --kernel-trace --stats run of the bench on ONE stream (--streams 1; without the CPU leg and the extra compute()
# timings, so that every kernel has one launch shape and no other stream's kernels share the device with it), whose
# kernel averages must agree with the HIP-event times in its own bench line (<tag>_<w>_bench_under_rocprof.json) and
# with the single_stream pass of (2).
tag=${1:-rXX}
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
mkdir -p gpurun_out/$tag
for w in ${WORKLOADS:-plane sphere line dense us}; do
  timeout -k 10 500 python3 tools/collect_counters.py $w gpurun_out/$tag > gpurun_out/$tag/counters_$w.log 2>&1 && cp gpurun_out/$tag/r02_${w}_scan_counters.json profiles/
  echo "counters $w done"
  [ -n "$SKIP_BENCH" ] || timeout -k 10 400 python3 bench.py --workload $w > gpurun_out/$tag/${tag}_bench_$w.json 2> gpurun_out/$tag/bench_$w.err || exit 1
  [ -n "$SKIP_BENCH" ] || echo "bench $w done"
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/prof_$w -- python3 bench.py --workload $w --steps 5 --warmup 1 --streams 1 --no-cpu-baseline --no-end-to-end > gpurun_out/$tag/${tag}_${w}_bench_under_rocprof.json 2> gpurun_out/$tag/prof_$w.err || exit 1
  f=$(find gpurun_out/$tag/prof_$w -name '*kernel_stats.csv' | head -1)
  cp "$f" gpurun_out/$tag/${tag}_${w}_kernel_stats.csv
  rm -rf gpurun_out/$tag/prof_$w
  echo "rocprof $w done"
done
