# Collect the per-round evidence under gpurun_out/<tag>/ (then copy what is to be judged into profiles/):
#   bash tools/collect_profiles.sh r04
# per workload and hypothesis rate (full_count = every hypothesis counted, scan_bound 0 -- bench.py's `value`;
# early_exit = scan_bound 1): (1) the SQ / HBM counters of the scan kernels (tools/collect_counters.py: separate
# rocprofv3 --pmc passes), copied into profiles/ so that the bench lines below quote them; (2) a rocprofv3
# --kernel-trace --stats run of the bench on ONE stream and ONE rate (--streams 1 --rates full|early; without the CPU
# leg, the legs of the other configs and the extra compute() timings, so that every kernel has one launch shape and no
# other stream's kernels share the device with it), whose kernel averages must agree with the HIP-event times in its
# own bench line (<tag>_<w>_<rate>_bench_under_rocprof.json); (3) the plain bench line of the workload (with the CPU
# leg; the headline plane line is the default run, other_configs included).
tag=${1:-rXX}
export LSQR_ROUND=$tag
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
mkdir -p gpurun_out/$tag
for w in ${WORKLOADS:-plane sphere line dense us}; do
  for rate in ${RATES:-full_count early_exit}; do
    short=full; [ $rate = early_exit ] && short=early
    timeout -k 10 500 python3 tools/collect_counters.py $w gpurun_out/$tag $rate > gpurun_out/$tag/counters_${w}_$rate.log 2>&1 && cp gpurun_out/$tag/${tag}_${w}_${rate}_scan_counters.json profiles/
    echo "counters $w $rate done"
    extra=""; [ $w = us ] && extra="--us-fit analytic"
    timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/prof_${w}_$rate -- python3 bench.py --workload $w $extra --steps 5 --warmup 1 --repeats 1 --streams 1 --rates $short --no-cpu-baseline --no-end-to-end --no-other-configs --detail gpurun_out/$tag/${tag}_${w}_${rate}_bench_under_rocprof_detail.json > gpurun_out/$tag/${tag}_${w}_${rate}_bench_under_rocprof.json 2> gpurun_out/$tag/prof_${w}_$rate.err || exit 1
    f=$(find gpurun_out/$tag/prof_${w}_$rate -name '*kernel_stats.csv' | head -1)
    cp "$f" gpurun_out/$tag/${tag}_${w}_${rate}_kernel_stats.csv
    rm -rf gpurun_out/$tag/prof_${w}_$rate
    echo "rocprof $w $rate done"
  done
  if [ -z "$SKIP_BENCH" ]; then
    if [ $w = plane ]; then
      timeout -k 10 600 python3 bench.py --detail gpurun_out/$tag/${tag}_bench_plane_detail.json > gpurun_out/$tag/${tag}_bench_plane.json 2> gpurun_out/$tag/bench_plane.err || exit 1
    else
      timeout -k 10 400 python3 bench.py --workload $w --detail gpurun_out/$tag/${tag}_bench_${w}_detail.json > gpurun_out/$tag/${tag}_bench_$w.json 2> gpurun_out/$tag/bench_$w.err || exit 1
    fi
    echo "bench $w done"
  fi
done
