# per-kernel times of one plane scan: bash tools/axis_prof.sh <tag> [scan_bound]
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
tag=${1:-axis}
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag -- python3 tools/scan_once.py plane 20 ${2:-1} > gpurun_out/$tag.log 2>&1
f=$(find gpurun_out/$tag -name '*kernel_stats.csv' | head -1)
cp "$f" gpurun_out/${tag}_kernel_stats.csv && rm -rf gpurun_out/$tag
