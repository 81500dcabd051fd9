# kernel timeline of one bench step: bash tools/step_trace.sh <tag> <workload> <scan_bound>
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
tag=${1:-trace}
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$tag -- python3 tools/scan_once.py ${2:-plane} 6 ${3:-1} > gpurun_out/$tag.log 2>&1
f=$(find gpurun_out/$tag -name '*kernel_trace.csv' | head -1)
cp "$f" gpurun_out/${tag}_kernel_trace.csv && rm -rf gpurun_out/$tag
