run() { label="$1"; shift
  timeout -k 10 200 python bench.py "$@" --no-cpu-baseline --no-end-to-end --no-other-configs --repeats 1 --detail gpurun_out/ab_detail.json 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$label', 'value', d['value'], 'one_stream', d.get('one_stream'), 'ms/step', d['ms_per_step'])"
}
run "us-iter persist=1 s4" --workload us --steps 12 --warmup 4 --rates full
run "us-iter persist=1 s8" --workload us --steps 24 --warmup 8 --rates full --streams 8
run "us-iter persist=1 s6" --workload us --steps 18 --warmup 6 --rates full --streams 6
run "us-iter persist=0 s8" --workload us --steps 24 --warmup 8 --rates full --streams 8 --option lm_persist=0
run "us-iter persist=0 s6" --workload us --steps 18 --warmup 6 --rates full --streams 6 --option lm_persist=0
