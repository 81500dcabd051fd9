#!/bin/bash
# A/B through bench.py: prints value (4 streams), one_stream value
run() { # label, args...
  label="$1"; shift
  timeout -k 10 200 python bench.py "$@" --no-cpu-baseline --no-end-to-end --no-other-configs --repeats 1 --detail gpurun_out/ab_detail.json 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$label', 'value', d['value'], 'early', d.get('value_early_exit'), 'one_stream', d.get('one_stream'), 'ms/step', d['ms_per_step'])"
}
run "line default" --workload line --steps 10 --warmup 3
run "line pairs=1" --workload line --steps 10 --warmup 3 --option scan_pairs=1
run "us-iter persist=1 s4" --workload us --steps 8 --warmup 2 --rates full
run "us-iter persist=0 s4" --workload us --steps 8 --warmup 2 --rates full --option lm_persist=0
run "us-iter persist=1 s8" --workload us --steps 16 --warmup 2 --rates full --streams 8
run "us-iter persist=0 s8" --workload us --steps 16 --warmup 2 --rates full --streams 8 --option lm_persist=0
run "us-iter persist=1 s1" --workload us --steps 3 --warmup 1 --rates full --streams 1
run "us-iter persist=0 s1" --workload us --steps 3 --warmup 1 --rates full --streams 1 --option lm_persist=0
