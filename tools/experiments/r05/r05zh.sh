#!/bin/bash
# is there a fixed cost in the timed region (ramp, drain, a one-off stall)?  period against the number of timed steps
export TMPDIR=/tmp
out=gpurun_out/r05zh_steps_sweep.txt; : > $out
run() { python bench.py --no-cpu --no-counters "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'])"; }
for rep in 1 2; do
  for steps in 15 30 60 120; do
    echo "2^22 steps $steps: $(run --steps $steps --warmup 3)" >> $out
  done
done
for steps in 30 120 480; do echo "2^20 steps $steps: $(run --log2n 20 --steps $steps --warmup 3)" >> $out; done
cat $out
