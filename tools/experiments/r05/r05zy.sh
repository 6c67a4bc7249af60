#!/bin/bash
# lanes (independent sets of compute streams and a|b|c / h / sort(h) buffers) again, with the busy-mode level-1 plan
export TMPDIR=/tmp
out=gpurun_out/r05zy_lanes.txt; : > $out
run() { python bench.py --warmup 8 --no-cpu --no-counters "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'])"; }
for rep in 1 2; do
  for l in 4 6 3; do
    echo "2^22 lanes $l: $(ZKHIP_LANES=$l run --steps 30)" >> $out
    echo "2^20 lanes $l: $(ZKHIP_LANES=$l run --log2n 20 --steps 60)" >> $out
  done
done
cat $out
