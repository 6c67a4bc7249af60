#!/bin/bash
# proofs in flight, again, with the busy-mode lanes (in-tree build)
export TMPDIR=/tmp
out=gpurun_out/r05zq_in_flight.txt; : > $out
run() { python bench.py --warmup 5 --no-cpu --no-counters "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'])"; }
for rep in 1 2; do
  for f in 3 4 6 8; do
    echo "2^22 in flight $f: $(run --steps 32 --in-flight $f)" >> $out
    echo "2^20 in flight $f: $(run --log2n 20 --steps 64 --in-flight $f)" >> $out
  done
done
cat $out
