#!/bin/bash
# proofs in flight at 2^22 and 2^20 with the final code
export TMPDIR=/tmp
o=gpurun_out/r04be; mkdir -p $o
( for k in 22 20; do for d in 4 5 6 7 8; do
    python bench.py --log2n $k --in-flight $d --steps 24 --warmup 3 --no-cpu 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read());print('2^$k, $d in flight:', d['ms_per_step'], 'ms per proof; resident', d['resident_witness']['ms_per_step'])" 2>&1 | tail -1
  done; done ) > $o/in_flight_sweep.txt 2>&1
cat $o/in_flight_sweep.txt
