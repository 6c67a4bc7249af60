#!/bin/bash
# window width at 2^22 and 2^20 with the round's final code (tables precomputed per width)
export TMPDIR=/tmp
o=gpurun_out/r04bd; mkdir -p $o
( for k in 22 20; do for c in 18 19 20 21 22; do
    python bench.py --log2n $k --window-bits $c --steps 16 --warmup 3 --no-cpu 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read());print('2^$k window $c:', d['ms_per_step'], 'ms per proof; resident', d['resident_witness']['ms_per_step'], '; sync', d['ms_per_proof_sync'], '; G1 launch per MSM alone', d['roofline']['launch_ms_one_proof_in_flight'])" 2>&1 | tail -1
  done; done ) > $o/window_sweep.txt 2>&1
cat $o/window_sweep.txt
