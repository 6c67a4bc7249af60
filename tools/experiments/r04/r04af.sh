#!/bin/bash
# witnesses per submission at the small sizes (C-ABI loop, host witnesses): does a larger batch than the server's 4 pay?
export TMPDIR=/tmp
o=gpurun_out/r04af; mkdir -p $o
( for k in 14 15 16 17; do for b in 1 2 4 8; do
    python bench.py --log2n $k --steps 240 --warmup 8 --batch $b --no-cpu 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read());print('2^$k, $b per submission:', d['ms_per_step'], 'ms per proof,', round(d['value'],1), 'proofs/s; in flight', d['config']['proofs_in_flight'], '; window', d['config']['window_bits'])"
  done; done ) > $o/batch_sweep.txt 2>&1
cat $o/batch_sweep.txt
