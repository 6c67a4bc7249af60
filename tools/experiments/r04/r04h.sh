#!/bin/bash
export TMPDIR=/tmp
o=gpurun_out/r04h; mkdir -p $o
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1: host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'one at a time resident', d['latency_ms_one_at_a_time']['witness_in_hbm'], 'sync', d['ms_per_proof_sync'])"; }
( for rep in 1 2 3; do for k in 22 21 20; do for b in 0 1; do
    ZKHIP_BATCH_ABC=$b python bench.py --log2n $k --steps 30 --warmup 3 --no-cpu 2>/dev/null | line "2^$k ZKHIP_BATCH_ABC=$b"
  done; done; done
  for b in 0 1; do ZKHIP_BATCH_ABC=$b python bench.py --log2n 24 --steps 8 --warmup 2 --no-cpu 2>/dev/null | line "2^24 ZKHIP_BATCH_ABC=$b"; done
  for b in 0 1; do ZKHIP_BATCH_ABC=$b python bench.py --steps 20 --warmup 3 --no-cpu --witness realistic --shape circuit 2>/dev/null | line "2^22 circuit-shaped, realistic witness, ZKHIP_BATCH_ABC=$b"; done
  for b in 0 1; do ZKHIP_BATCH_ABC=$b python bench.py --steps 20 --warmup 3 --no-cpu --precomp 0 2>/dev/null | line "2^22 tables as in the zkey, ZKHIP_BATCH_ABC=$b"; done
) > $o/ab_batch_abc_unsharded.txt 2>&1
cat $o/ab_batch_abc_unsharded.txt
