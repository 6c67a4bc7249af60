#!/bin/bash
# lone proofs: witness MSMs behind the transform chain (ZKHIP_LONE_ORDER=1, now also with A|B1|C batched) with sort workgroups of
# 1024 / 512 / 256 threads — 512 fits beside the G2 level-1 launch's waves (2 x 185 VGPRs per SIMD), 256 beside the G1 one's too
export TMPDIR=/tmp
o=gpurun_out/r04bq; mkdir -p $o
(   for rep in 1 2; do for l in probes s512 s256; do for lo in 0 1; do
    ZKHIP_LIB=$PWD/rapidsnark-old_amd/libzkhip_$l.so ZKHIP_LONE_ORDER=$lo python bench.py --steps 12 --warmup 3 --no-cpu 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read());print('$l lone order $lo: 2^22 period', d['ms_per_step'], 'sync', d['ms_per_proof_sync'], 'one at a time resident', d['latency_ms_one_at_a_time']['witness_in_hbm'])"
  done; done; done ) > $o/lone_order_sort_threads.txt 2>&1
cat $o/lone_order_sort_threads.txt
