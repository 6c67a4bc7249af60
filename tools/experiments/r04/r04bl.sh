#!/bin/bash
# lone proofs with more, shorter rounds of level-1 lanes (chunk ceiling 160 -> 80 / 40), now that the chains' waves have priority
export TMPDIR=/tmp
o=gpurun_out/r04bl; mkdir -p $o
export ZKHIP_LIB=$PWD/rapidsnark-old_amd/libzkhip_probes.so
( for rep in 1 2 3; do for cm in 160 80 40; do
    ZKHIP_ACC_CHUNK_MAX=$cm python bench.py --steps 16 --warmup 3 --no-cpu --no-2p20 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read());print('chunk ceiling $cm, 2^22: period', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'sync', d['ms_per_proof_sync'], 'one at a time', d['latency_ms_one_at_a_time']['witness_in_hbm'])"
  done; done
  for cm in 160 80 40; do
    ZKHIP_ACC_CHUNK_MAX=$cm python bench.py --log2n 20 --steps 40 --warmup 3 --no-cpu 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read());print('chunk ceiling $cm, 2^20: period', d['ms_per_step'], 'sync', d['ms_per_proof_sync'], 'one at a time', d['latency_ms_one_at_a_time']['witness_in_hbm'])"
  done ) > $o/chunk_ceiling_lone.txt 2>&1
cat $o/chunk_ceiling_lone.txt
