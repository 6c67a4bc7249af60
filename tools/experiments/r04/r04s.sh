#!/bin/bash
export TMPDIR=/tmp
o=gpurun_out/r04s; mkdir -p $o
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1: host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'sync', d['ms_per_proof_sync'])"; }
( for depth in 2 3 4 6; do for b in 0 1; do
    ZKHIP_BATCH_ABC=$b python bench.py --steps 16 --warmup 3 --no-cpu --in-flight $depth 2>/dev/null | line "2^22, $depth in flight, ZKHIP_BATCH_ABC=$b"
  done; done ) > $o/batch_abc_by_depth.txt 2>&1
cat $o/batch_abc_by_depth.txt
( export ZKHIP_LIB=$PWD/rapidsnark-old_amd/libzkhip_probes.so
  for rep in 1 2; do for rc in 16 32 64; do
    ZKHIP_REDUCE_CHUNK=$rc python bench.py --steps 20 --warmup 3 --no-cpu 2>/dev/null | line "2^22 ZKHIP_REDUCE_CHUNK=$rc"
  done; done
  for rc in 16 32; do ZKHIP_REDUCE_CHUNK=$rc python bench.py --steps 20 --warmup 3 --no-cpu --witness realistic --shape circuit 2>/dev/null | line "2^22 circuit realistic ZKHIP_REDUCE_CHUNK=$rc"; done
  for rc in 16 32; do ZKHIP_REDUCE_CHUNK=$rc python bench.py --steps 30 --warmup 3 --no-cpu --log2n 20 2>/dev/null | line "2^20 ZKHIP_REDUCE_CHUNK=$rc"; done
) > $o/reduce_chunk.txt 2>&1
cat $o/reduce_chunk.txt
