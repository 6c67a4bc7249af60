#!/bin/bash
export TMPDIR=/tmp
o=gpurun_out/r04w; mkdir -p $o
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1: host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'g1 launch/MSM', d['stage_ms']['g1_l1_kernel'], 'g2', d['stage_ms']['g2_l1_kernel'], 'sync', d['ms_per_proof_sync'])"; }
export ZKHIP_LIB=$PWD/rapidsnark-old_amd/libzkhip_probes.so
( for rep in 1 2; do for cm in 160 112 224 320; do
    ZKHIP_ACC_CHUNK_MAX=$cm python bench.py --steps 20 --warmup 3 --no-cpu 2>/dev/null | line "2^22 ZKHIP_ACC_CHUNK_MAX=$cm"
  done; done ) > $o/chunk_max.txt 2>&1
cat $o/chunk_max.txt
