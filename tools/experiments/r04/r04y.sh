#!/bin/bash
export TMPDIR=/tmp
o=gpurun_out/r04y; mkdir -p $o
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1: host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'one at a time', d['latency_ms_one_at_a_time']['witness_in_hbm'], 'lone G1 launch per MSM', d['roofline']['launch_ms_one_proof_in_flight'], 'sync', d['ms_per_proof_sync'])"; }
( for rep in 1 2 3; do for lib in probes l1b128 l1b64; do
    ZKHIP_LIB=$PWD/rapidsnark-old_amd/libzkhip_$lib.so python bench.py --steps 20 --warmup 3 --no-cpu 2>/dev/null | line "2^22 level-1 workgroups: $lib"
  done; done
  for lib in probes l1b64; do
    ZKHIP_LIB=$PWD/rapidsnark-old_amd/libzkhip_$lib.so ZKHIP_SERIAL=1 python bench.py --steps 6 --warmup 2 --no-cpu --pipeline 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib serial: G1 launch per MSM', d['stage_ms']['g1_l1_kernel'], 'G2', d['stage_ms']['g2_l1_kernel'], 'proof', d['ms_per_step'])"
  done
  for lib in probes l1b64; do ZKHIP_LIB=$PWD/rapidsnark-old_amd/libzkhip_$lib.so python bench.py --steps 30 --warmup 3 --no-cpu --log2n 20 2>/dev/null | line "2^20 $lib"; done
) > $o/l1_block.txt 2>&1
cat $o/l1_block.txt
