#!/bin/bash
# same-box A/B: Fq2 lane-pair arithmetic without v_cndmask on vcc (conditional negation / masking by xor, sub, and on an opaque lane mask)
export TMPDIR=/tmp
o=gpurun_out/r04aw; mkdir -p $o
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1: host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'one at a time', d['latency_ms_one_at_a_time']['witness_in_hbm'], 'sync', d['ms_per_proof_sync'])"; }
( for lib in prev new; do
    f=$PWD/rapidsnark-old_amd/libzkhip_$lib.so; [ $lib = new ] && f=$PWD/rapidsnark-old_amd/libzkhip.so
    ZKHIP_LIB=$f ZKHIP_SERIAL=1 python bench.py --steps 6 --warmup 2 --no-cpu --pipeline 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib serial: G1 launch per MSM', d['stage_ms']['g1_l1_kernel'], 'G2 launch', d['stage_ms']['g2_l1_kernel'], 'proof', d['ms_per_step'])"
  done
  for rep in 1 2 3; do for lib in prev new; do
    f=$PWD/rapidsnark-old_amd/libzkhip_$lib.so; [ $lib = new ] && f=$PWD/rapidsnark-old_amd/libzkhip.so
    ZKHIP_LIB=$f python bench.py --steps 20 --warmup 3 --no-cpu 2>/dev/null | line "2^22 $lib"
  done; done
  timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_prove.py -m gpu -x -q 2>&1 | tail -3
) > $o/fq2_no_cndmask.txt 2>&1
cat $o/fq2_no_cndmask.txt
