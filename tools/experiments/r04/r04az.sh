#!/bin/bash
# same-box A/B: table rows of the level-1 kernels (G1 and G2) loaded with the non-temporal hint (read once per MSM)
export TMPDIR=/tmp
o=gpurun_out/r04az; mkdir -p $o
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1: host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'one at a time', d['latency_ms_one_at_a_time']['witness_in_hbm'], 'sync', d['ms_per_proof_sync'], 'clock', d['roofline']['issue_bound']['clock_ghz'])"; }
( for lib in probes nt; do
    ZKHIP_LIB=$PWD/rapidsnark-old_amd/libzkhip_$lib.so ZKHIP_SERIAL=1 python bench.py --steps 12 --warmup 2 --no-cpu --pipeline 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib serial: G1 launch per MSM', d['stage_ms']['g1_l1_kernel'], 'G2 launch', d['stage_ms']['g2_l1_kernel'], 'proof', d['ms_per_step'], 'clock', d['roofline']['issue_bound']['clock_ghz'])"
  done
  for rep in 1 2 3; do for lib in probes nt; do
    ZKHIP_LIB=$PWD/rapidsnark-old_amd/libzkhip_$lib.so python bench.py --steps 20 --warmup 3 --no-cpu 2>/dev/null | line "2^22 $lib"
  done; done
  ZKHIP_LIB=$PWD/rapidsnark-old_amd/libzkhip_nt.so timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_prove.py -m gpu -x -q 2>&1 | tail -3
) > $o/nt_gather.txt 2>&1
cat $o/nt_gather.txt
