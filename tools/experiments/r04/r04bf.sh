#!/bin/bash
# same-box A/B: the library compiled with -mllvm -amdgpu-sched-strategy=iterative-ilp / max-ilp against the default scheduler
export TMPDIR=/tmp
o=gpurun_out/r04bf; mkdir -p $o
lib() { [ $1 = default ] && echo $PWD/rapidsnark-old_amd/libzkhip.so || echo $PWD/rapidsnark-old_amd/libzkhip_$1.so; }
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1: host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'one at a time', d['latency_ms_one_at_a_time']['witness_in_hbm'], 'sync', d['ms_per_proof_sync'])"; }
( for l in default iilp milp; do
    ZKHIP_LIB=$(lib $l) ZKHIP_SERIAL=1 python bench.py --steps 8 --warmup 2 --no-cpu --pipeline 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('$l serial: G1 launch per MSM', s['g1_l1_kernel'], 'G2 launch', s['g2_l1_kernel'], 'chain', s['ntt_chain_wall'], 'spmv', s['spmv'], 'proof', d['ms_per_step'])"
  done
  for rep in 1 2 3; do for l in default iilp milp; do
    ZKHIP_LIB=$(lib $l) python bench.py --steps 20 --warmup 3 --no-cpu 2>/dev/null | line "2^22 $l"
  done; done
  for l in iilp milp; do ZKHIP_LIB=$(lib $l) timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_prove.py tests/test_gpu_field_ntt.py -m gpu -x -q 2>&1 | tail -1; done
) > $o/sched_strategy.txt 2>&1
cat $o/sched_strategy.txt
