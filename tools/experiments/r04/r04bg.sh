#!/bin/bash
# same-box A/B: s_setprio 3 at the top of the follow-up kernels (merges, bucket reductions)
export TMPDIR=/tmp
o=gpurun_out/r04bg; mkdir -p $o
lib() { [ $1 = default ] && echo $PWD/rapidsnark-old_amd/libzkhip.so || echo $PWD/rapidsnark-old_amd/libzkhip_$1.so; }
( ZKHIP_LIB=$(lib prio) timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_prove.py -m gpu -x -q 2>&1 | tail -1
  for rep in 1 2 3; do for l in default prio; do
    for k in 14 16 18; do
      ZKHIP_LIB=$(lib $l) python bench.py --log2n $k --steps 400 --warmup 8 --no-cpu 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read());print('$l 2^$k: period', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'sync', d['ms_per_proof_sync'], 'one at a time', d['latency_ms_one_at_a_time']['witness_in_hbm'])"
    done
    ZKHIP_LIB=$(lib $l) python bench.py --log2n 16 --batch 8 --steps 400 --warmup 8 --no-cpu 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read());print('$l 2^16 x 8 per submission: period', d['ms_per_step'])"
    ZKHIP_LIB=$(lib $l) python bench.py --steps 20 --warmup 3 --no-cpu --no-2p20 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read());print('$l 2^22: period', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'sync', d['ms_per_proof_sync'], 'one at a time', d['latency_ms_one_at_a_time']['witness_in_hbm'])"
  done; done
  for l in default prio; do ZKHIP_LIB=$(lib $l) python tools/shard_probe.py 22 4,8 partitioned 2>/dev/null | grep world | cut -c1-110 | sed "s/^/$l /"; done
) > $o/tail_wave_prio.txt 2>&1
cat $o/tail_wave_prio.txt
