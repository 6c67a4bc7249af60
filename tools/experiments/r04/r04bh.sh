#!/bin/bash
# same-box A/B: wave priority (s_setprio) — follow-up kernels at 3 alone (prio), plus the sort and transform chains at 2 (pc2) or 1 (pc1)
export TMPDIR=/tmp
o=gpurun_out/r04bh; mkdir -p $o
lib() { [ $1 = default ] && echo $PWD/rapidsnark-old_amd/libzkhip.so || echo $PWD/rapidsnark-old_amd/libzkhip_$1.so; }
( for l in pc2 pc1; do ZKHIP_LIB=$(lib $l) timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_prove.py tests/test_gpu_field_ntt.py -m gpu -x -q 2>&1 | tail -1; done
  for rep in 1 2 3; do for l in default prio pc2 pc1; do
    for k in 14 16 18 20; do
      ZKHIP_LIB=$(lib $l) python bench.py --log2n $k --steps $((k < 20 ? 400 : 60)) --warmup 8 --no-cpu 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read());print('$l 2^$k: period', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'sync', d['ms_per_proof_sync'], 'one at a time', d['latency_ms_one_at_a_time']['witness_in_hbm'])"
    done
    ZKHIP_LIB=$(lib $l) python bench.py --steps 20 --warmup 3 --no-cpu --no-2p20 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read());print('$l 2^22: period', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'sync', d['ms_per_proof_sync'], 'one at a time', d['latency_ms_one_at_a_time']['witness_in_hbm'])"
  done; done
) > $o/wave_prio_variants.txt 2>&1
cat $o/wave_prio_variants.txt
