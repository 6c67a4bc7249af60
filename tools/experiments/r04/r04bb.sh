#!/bin/bash
# same-box A/B: exclusive scans as two launches (block totals summed inside the add pass) instead of three
export TMPDIR=/tmp
o=gpurun_out/r04bb; mkdir -p $o
( timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_prove.py tests/test_gpu_field_ntt.py -m gpu -x -q 2>&1 | tail -2
  for rep in 1 2 3; do for lib in prev new; do
    f=$PWD/rapidsnark-old_amd/libzkhip_$lib.so; [ $lib = new ] && f=$PWD/rapidsnark-old_amd/libzkhip.so
    for k in 14 16 18; do
      ZKHIP_LIB=$f python bench.py --log2n $k --steps 400 --warmup 8 --no-cpu 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read());print('$lib 2^$k: period', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'sync', d['ms_per_proof_sync'], 'one at a time', d['latency_ms_one_at_a_time'])"
    done
  done; done
  for lib in prev new; do
    f=$PWD/rapidsnark-old_amd/libzkhip_$lib.so; [ $lib = new ] && f=$PWD/rapidsnark-old_amd/libzkhip.so
    ZKHIP_LIB=$f python bench.py --steps 16 --warmup 3 --no-cpu 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read());print('$lib 2^22: period', d['ms_per_step'], 'sync', d['ms_per_proof_sync'], '2^20', d['also_2p20']['ms_per_step'], d['also_2p20']['ms_per_proof_sync'])"
  done
) > $o/scan_two_launches.txt 2>&1
cat $o/scan_two_launches.txt
