#!/bin/bash
out=gpurun_out/r03m
mkdir -p $out
( timeout 600 python -m pytest tests/test_gpu_msm.py tests/test_gpu_prove.py -m gpu -x -q 2>&1 | tail -3 ) > $out/pytest.txt
cat $out/pytest.txt
for rep in 1 2 3; do
for which in tree nofilter g2w3; do
  unset ZKHIP_LIB
  if [ $which != tree ]; then export ZKHIP_LIB=$PWD/tools/_ab/libzkhip_$which.so; fi
  python bench.py --steps 15 --warmup 3 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$which: host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'latency', d['latency_ms_one_at_a_time']['witness_in_hbm'], 'g1', d['stage_ms']['g1_l1_kernel'], 'g2', d['stage_ms']['g2_l1_kernel'], 'lone g1', d['roofline']['launch_ms_one_proof_in_flight'])"
done
done > $out/ab.txt 2>&1
cat $out/ab.txt
