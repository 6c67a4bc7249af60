#!/bin/bash
out=gpurun_out/r03j
mkdir -p $out
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 ) > $out/pytest.txt
cat $out/pytest.txt
for rep in 1 2; do
for which in tree radix2 old; do
  unset ZKHIP_LIB ZKHIP_NTT_RADIX2
  if [ $which = old ]; then export ZKHIP_LIB=$PWD/tools/_ab/libzkhip_old.so; fi
  if [ $which = radix2 ]; then export ZKHIP_LIB=$PWD/rapidsnark-old_amd/libzkhip_probes.so ZKHIP_NTT_RADIX2=1; fi
  python bench.py --steps 15 --warmup 3 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$which: host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'latency', d['latency_ms_one_at_a_time'], 'ntt', d['stage_ms']['ntt_chain_wall'])"
done
done > $out/ab.txt 2>&1
cat $out/ab.txt
for k in 20 16; do
for which in tree radix2; do
  unset ZKHIP_LIB ZKHIP_NTT_RADIX2
  if [ $which = radix2 ]; then export ZKHIP_LIB=$PWD/rapidsnark-old_amd/libzkhip_probes.so ZKHIP_NTT_RADIX2=1; fi
  python bench.py --log2n $k --steps 60 --warmup 6 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('2^$k $which: host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'latency', d['latency_ms_one_at_a_time'])"
done
done > $out/ab_small.txt 2>&1
cat $out/ab_small.txt
