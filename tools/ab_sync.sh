#!/bin/bash
# same-box A/B of two builds of libzkhip.so on the SYNCHRONOUS proof (SURVEY section 8(d)'s ms/proof) and the pipelined period:
#   tools/ab_sync.sh [old lib = tools/_ab/libzkhip_old.so]
old=${1:-$PWD/tools/_ab/libzkhip_old.so}
for rep in 1 2 3; do
for which in new old; do
  if [ $which = old ]; then export ZKHIP_LIB=$old; else unset ZKHIP_LIB; fi
  for k in 22 20; do
  python bench.py --steps 12 --warmup 3 --no-cpu --log2n $k 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$which 2^$k: period host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], '| one at a time: resident', d['latency_ms_one_at_a_time']['witness_in_hbm'], 'SYNC host witness', d['ms_per_proof_sync'])"
  done
done
done
