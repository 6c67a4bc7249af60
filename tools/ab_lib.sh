#!/bin/bash
# same-box A/B of two builds of libzkhip.so: tools/_ab/libzkhip_old.so vs the tree's.  tools/ab_lib.sh [bench args]
for rep in 1 2; do
for which in new old; do
  if [ $which = old ]; then export ZKHIP_LIB=$PWD/tools/_ab/libzkhip_old.so; else unset ZKHIP_LIB; fi
  python bench.py --steps 15 --warmup 3 --no-cpu "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$which: host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'latency', d['latency_ms_one_at_a_time']['witness_in_hbm'], 'g1', d['stage_ms']['g1_l1_kernel'], 'g2', d['stage_ms']['g2_l1_kernel'])"
done
done
