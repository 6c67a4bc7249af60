#!/bin/bash
# same-box A/B of builds of libzkhip.so: the tree's ("new") against tools/_ab/libzkhip_<name>.so for every name given
#   tools/ab_lib.sh "old g2w3" [bench args]        (default: "old"; REPS alternations, default 2)
names=${1:-old}; shift
for rep in $(seq 1 ${REPS:-2}); do
for which in new $names; do
  if [ $which = new ]; then unset ZKHIP_LIB; else export ZKHIP_LIB=$PWD/tools/_ab/libzkhip_$which.so; fi
  python bench.py --steps 15 --warmup 3 --no-cpu --no-counters "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$which: host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'one at a time', d['latency_ms_one_at_a_time']['witness_in_hbm'], 'g1 alone', r['launch_ms_one_in_flight'], 'g2 alone', r['g2_launch_ms_one_in_flight'])"
done
done
