#!/bin/bash
# the bit-sum tree (8 wave-level additions per bucket, shallow) against the split form (2.4, deeper) for sets of 2^15 / 2^16 buckets:
# 2^18 and 2^19 circuits, eight shards of 2^22  (-DZK_PROBES build, ZKHIP_REDUCE_BITS=0 turns the tree off)
export TMPDIR=/tmp ZKHIP_LIB=$PWD/rapidsnark-old_amd/libzkhip_probes.so
out=gpurun_out/r05zu_bits_tree_or_split.txt; : > $out
run() { python bench.py --warmup 5 --no-cpu --no-counters "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'one at a time', d['latency_ms_one_at_a_time']['witness_in_hbm'])"; }
for rep in 1 2; do
  for k in 18 19; do
    echo "2^$k bit-sum tree: $(run --log2n $k --steps 100)" >> $out
    echo "2^$k split form:   $(ZKHIP_REDUCE_BITS=0 run --log2n $k --steps 100)" >> $out
  done
  echo "bit-sum tree: $(python tools/shard_probe.py 22 8 partitioned 2>&1 | grep world | cut -c1-90)" >> $out
  echo "split form:   $(ZKHIP_REDUCE_BITS=0 python tools/shard_probe.py 22 8 partitioned 2>&1 | grep world | cut -c1-90)" >> $out
done
cat $out
