#!/bin/bash
# the busy-mode lanes for the shards of a sharded proof too? (-DZK_PROBES build; rank-0 share on one GPU, tools/shard_probe.py)
export TMPDIR=/tmp ZKHIP_LIB=$PWD/rapidsnark-old_amd/libzkhip_probes.so
out=gpurun_out/r05zt_busy_lanes_shards.txt; : > $out
for rep in 1 2; do
  echo "rule off:" >> $out
  python tools/shard_probe.py 22 2,4,8 partitioned 2>&1 | grep world | cut -c1-90 >> $out
  echo "rule on for shards:" >> $out
  ZKHIP_L1_BUSY_SHARDS=1 python tools/shard_probe.py 22 2,4,8 partitioned 2>&1 | grep world | cut -c1-90 >> $out
done
echo "2^24:" >> $out
python tools/shard_probe.py 24 8 partitioned 2>&1 | grep world | cut -c1-90 >> $out
ZKHIP_L1_BUSY_SHARDS=1 python tools/shard_probe.py 24 8 partitioned 2>&1 | grep world | cut -c1-90 >> $out
cat $out
