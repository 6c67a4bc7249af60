#!/bin/bash
# kernel timeline of ONE rank's share of a 2^22 proof over 8 (and 4) shards, chain partitioned
export TMPDIR=/tmp
o=$PWD/gpurun_out/r04bp; mkdir -p $o
for G in 8 4; do
  ( cd /tmp && rocprofv3 --kernel-trace -d /tmp/rp_sh$G -o t --output-format csv -- python $OLDPWD/tools/shard_lone.py 22 $G 4 > $o/shard_lone_$G.log 2>&1 )
  python tools/lone_timeline.py /tmp/rp_sh$G 15 -1 > $o/shard_timeline_2p22_of_$G.txt 2>&1
  cat $o/shard_lone_$G.log | tail -4 >> $o/shard_timeline_2p22_of_$G.txt
done
cat $o/shard_timeline_2p22_of_8.txt
