#!/bin/bash
# shards of a 2^22 proof one at a time: B2's merges + reduction on the finishing stream (the condition now looks at the witness SLICE)
export TMPDIR=/tmp
o=gpurun_out/r04au; mkdir -p $o
export ZKHIP_LIB=$PWD/rapidsnark-old_amd/libzkhip_probes.so
( for rep in 1 2 3; do for mx in 0 19; do
    ZKHIP_G2_ASIDE_MAXLOG=$mx python tools/shard_probe.py 22 2,4,8 partitioned 2>/dev/null | grep world | sed "s/^/B2 follow-ups aside up to 2^$mx: /" | cut -c1-200
  done; done ) > $o/shards_g2_aside.txt 2>&1
cat $o/shards_g2_aside.txt
