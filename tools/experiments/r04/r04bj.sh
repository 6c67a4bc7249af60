#!/bin/bash
# shards: priority of the sort / transform chains 1 (default) against 2 and 3 (follow-ups at 3 in all)
export TMPDIR=/tmp
o=gpurun_out/r04bj; mkdir -p $o
( for rep in 1 2; do for l in default c2 c3; do
    f=$PWD/rapidsnark-old_amd/libzkhip.so; [ $l != default ] && f=$PWD/rapidsnark-old_amd/libzkhip_$l.so
    ZKHIP_LIB=$f python tools/shard_probe.py 22 4,8 partitioned 2>/dev/null | grep world | cut -c1-100 | sed "s/^/$l /"
    ZKHIP_LIB=$f python tools/shard_probe.py 24 8 partitioned 2>/dev/null | grep world | cut -c1-100 | sed "s/^/$l 2^24 /"
    ZKHIP_LIB=$f python bench.py --log2n 20 --steps 60 --warmup 4 --no-cpu 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read());print('$l 2^20: period', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'sync', d['ms_per_proof_sync'])"
  done; done ) > $o/chain_prio_shards.txt 2>&1
cat $o/chain_prio_shards.txt
