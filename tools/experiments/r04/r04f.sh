#!/bin/bash
export TMPDIR=/tmp
o=gpurun_out/r04f; mkdir -p $o
export ZKHIP_LIB=$PWD/rapidsnark-old_amd/libzkhip_probes.so
( for G in 4 2; do
    echo "G=$G default"; python tools/shard_probe.py 22 $G partitioned 2>&1 | tail -1
    echo "G=$G ZKHIP_BATCH_ABC=1"; ZKHIP_BATCH_ABC=1 python tools/shard_probe.py 22 $G partitioned 2>&1 | tail -1
  done
  for k in 19 20; do for b in 0 1; do
    ZKHIP_BATCH_ABC=$b python bench.py --log2n $k --steps 40 --warmup 5 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('unsharded 2^$k ZKHIP_BATCH_ABC=$b: period host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'sync', d['ms_per_proof_sync'])"
  done; done
  echo "G=8 at 2^24 default"; python tools/shard_probe.py 24 8 partitioned 2>&1 | tail -1
  echo "G=8 at 2^24 ZKHIP_BATCH_ABC=1"; ZKHIP_BATCH_ABC=1 python tools/shard_probe.py 24 8 partitioned 2>&1 | tail -1
) > $o/batch_abc_experiments.txt 2>&1
unset ZKHIP_LIB
cat $o/batch_abc_experiments.txt
tools/profile_bench.sh r04p > $o/profile.log 2>&1
tail -25 $o/profile.log
cat gpurun_out/r04p/profiles/r04p_legs.txt
python -c "
import json
d=json.loads([l for l in open('gpurun_out/r04p/bench.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['ms_per_proof_sync'], d['also_2p20']['ms_per_step'], d['also_2p20']['ms_per_proof_sync'])
print(d.get('also_realistic',{}).get('ms_per_step'), d.get('also_realistic',{}).get('ms_per_proof_sync'))
print(d['roofline']['issue_bound'])
"
