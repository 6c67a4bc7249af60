#!/bin/bash
export TMPDIR=/tmp
o=gpurun_out/r04e; mkdir -p $o
# sync A/B: staging pool + bucket memsets beside the upload (new) vs the round-3 library (old)
( for rep in 1 2 3; do
  for which in new old; do
    if [ $which = old ]; then export ZKHIP_LIB=$PWD/tools/_ab/libzkhip_old.so; else unset ZKHIP_LIB; fi
    for k in 22 20 18; do
      python tools/lone_proof.py $k 8 2>/dev/null | awk -v w=$which -v k=$k '/lone proof/ {s+=$4; n++} END {printf "%s lib 2^%d: %.2f ms per synchronous proof (mean of %d)\n", w, k, s/n, n}'
    done
  done
done ) > $o/ab_sync_lone.txt 2>&1
unset ZKHIP_LIB
cat $o/ab_sync_lone.txt
# (ii) limb-form tables: the G1 level-1 launch with 72-byte rows and no unpacking (wrong sums) against the probes build
( for rep in 1 2; do
for lib in probes limbrows; do
  ZKHIP_LIB=$PWD/rapidsnark-old_amd/libzkhip_$lib.so ZKHIP_SERIAL=1 python bench.py --steps 6 --warmup 2 --no-cpu --pipeline 0 2>$o/limb_$lib.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib (serial, one proof at a time): G1 level-1 launch', d['stage_ms']['g1_l1_kernel'], 'ms; G2', d['stage_ms']['g2_l1_kernel'], 'ms; proof', d['ms_per_step'])"
  ZKHIP_LIB=$PWD/rapidsnark-old_amd/libzkhip_$lib.so python bench.py --steps 12 --warmup 3 --no-cpu 2>>$o/limb_$lib.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib (pipelined): period', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'g1 launch', d['stage_ms']['g1_l1_kernel'])"
done
done ) > $o/limb_rows_probe.txt 2>&1
grep -v "^  File\|^Traceback\|json\.\|raise\|return\|obj, end" $o/limb_rows_probe.txt; tail -3 $o/limb_limbrows.err
# small shards
export ZKHIP_LIB=$PWD/rapidsnark-old_amd/libzkhip_probes.so
( echo "default"; python tools/shard_probe.py 22 8 partitioned 2>&1 | tail -2
  echo "ZKHIP_BATCH_ABC=1"; ZKHIP_BATCH_ABC=1 python tools/shard_probe.py 22 8 partitioned 2>&1 | tail -2
  for cm in 48 64 96; do echo "ZKHIP_ACC_CHUNK_MIN=$cm"; ZKHIP_ACC_CHUNK_MIN=$cm python tools/shard_probe.py 22 8 partitioned 2>&1 | tail -1; done
  echo "ZKHIP_BATCH_ABC=1 ZKHIP_ACC_CHUNK_MIN=64"; ZKHIP_BATCH_ABC=1 ZKHIP_ACC_CHUNK_MIN=64 python tools/shard_probe.py 22 8 partitioned 2>&1 | tail -1
  echo "default again"; python tools/shard_probe.py 22 8 partitioned 2>&1 | tail -1 ) > $o/shard8_experiments.txt 2>&1
unset ZKHIP_LIB
cat $o/shard8_experiments.txt
