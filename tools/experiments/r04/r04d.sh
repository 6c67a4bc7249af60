#!/bin/bash
export TMPDIR=/tmp
o=gpurun_out/r04d; mkdir -p $o
timeout 1200 python -m pytest tests -m gpu -x -q > $o/pytest_gpu.log 2>&1; tail -6 $o/pytest_gpu.log
# (ii) limb-form tables: the G1 level-1 launch with 72-byte rows and no unpacking (wrong sums) against the probes build, one proof at a time
for rep in 1 2; do
for lib in probes limbrows; do
  ZKHIP_LIB=$PWD/rapidsnark-old_amd/libzkhip_$lib.so ZKHIP_SERIAL=1 python bench.py --steps 6 --warmup 2 --no-cpu --pipeline 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib (serial, one proof at a time): G1 level-1 launch', d['stage_ms']['g1_l1_kernel'], 'ms; G2', d['stage_ms']['g2_l1_kernel'], 'ms; proof', d['ms_per_step'])"
  ZKHIP_LIB=$PWD/rapidsnark-old_amd/libzkhip_$lib.so python bench.py --steps 12 --warmup 3 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib (pipelined): period', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'g1 launch', d['stage_ms']['g1_l1_kernel'])"
done
done > $o/limb_rows_probe.txt 2>&1
cat $o/limb_rows_probe.txt
# small shards (G = 8): A, B1, C as one set of launches; level-1 chunk floor
export ZKHIP_LIB=$PWD/rapidsnark-old_amd/libzkhip_probes.so
( echo "default"; python tools/shard_probe.py 22 8 partitioned 2>&1 | grep world
  echo "ZKHIP_BATCH_ABC=1"; ZKHIP_BATCH_ABC=1 python tools/shard_probe.py 22 8 partitioned 2>&1 | grep world
  for cm in 48 64 96; do echo "ZKHIP_ACC_CHUNK_MIN=$cm"; ZKHIP_ACC_CHUNK_MIN=$cm python tools/shard_probe.py 22 8 partitioned 2>&1 | grep world; done
  echo "ZKHIP_BATCH_ABC=1 ZKHIP_ACC_CHUNK_MIN=64"; ZKHIP_BATCH_ABC=1 ZKHIP_ACC_CHUNK_MIN=64 python tools/shard_probe.py 22 8 partitioned 2>&1 | grep world
  echo "default again"; python tools/shard_probe.py 22 8 partitioned 2>&1 | grep world ) > $o/shard8_experiments.txt 2>&1
unset ZKHIP_LIB
cat $o/shard8_experiments.txt
