#!/bin/bash
# level-1 lanes: minimum and maximum entries per lane in the pipelined mode (-DZK_PROBES build), dense legs
export TMPDIR=/tmp ZKHIP_LIB=$PWD/rapidsnark-old_amd/libzkhip_probes.so
out=gpurun_out/r05zl_chunk_min_max_dense.txt; : > $out
run() { python bench.py --steps 30 --warmup 5 --no-cpu --no-counters "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'one at a time', d['latency_ms_one_at_a_time']['witness_in_hbm'])"; }
for rep in 1 2; do
  for v in "32 160" "128 160" "128 320" "32 320" "128 640"; do
    set -- $v
    echo "2^22 chunk_min $1 chunk_max $2: $(ZKHIP_ACC_CHUNK_MIN=$1 ZKHIP_ACC_CHUNK_MAX=$2 run)" >> $out
  done
done
for rep in 1 2; do
  for cm in 32 96 128 192; do
    echo "2^20 chunk_min $cm: $(ZKHIP_ACC_CHUNK_MIN=$cm run --log2n 20)" >> $out
  done
done
for cm in 32 128 192; do
  echo "2^21 chunk_min $cm: $(ZKHIP_ACC_CHUNK_MIN=$cm run --log2n 21)" >> $out
  echo "2^18 chunk_min $cm: $(ZKHIP_ACC_CHUNK_MIN=$cm run --log2n 18 --steps 100)" >> $out
  echo "2^16 chunk_min $cm: $(ZKHIP_ACC_CHUNK_MIN=$cm run --log2n 16 --steps 200)" >> $out
done
cat $out
