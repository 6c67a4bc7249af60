#!/bin/bash
# the split reduction's two parameters (-DZK_PROBES build): buckets per lane of the first level, chunk of the T level
export TMPDIR=/tmp
out=gpurun_out/r05zd_split_parameters.txt; : > $out
export ZKHIP_LIB=$PWD/rapidsnark-old_amd/libzkhip_probes.so
run() { python bench.py --steps 15 --warmup 3 --no-cpu --no-counters "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'one at a time', d['latency_ms_one_at_a_time']['witness_in_hbm'])"; }
for size in 22 20; do
  echo "parameters at 2^$size (probes build)" >> $out
  for rep in 1 2; do
    for v in "16 4" "0 4" "8 4" "32 4" "16 2" "16 8" "8 8"; do
      set -- $v
      echo "split $1 top $2: $(ZKHIP_REDUCE_SPLIT=$1 ZKHIP_REDUCE_SPLIT_TOP=$2 run --log2n $size)" >> $out
    done
  done
done
cat $out
