#!/bin/bash
# busy mode, continued: even fewer rounds of lanes (640, 1280, one round whatever the size); -DZK_PROBES build
export TMPDIR=/tmp ZKHIP_LIB=$PWD/rapidsnark-old_amd/libzkhip_probes.so
out=gpurun_out/r05zo_busy_chunk_max_large.txt; : > $out
run() { python bench.py --warmup 5 --no-cpu --no-counters "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'one at a time', d['latency_ms_one_at_a_time']['witness_in_hbm'])"; }
for rep in 1 2; do
  for cm in 640 1280 1000000; do
    echo "2^22 busy chunk_max $cm: $(ZKHIP_L1_CHUNK_MAX_BUSY=$cm run --steps 30)" >> $out
  done
  for cm in 640 1280 1000000; do
    echo "2^24 busy chunk_max $cm: $(ZKHIP_L1_CHUNK_MAX_BUSY=$cm run --log2n 24 --steps 8 --warmup 2)" >> $out
  done
done
cat $out
