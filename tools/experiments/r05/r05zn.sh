#!/bin/bash
# busy mode: entries per lane beyond which another round of lanes is launched (160 = as for a lone proof, 320, 640); -DZK_PROBES build
export TMPDIR=/tmp ZKHIP_LIB=$PWD/rapidsnark-old_amd/libzkhip_probes.so
out=gpurun_out/r05zn_busy_chunk_max.txt; : > $out
run() { python bench.py --warmup 5 --no-cpu --no-counters "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'one at a time', d['latency_ms_one_at_a_time']['witness_in_hbm'])"; }
for rep in 1 2 3; do
  for cm in 160 320 640; do
    echo "2^22 busy chunk_max $cm: $(ZKHIP_L1_CHUNK_MAX_BUSY=$cm run --steps 30)" >> $out
  done
done
for cm in 160 320 640; do
  echo "2^24 busy chunk_max $cm: $(ZKHIP_L1_CHUNK_MAX_BUSY=$cm run --log2n 24 --steps 8 --warmup 2)" >> $out
  echo "2^22 plain tables busy chunk_max $cm: $(ZKHIP_L1_CHUNK_MAX_BUSY=$cm run --precomp 0 --steps 20)" >> $out
done
cat $out
