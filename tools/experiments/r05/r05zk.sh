#!/bin/bash
# circuit-shaped key + realistic witness: the partial merges are 7.7 % of its instructions (k_msm_accum_wave; counters of r05zj) because
# the lanes of its sparse witness MSMs take the minimum chunk of 32 entries.  Larger minimum chunks (-DZK_PROBES build)?
export TMPDIR=/tmp ZKHIP_LIB=$PWD/rapidsnark-old_amd/libzkhip_probes.so
out=gpurun_out/r05zk_chunk_min_realistic.txt; : > $out
run() { python bench.py --steps 30 --warmup 5 --no-cpu --no-counters "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'one at a time', d['latency_ms_one_at_a_time']['witness_in_hbm'])"; }
for rep in 1 2; do
  for cm in 32 64 128 256; do
    echo "circuit-shaped realistic chunk_min $cm: $(ZKHIP_ACC_CHUNK_MIN=$cm run --witness realistic --shape circuit)" >> $out
  done
done
for cm in 32 64 128; do
  echo "dense realistic chunk_min $cm: $(ZKHIP_ACC_CHUNK_MIN=$cm run --witness realistic)" >> $out
  echo "dense 2^20 chunk_min $cm: $(ZKHIP_ACC_CHUNK_MIN=$cm run --log2n 20)" >> $out
done
cat $out
