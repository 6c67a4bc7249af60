#!/bin/bash
# entries per level-1 lane at least 128 for a proof submitted beside others (32 for a lone one): parity tests with the shipped
# build, then periods with the rule on (default) and off (ZKHIP_L1_CHUNK_MIN_BUSY=32) in the -DZK_PROBES build, by size
export TMPDIR=/tmp
python -m pytest tests/test_gpu_msm.py tests/test_gpu_prove.py tests/test_gpu_scale.py tests/test_gpu_synth.py -q -m gpu -x 2>&1 | tail -3
export ZKHIP_LIB=$PWD/rapidsnark-old_amd/libzkhip_probes.so
out=gpurun_out/r05zm_busy_chunk_min.txt; : > $out
run() { python bench.py --warmup 5 --no-cpu --no-counters "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'one at a time', d['latency_ms_one_at_a_time']['witness_in_hbm'])"; }
for rep in 1 2; do
  for v in "17 200" "18 100" "19 60" "20 30" "21 30" "22 30"; do
    set -- $v
    for cm in 128 32; do
      echo "2^$1 busy chunk_min $cm: $(ZKHIP_L1_CHUNK_MIN_BUSY=$cm ZKHIP_L1_CHUNK_MIN_BUSY_LOG=17 run --log2n $1 --steps $2)" >> $out
    done
  done
  for cm in 128 32; do
    echo "2^22 circuit-shaped realistic busy chunk_min $cm: $(ZKHIP_L1_CHUNK_MIN_BUSY=$cm run --witness realistic --shape circuit --steps 30)" >> $out
    echo "2^22 dense realistic busy chunk_min $cm: $(ZKHIP_L1_CHUNK_MIN_BUSY=$cm run --witness realistic --steps 30)" >> $out
  done
done
cat $out
