#!/bin/bash
# A|B1|C as one set of launches at 2^20 / 2^21 (off there since round 4), again, now that busy proofs run fewer, longer lanes
export TMPDIR=/tmp
out=gpurun_out/r05zp_batch_abc_mid_sizes.txt; : > $out
run() { python bench.py --warmup 5 --no-cpu --no-counters "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'one at a time', d['latency_ms_one_at_a_time']['witness_in_hbm'])"; }
for rep in 1 2 3; do
  for b in 0 1; do
    echo "2^20 one launch $b: $(ZKHIP_BATCH_ABC=$b run --log2n 20 --steps 60)" >> $out
    echo "2^21 one launch $b: $(ZKHIP_BATCH_ABC=$b run --log2n 21 --steps 30)" >> $out
  done
done
cat $out
