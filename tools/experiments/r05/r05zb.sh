#!/bin/bash
# what do the follow-up kernels (partial merges, bucket reductions) and the sorts cost in the pipelined period?
# -DZK_PROBES build, WRONG results on purpose: periods only
export TMPDIR=/tmp ZKHIP_LIB=$PWD/rapidsnark-old_amd/libzkhip_probes.so
out=gpurun_out/r05zb.txt; : > $out
run() { python bench.py --steps 15 --warmup 3 --no-cpu --no-counters "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'one at a time', d['latency_ms_one_at_a_time']['witness_in_hbm'])"; }
for size in "" "--log2n 20"; do
  echo "size: ${size:-2^22}" >> $out
  for rep in 1 2; do
    for probe in none ZKHIP_PROBE_SKIP_FOLLOWUPS ZKHIP_PROBE_SKIP_SORT both; do
      unset ZKHIP_PROBE_SKIP_FOLLOWUPS ZKHIP_PROBE_SKIP_SORT
      case $probe in both) export ZKHIP_PROBE_SKIP_FOLLOWUPS=1 ZKHIP_PROBE_SKIP_SORT=1;; none) ;; *) export $probe=1;; esac
      echo "$probe: $(run $size)" >> $out
    done
  done
done
cat $out
