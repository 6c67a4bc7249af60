#!/bin/bash
# what do the transforms cost the pipelined period?  made free in a -DZK_PROBES build (wrong results), alone and with the sorts and the
# follow-ups free as well: what is left is the level-1 launches + SpMV
export TMPDIR=/tmp ZKHIP_LIB=$PWD/rapidsnark-old_amd/libzkhip_probes.so
out=gpurun_out/r05zz_transforms_made_free.txt; : > $out
run() { python bench.py --warmup 5 --no-cpu --no-counters "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'one at a time', d['latency_ms_one_at_a_time']['witness_in_hbm'])"; }
for size in 22 20; do
  for rep in 1 2; do
    for probe in none ntt ntt+sort all; do
      unset ZKHIP_PROBE_SKIP_NTT ZKHIP_PROBE_SKIP_SORT ZKHIP_PROBE_SKIP_FOLLOWUPS
      case $probe in ntt) export ZKHIP_PROBE_SKIP_NTT=1;; ntt+sort) export ZKHIP_PROBE_SKIP_NTT=1 ZKHIP_PROBE_SKIP_SORT=1;; all) export ZKHIP_PROBE_SKIP_NTT=1 ZKHIP_PROBE_SKIP_SORT=1 ZKHIP_PROBE_SKIP_FOLLOWUPS=1;; esac
      echo "2^$size free: $probe: $(run --log2n $size --steps 30)" >> $out
    done
  done
done
cat $out
