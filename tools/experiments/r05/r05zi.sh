#!/bin/bash
# stream priorities in the pipelined mode (-DZK_PROBES build): the chain's stream (SpMV, transforms, sort(h), L1(H)) and/or
# stream 2 (sort(w), L1(B2), L1(A|B1|C)) of every lane at the high priority — do the starved transforms (50-100 ms beside
# level-1 launches in profiles/r05z kernel traces) cost the period?
export TMPDIR=/tmp ZKHIP_LIB=$PWD/rapidsnark-old_amd/libzkhip_probes.so
out=gpurun_out/r05zi_chain_priority.txt; : > $out
run() { python bench.py --steps 30 --warmup 3 --no-cpu --no-counters "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'one at a time', d['latency_ms_one_at_a_time']['witness_in_hbm'])"; }
for size in 22 20; do
  for rep in 1 2; do
    for pr in 0 1 2; do
      echo "2^$size chain_prio $pr: $(ZKHIP_PROBE_CHAIN_PRIO=$pr run --log2n $size)" >> $out
    done
  done
done
cat $out
