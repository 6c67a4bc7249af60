#!/bin/bash
# what do the two sorts of a proof cost the PERIOD?  probes build; digits + counting sort run once per buffer set and are reused (wrong sums)
export ZKHIP_LIB=$PWD/rapidsnark-old_amd/libzkhip_probes.so
for k in 22 20; do for rep in 1 2 3; do for skip in 0 1; do
  if [ $skip = 1 ]; then export ZKHIP_PROBE_SKIP_SORT=1; else unset ZKHIP_PROBE_SKIP_SORT; fi
  python bench.py --log2n $k --steps 15 --warmup 3 --no-cpu --no-counters 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('2^$k sorts skipped $skip: host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'one at a time', d['latency_ms_one_at_a_time']['witness_in_hbm'], 'sync', d['ms_per_proof_sync'])"
done; done; done
