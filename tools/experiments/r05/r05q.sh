#!/bin/bash
# does the table FOOTPRINT of the A|B1|C launch matter?  probes build; all three MSMs gathering from table A (3.25 GiB) against A, B1, C (9.75 GiB)
export ZKHIP_LIB=$PWD/rapidsnark-old_amd/libzkhip_probes.so
for rep in 1 2 3; do for one in 0 1; do
  if [ $one = 1 ]; then export ZKHIP_PROBE_ONE_TABLE=1; else unset ZKHIP_PROBE_ONE_TABLE; fi
  python bench.py --steps 15 --warmup 3 --no-cpu --no-counters 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('one table $one: host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'one at a time', d['latency_ms_one_at_a_time']['witness_in_hbm'], 'g1 per MSM alone', r['launch_ms_one_in_flight'], 'in the pipeline', r['launch_ms'], 'g2 alone', r['g2_launch_ms_one_in_flight'])"
done; done
