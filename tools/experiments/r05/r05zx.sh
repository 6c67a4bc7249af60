#!/bin/bash
# level-1 rounds sized for fewer workgroups per CU than fit (G1: 3 fit), leaving wave slots to the other proofs' kernels (-DZK_PROBES build)
export TMPDIR=/tmp ZKHIP_LIB=$PWD/rapidsnark-old_amd/libzkhip_probes.so
out=gpurun_out/r05zx_round_workgroups.txt; : > $out
run() { python bench.py --warmup 5 --no-cpu --no-counters "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'one at a time', d['latency_ms_one_at_a_time']['witness_in_hbm'])"; }
for rep in 1 2; do
  for w in 0 2 1; do
    if [ $w = 0 ]; then unset ZKHIP_ACC_ROUND_WGS; else export ZKHIP_ACC_ROUND_WGS=$w; fi
    echo "2^22 workgroups per CU and round $w: $(run --steps 30)" >> $out
    echo "2^20 workgroups per CU and round $w: $(run --log2n 20 --steps 60)" >> $out
  done
done
cat $out
