#!/bin/bash
# with the cheaper bucket reduction, does a wider window pay where it did not before?  (in-tree build; parity tests first)
export TMPDIR=/tmp
python -m pytest tests/test_gpu_msm.py tests/test_gpu_prove.py tests/test_gpu_scale.py -q -m gpu -x 2>&1 | tail -3
out=gpurun_out/r05ze_window_sweep.txt; : > $out
run() { python bench.py --steps 15 --warmup 3 --no-cpu --no-counters "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'one at a time', d['latency_ms_one_at_a_time']['witness_in_hbm'], 'windows', d['config'].get('window_bits'), d['config'].get('windows'))"; }
for v in "18 0 17" "19 0 18" "20 0 20" "21 0 20" ; do
  set -- $v
  for rep in 1 2; do
    for wb in $2 $3; do
      echo "2^$1 window_bits $wb: $(run --log2n $1 --window-bits $wb)" >> $out
    done
  done
done
cat $out
