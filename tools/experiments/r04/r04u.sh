#!/bin/bash
export TMPDIR=/tmp
o=gpurun_out/r04u; mkdir -p $o
timeout 1500 python -m pytest tests -m gpu -x -q > $o/pytest_gpu.log 2>&1; tail -4 $o/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --gpus 1 --steps 20 --warmup 5 > $o/bench_driver_cmd.json 2> $o/bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04u/bench_driver_cmd.json') if l.startswith('{')][-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'sync', d['ms_per_proof_sync'], 'resident', d['resident_witness']['ms_per_step'])
print('realistic', d['also_realistic']['ms_per_step'], d['also_realistic']['ms_per_proof_sync'], d['also_realistic']['config'].get('witness_msm_window_bits'))
print('2p20', d['also_2p20']['ms_per_step'], d['also_2p20']['ms_per_proof_sync'])
r=d['roofline']; print('frac', r['frac'], r['launch_ms'], 'traffic', r['traffic'], r['issue_bound']['bound_ms'], r['issue_bound']['achieved_frac'], r['issue_bound']['clock_ghz'])
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['gpu_proof_bit_exact_vs_cpu'])
PY
