#!/bin/bash
export TMPDIR=/tmp
tools/profile_bench.sh r04z > gpurun_out/r04z_profile.log 2>&1
tail -5 gpurun_out/r04z_profile.log
cat gpurun_out/r04z/profiles/r04z_legs.txt
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04z/bench.json') if l.startswith('{')][-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'sync', d['ms_per_proof_sync'], 'resident', d['resident_witness']['ms_per_step'])
print('realistic', d['also_realistic']['ms_per_step'], d['also_realistic']['resident_witness']['ms_per_step'], d['also_realistic']['ms_per_proof_sync'])
print('2p20', d['also_2p20']['ms_per_step'], d['also_2p20']['ms_per_proof_sync'])
r=d['roofline']; print('frac', r['frac'], r['launch_ms'], r['issue_bound']['bound_ms'], r['issue_bound']['achieved_frac'], r['issue_bound']['clock_ghz'])
PY
