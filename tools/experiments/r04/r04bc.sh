#!/bin/bash
# end-of-round check of the committed state on one box: build check, full GPU suite, smoke(), the driver's bench command
export TMPDIR=/tmp
o=gpurun_out/r04bc; mkdir -p $o
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
  python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
  python bench.py 2>/dev/null | tail -1 > $o/bench.json
  python -c "import json; d=json.loads(open('$o/bench.json').read()); print('bench:', d['value'], d['unit'], d['ms_per_step'], 'ms; sync', d['ms_per_proof_sync'], '; realistic', d['also_realistic']['ms_per_step'], '; 2^20', d['also_2p20']['ms_per_step'], d['also_2p20']['ms_per_proof_sync'], '; frac', d['roofline']['frac'], 'issue', d['roofline']['issue_bound']['achieved_frac'], 'clock', d['roofline']['issue_bound']['clock_ghz'], '; traffic', d['roofline']['traffic'], '; cpu', d['cpu_baseline']['value'])"
) > $o/final_check.txt 2>&1
cat $o/final_check.txt
