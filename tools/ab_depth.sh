#!/bin/bash
for rep in 1 2; do for d in 2 3; do
python bench.py --steps 15 --warmup 3 --no-cpu --in-flight $d 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('depth $d: host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'g1', d['stage_ms']['g1_l1_kernel'], 'g2', d['stage_ms']['g2_l1_kernel'], 'frac', d['roofline']['frac'])"
done; done
