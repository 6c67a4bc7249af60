#!/bin/bash
# Shader clock, power and temperature while bench.py runs 2^22 proofs (rocm-smi samples every 1.5 s): bash tools/power_probe.sh
python bench.py --no-cpu --no-2p20 --steps 120 --warmup 3 > /tmp/b.json 2>/dev/null &
BP=$!
sleep 22
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -i -E "sclk|mclk|power|junction|Temperature \(Sensor" | head -8
  echo ---
  sleep 1.5
done
wait $BP
python -c "import json; d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'])"
rocm-smi --showclocks --showpower 2>/dev/null | grep -i -E "sclk|power" | head -4
rocm-smi --showmaxpower 2>/dev/null | grep -i power | head -3
