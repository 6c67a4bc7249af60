#!/bin/bash
# small circuits: is the period bound by launches (front end) or by the kernels?  kernel trace of the headline leg at 2^14 / 2^16 (+ batch 4)
export TMPDIR=/tmp
o=$PWD/gpurun_out/r04ae; mkdir -p $o
for cfg in "14 1" "16 1" "16 4" "18 1"; do set -- $cfg; k=$1; b=$2
  rm -rf /tmp/rp_${k}_$b
  ( cd /tmp && ZK_BENCH_LEG_MARKERS=1 ZK_BENCH_CLOCK=0 rocprofv3 --kernel-trace -d /tmp/rp_${k}_$b -o t --output-format csv -- python $OLDPWD/bench.py --log2n $k --steps 200 --warmup 8 --batch $b --no-cpu > $o/bench_${k}_b$b.json 2> $o/bench_${k}_b$b.err )
  python tools/busy.py /tmp/rp_${k}_$b $o/bench_${k}_b$b.err 2p${k}_headline 200 > $o/busy_2p${k}_batch$b.txt 2>&1
  python -c "import json;d=json.loads([l for l in open('$o/bench_${k}_b$b.json') if l.startswith('{')][-1]);print('2^$k batch $b under the tracer:', d['ms_per_step'], 'ms per proof')" >> $o/busy_2p${k}_batch$b.txt
  python bench.py --log2n $k --steps 200 --warmup 8 --batch $b --no-cpu 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read());print('2^$k batch $b without:', d['ms_per_step'], 'ms per proof; resident', d['resident_witness']['ms_per_step'], '; one at a time', d['latency_ms_one_at_a_time'])" >> $o/busy_2p${k}_batch$b.txt
  cat $o/busy_2p${k}_batch$b.txt
done
