#!/bin/bash
# Every number DESIGN.md quotes for a round, from ONE box:  tools/round_numbers.sh <tag>
tag=$1
out=gpurun_out/$tag/profiles
tools/profile_bench.sh $tag > gpurun_out/${tag}_profile.log 2>&1        # the driver's command: python bench.py (2^22 + the circuit-shaped leg + the 2^20 leg + CPU leg)
for k in 22 20; do                                                      # ONE synchronous proof as a timeline
  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/${tag}_lone$k -o t -- python tools/lone_proof.py $k 4 > gpurun_out/${tag}_lone$k.log 2>&1
  python tools/lone_timeline.py gpurun_out/${tag}_lone$k 100 > $out/${tag}_lone_proof_timeline_2p$k.txt 2>&1
  grep "^lone" gpurun_out/${tag}_lone$k.log >> $out/${tag}_lone_proof_timeline_2p$k.txt
  rm -rf gpurun_out/${tag}_lone$k
done
python bench.py --log2n 20 --no-2p20 > $out/${tag}_bench_2p20.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-cpu --witness realistic > $out/${tag}_bench_2p22_realistic.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-cpu --witness realistic --shape circuit > $out/${tag}_bench_2p22_circuit_realistic.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-cpu --precomp 0 > $out/${tag}_bench_2p22_plain.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-cpu --witness-in hbm > $out/${tag}_bench_2p22_resident.json 2>/dev/null
python bench.py --steps 8 --warmup 2 --no-cpu --log2n 24 > $out/${tag}_bench_2p24.json 2>/dev/null
python tools/shard_probe.py 22 1,2,4,8 2>&1 | grep world > $out/${tag}_shard_probe.txt
python tools/cli_timing.py 22 /tmp/zk_cli 2 2>&1 | grep -v amdgpu > $out/${tag}_cli_timing_2p22.txt
bash tools/ntt_counters.sh ${tag}_ntt 22 > /dev/null 2>&1; cp gpurun_out/${tag}_ntt/summary.txt $out/${tag}_ntt_pipeline_counters.txt
tools/mul_rate_probe > $out/${tag}_mul_rate.txt 2>&1
for f in $out/*bench*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[1].split("/")[-1], d.get("ms_per_step"), d.get("value"), d.get("resident_witness", d.get("host_witness", {})).get("ms_per_step"), d.get("latency_ms_one_at_a_time"))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
cat $out/${tag}_shard_probe.txt
for k in 14 16 18; do python bench.py --log2n $k --steps 100 --warmup 10 --no-cpu > $out/${tag}_bench_2p$k.json 2>/dev/null; python tools/server_bench.py $k 1024 0 input 2>/dev/null; python tools/server_bench.py $k 1024 0 witness 2>/dev/null; python bench.py --log2n $k --batch 4 --steps 512 --warmup 16 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C-ABI loop 2^$k, four witnesses per submission:', d['value'], 'proofs/s')"; done | tee $out/${tag}_server_throughput.txt
for k in 14 16 18; do python - "$out/${tag}_bench_2p$k.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[1].split("/")[-1], d.get("ms_per_step"), d.get("value"), d.get("resident_witness", {}).get("ms_per_step"), d.get("latency_ms_one_at_a_time"))
PY
done
