#!/bin/bash
# Every number DESIGN.md quotes for a round, from ONE box:  tools/round_numbers.sh <tag>
tag=$1
out=gpurun_out/$tag/profiles
tools/profile_bench.sh $tag --steps 20 --warmup 5 > gpurun_out/${tag}_profile.log 2>&1
python bench.py --steps 20 --warmup 5 --log2n 20 > $out/${tag}_bench_2p20.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-cpu --witness realistic > $out/${tag}_bench_2p22_realistic.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-cpu --precomp 0 > $out/${tag}_bench_2p22_plain.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-cpu --witness-in hbm > $out/${tag}_bench_2p22_resident.json 2>/dev/null
python bench.py --steps 8 --warmup 2 --no-cpu --log2n 24 > $out/${tag}_bench_2p24.json 2>/dev/null
python tools/shard_probe.py 22 1,2,4,8 2>&1 | grep world > $out/${tag}_shard_probe.txt
python tools/cli_timing.py 22 /tmp/zk_cli 2 2>&1 | grep -v amdgpu > $out/${tag}_cli_timing_2p22.txt
for f in $out/*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[1].split("/")[-1], d.get("ms_per_step"), d.get("value"), d.get("resident_witness", d.get("host_witness", {})).get("ms_per_step"), d.get("latency_ms_one_at_a_time"))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
cat $out/${tag}_shard_probe.txt
for k in 14 16 18; do python bench.py --log2n $k --steps 100 --warmup 10 --no-cpu > $out/${tag}_bench_2p$k.json 2>/dev/null; python tools/server_bench.py $k 512 0 2>/dev/null; python tools/server_bench.py $k 512 0,0 2>/dev/null; done | tee $out/${tag}_server_throughput.txt
for k in 14 16 18; do python - "$out/${tag}_bench_2p$k.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[1].split("/")[-1], d.get("ms_per_step"), d.get("value"), d.get("resident_witness", {}).get("ms_per_step"), d.get("latency_ms_one_at_a_time"))
PY
done
