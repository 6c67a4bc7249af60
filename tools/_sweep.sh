python -m pytest tests/test_gpu_prove.py tests/test_gpu_synth.py -x -q -m gpu 2>&1 | tail -3
for pl in 1 0 1; do python bench.py --steps 10 --warmup 2 --no-cpu --pipeline $pl 2>&1 | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("pipeline", d["config"]["proofs_in_flight"], d["ms_per_step"], d["value"], d.get("latency_ms_one_at_a_time"), d["stage_ms"])'; done
python bench.py --log2n 20 --steps 20 --warmup 2 --no-cpu 2>&1 | tail -1 | cut -c1-140
python bench.py --log2n 20 --steps 20 --warmup 2 --no-cpu --pipeline 0 2>&1 | tail -1 | cut -c1-140
