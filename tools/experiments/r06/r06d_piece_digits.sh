# same-box A/B: the digits / first-level bin counts of sort(w) taken piece by piece behind the witness upload (ZKHIP_PIECE_DIGITS=1, default) or not (=0)
mkdir -p gpurun_out/r06d
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4) > gpurun_out/r06d/gputest.txt
for rep in 1 2 3; do
for pd in 1 0; do
  for k in 22 20; do
  ZKHIP_PIECE_DIGITS=$pd python bench.py --steps 12 --warmup 3 --no-cpu --log2n $k 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('piece_digits=$pd 2^$k: period host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], '| one at a time: resident', d['latency_ms_one_at_a_time']['witness_in_hbm'], 'SYNC host witness', d['ms_per_proof_sync'])" >> gpurun_out/r06d/ab.txt
  done
done
done
cat gpurun_out/r06d/gputest.txt gpurun_out/r06d/ab.txt
