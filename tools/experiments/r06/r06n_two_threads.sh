# a second enqueueing thread for the witness MSMs of a lone proof (ZKHIP_TWO_THREADS=0 off / 1 on at every size) — same box, three alternations + the GPU suite
mkdir -p gpurun_out/r06n
(timeout 1000 python -m pytest tests -m gpu -x -q 2>&1 | tail -4) > gpurun_out/r06n/gputest.txt
cat gpurun_out/r06n/gputest.txt
for rep in 1 2 3; do
for v in 0 1; do
  for k in 14 16 18 20; do
  ZKHIP_TWO_THREADS=$v python bench.py --steps 64 --warmup 8 --no-cpu --log2n $k 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('two_threads=$v 2^$k: period host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], '| one at a time: resident', d['latency_ms_one_at_a_time']['witness_in_hbm'], 'SYNC host witness', d['ms_per_proof_sync'])" >> gpurun_out/r06n/ab.txt
  done
done
done
for rep in 1 2; do
for v in 0 1; do
  ZKHIP_TWO_THREADS=$v python bench.py --steps 12 --warmup 3 --no-cpu --log2n 22 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('two_threads=$v 2^22: period host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], '| one at a time: resident', d['latency_ms_one_at_a_time']['witness_in_hbm'], 'SYNC host witness', d['ms_per_proof_sync'])" >> gpurun_out/r06n/ab.txt
done
done
sort -s -k2,2 gpurun_out/r06n/ab.txt
