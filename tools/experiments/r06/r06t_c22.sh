# a 22-bit window (12 additions per point, 2^21 buckets per set) against the 20-bit one at 2^24, where the bucket reductions are 0.7 % of a proof — same box
mkdir -p gpurun_out/r06t
rm -f gpurun_out/r06t/ab.txt
for rep in 1 2; do
for wb in 20 22; do
  python bench.py --steps 6 --warmup 2 --no-cpu --log2n 24 --window-bits $wb 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('window_bits=$wb 2^24: period host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], '| one at a time: resident', d['latency_ms_one_at_a_time']['witness_in_hbm'], 'SYNC host witness', d['ms_per_proof_sync'], 'windows', d['config']['windows'])" >> gpurun_out/r06t/ab.txt 2>&1
done
done
for wb in 20 22; do
  python bench.py --steps 12 --warmup 3 --no-cpu --log2n 22 --window-bits $wb 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('window_bits=$wb 2^22: period host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], '| one at a time: resident', d['latency_ms_one_at_a_time']['witness_in_hbm'], 'windows', d['config']['windows'])" >> gpurun_out/r06t/ab.txt 2>&1
done
cat gpurun_out/r06t/ab.txt
