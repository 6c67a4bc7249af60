# the same at 2^22 / 2^24 (round 4 measured a loss at 2^22, before the wave priorities and the split reduction) — probes build, same box
mkdir -p gpurun_out/r06h
export ZKHIP_LIB=$PWD/rapidsnark-old_amd/libzkhip_probes.so
for rep in 1 2 3; do
for ml in 21 24; do
  ZKHIP_G2_ASIDE_MAXLOG=$ml python bench.py --steps 12 --warmup 3 --no-cpu --log2n 22 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('g2_aside_maxlog=$ml 2^22: period host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], '| one at a time: resident', d['latency_ms_one_at_a_time']['witness_in_hbm'], 'SYNC host witness', d['ms_per_proof_sync'])" >> gpurun_out/r06h/ab.txt
done
done
for rep in 1 2; do
for ml in 21 24; do
  ZKHIP_G2_ASIDE_MAXLOG=$ml python bench.py --steps 6 --warmup 2 --no-cpu --log2n 24 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('g2_aside_maxlog=$ml 2^24: period host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], '| one at a time: resident', d['latency_ms_one_at_a_time']['witness_in_hbm'], 'SYNC host witness', d['ms_per_proof_sync'])" >> gpurun_out/r06h/ab.txt
done
done
cat gpurun_out/r06h/ab.txt
