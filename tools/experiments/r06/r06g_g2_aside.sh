# the merges + bucket reduction of MSM B2 of a LONE proof on the idle finishing stream also at 2^20 / 2^21 (today: up to 2^19) — probes build, same box
mkdir -p gpurun_out/r06g
export ZKHIP_LIB=$PWD/rapidsnark-old_amd/libzkhip_probes.so
for rep in 1 2 3; do
for ml in 19 21; do
  for k in 20 21; do
  ZKHIP_G2_ASIDE_MAXLOG=$ml python bench.py --steps 16 --warmup 4 --no-cpu --log2n $k 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('g2_aside_maxlog=$ml 2^$k: period host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], '| one at a time: resident', d['latency_ms_one_at_a_time']['witness_in_hbm'], 'SYNC host witness', d['ms_per_proof_sync'])" >> gpurun_out/r06g/ab.txt
  done
done
done
cat gpurun_out/r06g/ab.txt
