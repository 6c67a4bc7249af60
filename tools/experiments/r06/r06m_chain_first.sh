# host order of a lone proof's launches: A.w/B.w + transforms enqueued before the witness MSMs (ZKHIP_CHAIN_FIRST_MAXLOG, probes build: 0 = off) — same box
mkdir -p gpurun_out/r06m
export ZKHIP_LIB=$PWD/rapidsnark-old_amd/libzkhip_probes.so
for rep in 1 2 3; do
for ml in 0 24; do
  for k in 14 16 18 20; do
  ZKHIP_CHAIN_FIRST_MAXLOG=$ml python bench.py --steps 64 --warmup 8 --no-cpu --log2n $k 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chain_first_maxlog=$ml 2^$k: period host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], '| one at a time: resident', d['latency_ms_one_at_a_time']['witness_in_hbm'], 'SYNC host witness', d['ms_per_proof_sync'])" >> gpurun_out/r06m/ab.txt
  done
done
done
for rep in 1 2; do
for ml in 0 24; do
  ZKHIP_CHAIN_FIRST_MAXLOG=$ml python bench.py --steps 12 --warmup 3 --no-cpu --log2n 22 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chain_first_maxlog=$ml 2^22: period host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], '| one at a time: resident', d['latency_ms_one_at_a_time']['witness_in_hbm'], 'SYNC host witness', d['ms_per_proof_sync'])" >> gpurun_out/r06m/ab.txt
done
done
sort -s -k2,2 gpurun_out/r06m/ab.txt
