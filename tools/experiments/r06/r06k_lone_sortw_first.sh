# a lone proof's chain held back until sort(w) is done (ZKHIP_LONE_SORTW_FIRST=1, probes build) — same box, three alternations
mkdir -p gpurun_out/r06k
export ZKHIP_LIB=$PWD/rapidsnark-old_amd/libzkhip_probes.so
for rep in 1 2 3; do
for v in 0 1; do
  if [ $v = 1 ]; then export ZKHIP_LONE_SORTW_FIRST=1; else unset ZKHIP_LONE_SORTW_FIRST; fi
  for k in 22 21; do
  python bench.py --steps 12 --warmup 3 --no-cpu --log2n $k 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sortw_first=$v 2^$k: period host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], '| one at a time: resident', d['latency_ms_one_at_a_time']['witness_in_hbm'], 'SYNC host witness', d['ms_per_proof_sync'])" >> gpurun_out/r06k/ab.txt
  done
done
done
cat gpurun_out/r06k/ab.txt
