# proof k+1's bucket memsets + sort(w) on the upload stream instead of behind proof k's witness MSMs on stream 2 (provers with one lane) — same box
mkdir -p gpurun_out/r06x
for rep in 1 2 3; do
for v in 0 1; do
  ZKHIP_SORTW_ASIDE=$v python tools/shard_lone.py 22 8 60 2 2>&1 | grep "share of" | sed "s/^/sortw_aside=$v /" >> gpurun_out/r06x/ab.txt
  ZKHIP_SORTW_ASIDE=$v python tools/shard_lone.py 24 8 16 2 2>&1 | grep "share of" | sed "s/^/sortw_aside=$v /" >> gpurun_out/r06x/ab.txt
done
done
for rep in 1 2; do
for v in 0 1; do
  ZKHIP_SORTW_ASIDE=$v python bench.py --steps 6 --warmup 2 --no-cpu --log2n 24 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sortw_aside=$v 2^24: period host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], '| one at a time: resident', d['latency_ms_one_at_a_time']['witness_in_hbm'], 'SYNC host witness', d['ms_per_proof_sync'])" >> gpurun_out/r06x/ab.txt
done
done
sort -s -k1,1 gpurun_out/r06x/ab.txt
