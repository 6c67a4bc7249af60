# MSM H's merges + bucket reduction of a lone 2^20 proof on the (idle) upload stream: probes build, ZKHIP_H_ASIDE_MAXLOG=0 (off) against 20
mkdir -p gpurun_out/r06j
export ZKHIP_LIB=$PWD/rapidsnark-old_amd/libzkhip_probes.so
for rep in 1 2 3; do
for ml in 0 20; do
  ZKHIP_H_ASIDE_MAXLOG=$ml python bench.py --steps 16 --warmup 4 --no-cpu --log2n 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('h_aside_maxlog=$ml 2^20: period host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], '| one at a time: resident', d['latency_ms_one_at_a_time']['witness_in_hbm'], 'SYNC host witness', d['ms_per_proof_sync'])" >> gpurun_out/r06j/ab.txt
done
done
unset ZKHIP_LIB
(timeout 600 python -m pytest tests -m gpu -x -q -k "2p20 or synth or pipeline or golden or cli" 2>&1 | tail -3) >> gpurun_out/r06j/ab.txt
cat gpurun_out/r06j/ab.txt
