# ZK_FLAG_PRECOMP_HALF: parity subset, the three table modes side by side at 2^20 / 2^22 / 2^24 / 2^25, same-box A/B of the driver-style period
mkdir -p gpurun_out/r06e
(timeout 900 python -m pytest tests -m gpu -x -q -k "precomp or window or infinity or irregular or circuit_shaped or beside or all_signals or multi_prover_equals or dlog" 2>&1 | tail -5) > gpurun_out/r06e/gputest.txt
cat gpurun_out/r06e/gputest.txt
free -g | head -2 > gpurun_out/r06e/modes.txt
for k in 20 22 24; do timeout 600 python tools/half_tables.py $k >> gpurun_out/r06e/modes.txt 2>&1; done
ram=$(free -g | awk '/Mem:/{print $7}')
if [ "$ram" -ge 90 ]; then timeout 900 python tools/half_tables.py 25 1,2,0 >> gpurun_out/r06e/modes.txt 2>&1; else echo "2^25 skipped: $ram GB of host memory available" >> gpurun_out/r06e/modes.txt; fi
grep -v "amdgpu.ids" gpurun_out/r06e/modes.txt
for rep in 1 2; do for pc in 1 2; do
  python bench.py --steps 16 --warmup 4 --no-cpu --log2n 22 --precomp $pc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('precomp=$pc 2^22: period host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], '| one at a time: resident', d['latency_ms_one_at_a_time']['witness_in_hbm'], 'SYNC host witness', d['ms_per_proof_sync'], 'create s', d['setup_s'])" >> gpurun_out/r06e/ab.txt
done; done
cat gpurun_out/r06e/ab.txt
