#!/bin/bash
export TMPDIR=/tmp
o=gpurun_out/r04r; mkdir -p $o
python -m pytest tests/test_gpu_field_ntt.py tests/test_gpu_synth.py tests/test_gpu_prove.py -m gpu -x -q > $o/pytest_first.log 2>&1; tail -3 $o/pytest_first.log
bash tools/ntt_counters.sh r04r_ntt 22 > $o/ntt_counters.log 2>&1; cp gpurun_out/r04r_ntt/summary.txt $o/ntt_pipeline_counters.txt; head -12 $o/ntt_pipeline_counters.txt; grep -E "^void|^zk|SQ_INSTS_VALU" $o/ntt_pipeline_counters.txt | tail -24
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1: host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'one at a time resident', d['latency_ms_one_at_a_time']['witness_in_hbm'], 'sync', d['ms_per_proof_sync'], 'spmv', d['stage_ms']['spmv'])"; }
( export ZKHIP_LIB=$PWD/rapidsnark-old_amd/libzkhip_probes.so
  for rep in 1 2 3; do for k in 22 20; do
    python bench.py --log2n $k --steps 30 --warmup 3 --no-cpu 2>/dev/null | line "2^$k a|b|c as lazy limbs (default)"
    ZKHIP_ABC_WORDS=1 python bench.py --log2n $k --steps 30 --warmup 3 --no-cpu 2>/dev/null | line "2^$k a|b|c as canonical words     "
  done; done
  for k in 16 24; do
    python bench.py --log2n $k --steps 8 --warmup 2 --no-cpu 2>/dev/null | line "2^$k limbs"
    ZKHIP_ABC_WORDS=1 python bench.py --log2n $k --steps 8 --warmup 2 --no-cpu 2>/dev/null | line "2^$k words"
  done ) > $o/ab_abc_storage.txt 2>&1
cat $o/ab_abc_storage.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $o/pytest_gpu.log 2>&1; tail -4 $o/pytest_gpu.log
