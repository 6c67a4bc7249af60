#!/bin/bash
export TMPDIR=/tmp
o=gpurun_out/r04g; mkdir -p $o
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1: host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'sync', d['ms_per_proof_sync'], 'clock', d['roofline'].get('issue_bound',{}).get('clock_ghz'))"; }
( for rep in 1 2 3; do
    python bench.py --no-cpu --steps 30 --warmup 3 2>/dev/null | line "new lib, clock sampler on "
    ZK_BENCH_CLOCK=0 python bench.py --no-cpu --steps 30 --warmup 3 2>/dev/null | line "new lib, clock sampler OFF"
    ZKHIP_LIB=$PWD/tools/_ab/libzkhip_old.so ZK_BENCH_CLOCK=0 python bench.py --no-cpu --steps 30 --warmup 3 2>/dev/null | line "round-3 lib, sampler OFF  "
  done ) > $o/ab_headline.txt 2>&1
cat $o/ab_headline.txt
( for k in 21 22; do for b in 0 1; do
    ZKHIP_LIB=$PWD/rapidsnark-old_amd/libzkhip_probes.so ZKHIP_BATCH_ABC=$b python bench.py --log2n $k --steps 20 --warmup 3 --no-cpu 2>/dev/null | line "unsharded 2^$k ZKHIP_BATCH_ABC=$b"
  done; done ) > $o/batch_abc_large.txt 2>&1
cat $o/batch_abc_large.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $o/pytest_gpu.log 2>&1; tail -4 $o/pytest_gpu.log
