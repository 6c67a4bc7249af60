#!/bin/bash
# wave priorities as the default: full GPU suite, shards and the REST server against the library without them
export TMPDIR=/tmp
o=gpurun_out/r04bi; mkdir -p $o
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
  for rep in 1 2; do for l in noprio default; do
    f=$PWD/rapidsnark-old_amd/libzkhip.so; [ $l = noprio ] && f=$PWD/rapidsnark-old_amd/libzkhip_noprio.so
    ZKHIP_LIB=$f python tools/shard_probe.py 22 2,4,8 partitioned 2>/dev/null | grep world | cut -c1-100 | sed "s/^/$l /"
    ZKHIP_LIB=$f python bench.py --log2n 16 --batch 8 --steps 400 --warmup 8 --no-cpu 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read());print('$l 2^16 x 8 per submission: period', d['ms_per_step'])"
    ZKHIP_LIB=$f python bench.py --log2n 14 --batch 8 --steps 400 --warmup 8 --no-cpu 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read());print('$l 2^14 x 8 per submission: period', d['ms_per_step'])"
    ZKHIP_LIB=$f python bench.py --steps 16 --warmup 3 --no-cpu --no-2p20 --shape circuit --witness realistic 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read());print('$l 2^22 circuit-shaped, realistic witness: period', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'sync', d['ms_per_proof_sync'])"
    ZKHIP_LIB=$f python bench.py --log2n 24 --steps 6 --warmup 2 --no-cpu 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read());print('$l 2^24: period', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'])"
  done; done
  timeout 200 python tools/soak_mixed.py 16 30 7 2>&1 | grep -v amdgpu
) > $o/prio_default_checks.txt 2>&1
cat $o/prio_default_checks.txt
