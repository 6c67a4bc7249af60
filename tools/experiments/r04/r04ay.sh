#!/bin/bash
# does the 10 % that L2-resident table rows save (r04ac) come with a higher shader clock?  serial proofs, clock sampled every 50 ms by bench.py
export TMPDIR=/tmp
o=gpurun_out/r04ay; mkdir -p $o
export ZKHIP_LIB=$PWD/rapidsnark-old_amd/libzkhip_probes.so
( for rep in 1 2; do for m in 0xffffffff 0xffff; do
    ZKHIP_GATHER_MASK=$m ZKHIP_SERIAL=1 python bench.py --steps 40 --warmup 3 --no-cpu --pipeline 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); ib=d['roofline']['issue_bound']; print('G1 gathers & $m, serial: G1 launch per MSM', d['stage_ms']['g1_l1_kernel'], 'ms; G2', d['stage_ms']['g2_l1_kernel'], '; proof', d['ms_per_step'], '; clock', ib['clock_ghz'], 'GHz over', ib['clock_samples'], 'samples; power', ib['power_w'], 'W')"
  done; done
  python tools/clock_during.py ./tools/mul_rate_probe 2>&1 | tail -5
) > $o/gather_mask_clock.txt 2>&1
cat $o/gather_mask_clock.txt
