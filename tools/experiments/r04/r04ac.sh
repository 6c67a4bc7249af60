#!/bin/bash
export TMPDIR=/tmp
o=gpurun_out/r04ac; mkdir -p $o
export ZKHIP_LIB=$PWD/rapidsnark-old_amd/libzkhip_probes.so
( for m in 0xffffffff 0xffff 0xff 0xfffff; do
    ZKHIP_GATHER_MASK=$m ZKHIP_SERIAL=1 python bench.py --steps 6 --warmup 2 --no-cpu --pipeline 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('G1 table gathers confined to rows & $m (serial, one proof at a time; WRONG sums by design): G1 launch per MSM', d['stage_ms']['g1_l1_kernel'], 'ms; G2 (not masked)', d['stage_ms']['g2_l1_kernel'], '; proof', d['ms_per_step'])"
  done
  for m in 0xffffffff 0xffff; do
    ZKHIP_GATHER_MASK=$m python bench.py --steps 16 --warmup 3 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mask $m pipelined: period', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'G1 launch per MSM', d['stage_ms']['g1_l1_kernel'])"
  done ) > $o/gather_mask.txt 2>&1
cat $o/gather_mask.txt
