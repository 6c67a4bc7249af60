#!/bin/bash
export TMPDIR=/tmp
o=gpurun_out/r04i; mkdir -p $o
( for rep in 1 2 3; do for b in 0 1; do for k in 22; do
    ZKHIP_BATCH_ABC=$b python tools/lone_proof.py $k 10 2>/dev/null | awk -v b=$b -v k=$k '/lone proof/ {s+=$4; n++} END {printf "ZKHIP_BATCH_ABC=%d 2^%d: %.2f ms per synchronous proof (mean of %d, 0.1 s pauses)\n", b, k, s/n, n}'
    ZKHIP_BATCH_ABC=$b python bench.py --log2n $k --steps 12 --warmup 3 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ZKHIP_BATCH_ABC=$b 2^$k bench: host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'sync (8 back to back)', d['ms_per_proof_sync'])"
  done; done; done ) > $o/ab_batch_abc_sync.txt 2>&1
cat $o/ab_batch_abc_sync.txt
ZKHIP_BATCH_ABC=1 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $o/lone22 -o t -- python tools/lone_proof.py 22 4 > $o/lone22.log 2>&1
python tools/lone_timeline.py $o/lone22 100 > $o/lone22_batch_abc_timeline.txt 2>&1; grep "^lone" $o/lone22.log >> $o/lone22_batch_abc_timeline.txt
find $o/lone22 -name '*.csv' -size +20M -delete
tail -32 $o/lone22_batch_abc_timeline.txt
