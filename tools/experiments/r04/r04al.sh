#!/bin/bash
# timeline of ONE lone proof at 2^14 and 2^16 (every kernel, no threshold)
export TMPDIR=/tmp
o=$PWD/gpurun_out/r04al; mkdir -p $o
for k in 14 16; do
  ( cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/lone$k -o t -- python $OLDPWD/tools/lone_proof.py $k 4 > $o/lone$k.log 2>&1 )
  python tools/lone_timeline.py /tmp/lone$k 0 -1 > $o/lone_proof_timeline_2p$k.txt; grep "lone proof" $o/lone$k.log >> $o/lone_proof_timeline_2p$k.txt
  python tools/lone_proof.py $k 6 2>/dev/null | grep "lone proof" | sed 's/^/without the tracer: /' >> $o/lone_proof_timeline_2p$k.txt
done
cat $o/lone_proof_timeline_2p14.txt
