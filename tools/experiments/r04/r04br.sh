#!/bin/bash
# mixed-schedule soak of the final library at the small sizes (the lone-proof and batch paths changed late in the round)
export TMPDIR=/tmp
o=gpurun_out/r04br; mkdir -p $o
( for k in 14 16 18 20; do timeout 300 python tools/soak_mixed.py $k 40 $k 2>&1 | grep -v amdgpu; echo "exit code $?"; done ) > $o/soak_mixed.txt 2>&1
cat $o/soak_mixed.txt
