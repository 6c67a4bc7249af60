#!/bin/bash
# same-box A/B of the lone-proof launch order (probes build: ZKHIP_LONE_ORDER=0/1) on the synchronous zk_prove, and of the
# tree's library against tools/_ab/libzkhip_old.so
export ZKHIP_LIB=$PWD/rapidsnark-old_amd/libzkhip_probes.so
for rep in 1 2 3; do
  for lo in 1 0; do
    for k in 22 20; do
      ZKHIP_LONE_ORDER=$lo python tools/lone_proof.py $k 6 2>/dev/null | awk -v lo=$lo -v k=$k '/lone proof/ {s+=$4; n++} END {printf "probes lib, lone order %d, 2^%d: %.2f ms per synchronous proof (mean of %d)\n", lo, k, s/n, n}'
    done
  done
done
for rep in 1 2; do
  for which in new old; do
    if [ $which = old ]; then export ZKHIP_LIB=$PWD/tools/_ab/libzkhip_old.so; else unset ZKHIP_LIB; fi
    for k in 22 20; do
      python tools/lone_proof.py $k 6 2>/dev/null | awk -v w=$which -v k=$k '/lone proof/ {s+=$4; n++} END {printf "%s lib 2^%d: %.2f ms per synchronous proof (mean of %d)\n", w, k, s/n, n}'
    done
  done
done
