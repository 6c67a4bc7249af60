#!/bin/bash
# Follow-up streams (ZKHIP_TAIL: 2 = two high-priority streams for the partial merges and bucket reductions, 0 = none, the
# default on an unsharded prover since round 3), alternating on one box, several sizes: pipelined period with host and with
# resident witnesses, one synchronous zk_prove, one at a time with a resident witness.
#   tools/ab_tailstreams.sh [log2n ...]        (profiles/r03z_ab_follow_up_streams.txt)
pick='import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print("   host %.2f  resident %.2f  sync %s  one-at-a-time resident %s" % (d["ms_per_step"], d["resident_witness"]["ms_per_step"], d.get("ms_per_proof_sync"), d["latency_ms_one_at_a_time"]["witness_in_hbm"]))'
for k in ${@:-22 20 18 16}; do
  for rep in 1 2; do
    for t in 2 0; do
      echo -n "2^$k ZKHIP_TAIL=$t"
      st=20; [ $k -le 18 ] && st=100
      ZKHIP_TAIL=$t python bench.py --no-cpu --log2n $k --steps $st --warmup 4 2>/dev/null | python -c "$pick"
    done
  done
done
