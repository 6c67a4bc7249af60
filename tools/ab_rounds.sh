#!/bin/bash
# Level-1 accumulation: whole rounds of lanes, by chunk ceiling (GPU box, repo root).  tools/ab_rounds.sh [log2n...]
pick='import json,sys; d=json.loads(sys.stdin.readlines()[-1]); r=d["roofline"]; print("   ms/proof %.2f   g1_l1 %.3f ms  g2_l1 %.3f ms   one-at-a-time %s" % (d["ms_per_step"], r["launch_ms"], r["also"]["launch_ms"], d.get("latency_ms_one_at_a_time")))'
for k in ${@:-22 20}; do
  for cm in 128 160 320; do
    echo "2^$k ZKHIP_ACC_CHUNK_MAX=$cm"
    ZKHIP_ACC_CHUNK_MAX=$cm python bench.py --no-cpu --log2n $k 2>/dev/null | python -c "$pick"
  done
done
