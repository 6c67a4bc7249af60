#!/bin/bash
# A/B of the host-witness upload path (one box): ms/proof host-witness vs resident.  (Kernel copies instead of
# hipMemcpyAsync for the upload / the window-sum download were tried with these runs and were slower: 39.0 vs 36.9 ms.)
run() { echo -n "$1: "; env $2 python bench.py --steps 12 --warmup 3 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['resident_witness']['ms_per_step'], 'spmv', d['stage_ms']['spmv'], 'h2d', d['stage_ms']['wtns_h2d'])"; }
run "default (16 hw queues)" "A=1"
run "staging inside submit " "ZKHIP_STAGE_SYNC=1"
run "4 hw queues           " "GPU_MAX_HW_QUEUES=4"
run "8 hw queues           " "GPU_MAX_HW_QUEUES=8"
run "24 hw queues          " "GPU_MAX_HW_QUEUES=24"
