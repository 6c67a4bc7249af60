#!/bin/bash
# A/B of the host-witness upload path (one box): ms/proof host-witness vs resident
run() { echo -n "$1: "; env $2 python bench.py --steps 12 --warmup 3 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['resident_witness']['ms_per_step'], 'spmv', d['stage_ms']['spmv'], 'h2d', d['stage_ms']['wtns_h2d'])"; }
run "default            " "A=1"
run "d2h kernel         " "ZKHIP_D2H_KERNEL=1"
run "h2d kernel         " "ZKHIP_H2D_KERNEL=1"
run "both kernels       " "ZKHIP_D2H_KERNEL=1 ZKHIP_H2D_KERNEL=1"
run "both + sync stage  " "ZKHIP_D2H_KERNEL=1 ZKHIP_H2D_KERNEL=1 ZKHIP_STAGE_SYNC=1"
run "8 hw queues        " "GPU_MAX_HW_QUEUES=8"
run "8 hwq + both       " "GPU_MAX_HW_QUEUES=8 ZKHIP_D2H_KERNEL=1 ZKHIP_H2D_KERNEL=1"
