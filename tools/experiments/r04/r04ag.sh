#!/bin/bash
# hardware queues: HIP maps a process's streams onto GPU_MAX_HW_QUEUES (default 4) queues; a small circuit's kernels fill a
# fraction of the chip each, so the number of kernels that can run side by side is what bounds it
export TMPDIR=/tmp
o=gpurun_out/r04ag; mkdir -p $o
( for k in 14 16 18 20; do for b in 1 4; do [ $k -ge 18 ] && [ $b = 4 ] && continue; for hq in 4 8 16; do
    GPU_MAX_HW_QUEUES=$hq python bench.py --log2n $k --steps $([ $k -ge 20 ] && echo 40 || echo 240) --warmup 8 --batch $b --no-cpu 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read());print('2^$k, $b per submission, GPU_MAX_HW_QUEUES=$hq:', d['ms_per_step'], 'ms per proof; one at a time', d['latency_ms_one_at_a_time']['witness_in_host_memory'])"
  done; done; done
  for hq in 4 8; do GPU_MAX_HW_QUEUES=$hq python bench.py --steps 16 --warmup 3 --no-cpu 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read());print('2^22 GPU_MAX_HW_QUEUES=$hq:', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'sync', d['ms_per_proof_sync'])"; done
) > $o/hw_queues.txt 2>&1
cat $o/hw_queues.txt
