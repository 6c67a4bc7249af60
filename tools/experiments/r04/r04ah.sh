#!/bin/bash
# second look at the hardware queues: unset (= 16: the library sets it when the process has not) against 8 / 16 / 24 / 32, two alternations
export TMPDIR=/tmp
o=gpurun_out/r04ah; mkdir -p $o
( for rep in 1 2; do for k in 14 16 18 20; do for hq in unset 8 16 24 32; do
    if [ $hq = unset ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$hq; fi
    python bench.py --log2n $k --steps $([ $k -ge 20 ] && echo 40 || echo 240) --warmup 8 --no-cpu 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read());print('2^$k GPU_MAX_HW_QUEUES=$hq:', d['ms_per_step'], 'ms per proof; resident', d['resident_witness']['ms_per_step'], '; one at a time', d['latency_ms_one_at_a_time']['witness_in_host_memory'])"
  done; done; done
  for rep in 1 2; do for hq in unset 16 32; do
    if [ $hq = unset ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$hq; fi
    python bench.py --steps 16 --warmup 3 --no-cpu 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read());print('2^22 GPU_MAX_HW_QUEUES=$hq:', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'sync', d['ms_per_proof_sync'], 'realistic', d['also_realistic']['ms_per_step'] if 'also_realistic' in d else None)"; done; done
) > $o/hw_queues2.txt 2>&1
cat $o/hw_queues2.txt
