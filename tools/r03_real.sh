#!/bin/bash
ZKHIP_SERIAL=1 python bench.py --steps 6 --warmup 2 --no-cpu --witness realistic 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('serial realistic: ', d['ms_per_step'], d['stage_ms'])"
python bench.py --steps 20 --warmup 3 --no-cpu --witness realistic 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('realistic: host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'latency', d['latency_ms_one_at_a_time'], d['stage_ms'])"
export TMPDIR=/tmp
ZKHIP_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03l/st -o s -- python bench.py --steps 4 --warmup 1 --in-flight 1 --no-cpu --witness realistic > gpurun_out/r03l.log 2>&1
python - <<'PY'
import csv, glob
rows=[]
for f in glob.glob("gpurun_out/r03l/st/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["TotalDurationNs"]), r["Name"][:90], r["Calls"], r["AverageNs"]))
for t,n,c,a in sorted(rows, reverse=True)[:28]:
    print("%9.3f ms total  calls %5s  avg %10.1f us  %s" % (t/1e6, c, float(a)/1e3, n))
PY
