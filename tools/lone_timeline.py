"""Timeline of ONE lone proof from a rocprofv3 --kernel-trace (--memory-copy-trace) run of tools/lone_proof.py:
    python tools/lone_timeline.py <rocprof output dir> [min_us=100] [which=-1]
Clusters of GPU activity separated by more than 30 ms of silence are proofs; prints every kernel / copy of cluster `which`
longer than min_us with start -> end in ms relative to the cluster's first event, per hardware queue, then the wall span,
the time during which NOTHING ran, the time during which only kernels shorter than min_us ran, and the sum of kernel time."""
import csv, glob, sys
d = sys.argv[1]
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
which = int(sys.argv[3]) if len(sys.argv) > 3 else -1
ev = []
for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'q' + r['Queue_Id'], r['Kernel_Name'].replace('void zk::', '').replace('zk::', '')))
for f in glob.glob(d + '/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'copy', r.get('Direction', 'copy')))
ev.sort()
clusters, cur, last_end = [], [], None
for e in ev:
    if last_end is not None and e[0] - last_end > 30e6:
        clusters.append(cur); cur = []
    cur.append(e)
    last_end = e[1] if last_end is None else max(last_end, e[1])
clusters.append(cur)
c = clusters[which]
t0 = c[0][0]; t1 = max(e[1] for e in c)
qs = {}
def short(n):
    n = n.split('(')[0]
    return n[:60] + (' [G2]' if 'Fp2T' in n or 'g2s' in n else '')
for s, e, q, n in c:
    qi = qs.setdefault(q, len(qs))
    if (e - s) / 1e3 >= thr:
        print("%8.3f -> %8.3f (%7.3f ms)  %-6s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q if q == 'copy' else 'Q%d' % qi, short(n)))
# coverage: time with nothing running / only short kernels running
pts = sorted([(s, 1, (e - s) / 1e3 >= thr) for s, e, q, n in c] + [(e, -1, (e - s) / 1e3 >= thr) for s, e, q, n in c])
idle = small = 0; act = big = 0; prev = t0
for t, dlt, isbig in pts:
    if act == 0: idle += t - prev
    elif big == 0: small += t - prev
    prev = t
    act += dlt
    if isbig: big += dlt
print("cluster %d of %d: span %.3f ms; nothing running %.3f ms; only kernels/copies < %.0f us running %.3f ms; sum of kernel+copy time %.3f ms; %d events"
      % (which, len(clusters), (t1 - t0) / 1e6, idle / 1e6, thr, small / 1e6, sum(e - s for s, e, q, n in c) / 1e6, len(c)))
