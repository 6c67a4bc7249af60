"""Is a size launch-bound or issue-bound?  From a rocprofv3 --kernel-trace of `ZK_BENCH_LEG_MARKERS=1 python bench.py --log2n K ...`:
    python tools/busy.py <rocprof dir> <bench stderr> <leg name, e.g. 2p16_headline> <proofs in that leg>
prints, for the leg: launches per proof, the span, the time during which NO kernel ran, the mean number of kernels running,
and per kernel: launches per proof, mean duration, share of the kernel time.  (A kernel that "runs" here may fill four
compute units; the point of the first three numbers is the front end — how long the GPU sat with nothing queued.)"""
import collections, csv, glob, re, sys
d, errfile, want, nproofs = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4])
names = {}
for line in open(errfile, errors="replace"):
    m = re.search(r"leg marker (\d+) \(grid of (\d+) workgroups\): (\S+)", line)
    if m:
        names[int(m.group(2))] = m.group(3)
rows = []
for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
leg, ev = "before_first_marker", []
for r in rows:
    if 'k_mul_vec' in r['Kernel_Name']:
        wg = int(r.get('Workgroup_Size_X') or r.get('Workgroup_Size') or 256)
        g = int(r.get('Grid_Size_X') or r.get('Grid_Size') or 0) // max(1, wg)
        if g in names:
            leg = names[g]
            continue
    if leg == want:
        n = r['Kernel_Name'].replace('void zk::', '').replace('zk::', '').split('(')[0]
        if 'Fp2T' in r['Kernel_Name'] or 'g2s' in n:
            n += ' [G2]'
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), n, int(r.get('Grid_Size_X') or r.get('Grid_Size') or 0) // max(1, int(r.get('Workgroup_Size_X') or r.get('Workgroup_Size') or 256))))
if not ev:
    sys.exit("no kernels in leg %s (legs seen: %s)" % (want, sorted(names.values())))
t0 = ev[0][0]; t1 = max(e[1] for e in ev)
pts = sorted([(s, 1) for s, e, n, g in ev] + [(e, -1) for s, e, n, g in ev])
idle = 0; act = 0; prev = t0; area = 0
for t, dlt in pts:
    if act == 0: idle += t - prev
    area += act * (t - prev)
    prev = t; act += dlt
span = t1 - t0
print("%s: %d kernels = %.1f per proof; span %.3f ms = %.3f ms per proof; nothing running %.1f %% of it; %.2f kernels running on average; kernel time %.3f ms per proof"
      % (want, len(ev), len(ev) / nproofs, span / 1e6, span / 1e6 / nproofs, 100.0 * idle / span, area / span, sum(e - s for s, e, n, g in ev) / 1e6 / nproofs))
per = collections.defaultdict(list); grids = collections.defaultdict(list)
for s, e, n, g in ev:
    per[n].append(e - s); grids[n].append(g)
tot = sum(sum(v) for v in per.values())
for n, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:24]:
    print("  %-46s %5.1f per proof  %8.1f us mean  %5.1f %%   %6d workgroups (median)" % (n[:46], len(v) / nproofs, sum(v) / len(v) / 1e3, 100.0 * sum(v) / tot, sorted(grids[n])[len(v) // 2]))
