"""One kernel table PER LEG of a bench.py run from a rocprofv3 --kernel-trace of the driver's own command
(ZK_BENCH_LEG_MARKERS=1 python bench.py):
    python tools/leg_stats.py <rocprof dir> <bench stderr with the '[bench] leg marker' lines> <out prefix>
bench.py separates its legs (headline: host witnesses, six in flight | the other witness placement | one proof at a time,
resident | one at a time, host witness = the synchronous zk_prove | after: the CPU leg's checker proof) by a k_mul_vec<Fr>
launch whose grid is 16 + leg workgroups.  Writes <out prefix>_<leg>.csv with the columns of rocprofv3's own kernel_stats
(Name, Calls, TotalDurationNs, AverageNs, Percentage, MinNs, MaxNs, StdDev), so every per-launch figure of the bench line
can be recomputed from the one file of its leg."""
import collections, csv, glob, math, re, sys

d, errfile, prefix = sys.argv[1], sys.argv[2], sys.argv[3]
names = {}
for line in open(errfile, errors="replace"):
    m = re.search(r"leg marker (\d+) \(grid of (\d+) workgroups\): (\S+)", line)
    if m:
        names[int(m.group(2))] = m.group(3)
rows = []
for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
leg = "before_first_marker"
per = collections.OrderedDict()
for r in rows:
    if 'k_mul_vec' in r['Kernel_Name']:
        wg = int(r.get('Workgroup_Size_X') or r.get('Workgroup_Size') or 256)
        g = int(r.get('Grid_Size_X') or r.get('Grid_Size') or 0) // max(1, wg)
        if g in names:
            leg = names[g]
            continue
    per.setdefault(leg, collections.defaultdict(list))[r['Kernel_Name']].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for leg, ks in per.items():
    tot = sum(sum(v) for v in ks.values())
    with open("%s_%s.csv" % (prefix, leg), "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for k, v in sorted(ks.items(), key=lambda kv: -sum(kv[1])):
            mean = sum(v) / len(v)
            sd = math.sqrt(sum((x - mean) ** 2 for x in v) / len(v))
            w.writerow([k, len(v), sum(v), round(mean, 3), round(100.0 * sum(v) / tot, 4), min(v), max(v), round(sd, 3)])
    print("%-28s %5d kernels, %10.3f ms of kernel time" % (leg, sum(len(v) for v in ks.values()), tot / 1e6))
