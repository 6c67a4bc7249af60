"""Per-queue kernel timeline of one steady-state proof period from a rocprofv3 --kernel-trace CSV directory.
    python tools/timeline.py <dir> [min_us=250] [period_index=3]
Prints every kernel longer than min_us between two consecutive k_spmv_abc launches (= one proof period)."""
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
thr=float(sys.argv[2]) if len(sys.argv)>2 else 250
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'k_spmv_abc' in r['Kernel_Name']]
pi=int(sys.argv[3]) if len(sys.argv)>3 else min(3,len(idx)-2)
s=idx[pi]; e=idx[pi+1]
t0=int(rows[s]['Start_Timestamp']); t1=int(rows[e]['Start_Timestamp'])
qs={}
for r in rows:
    st=int(r['Start_Timestamp']); en=int(r['End_Timestamp'])
    if en < t0 or st > t1: continue
    q=qs.setdefault(r['Queue_Id'],len(qs))
    if (en-st)/1e3>thr:
        name=r['Kernel_Name'].replace('void zk::','').replace('zk::','')
        print("%8.3f -> %8.3f (%6.3f) q%d %s" % ((st-t0)/1e6,(en-t0)/1e6,(en-st)/1e6,q,name[:40]+(' G2' if 'Fp2T' in name else '')))
print("period ms", (t1-t0)/1e6)
