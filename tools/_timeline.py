import csv,glob,sys
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
thr=float(sys.argv[2]) if len(sys.argv)>2 else 150
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'k_spmv_abc' in r['Kernel_Name']]
# proof boundaries: first k_msm_digits before each spmv
def start_of(i):
    j=i
    while j>0 and int(rows[i]['Start_Timestamp'])-int(rows[j-1]['Start_Timestamp'])<3e6 and 'reduce' not in rows[j-1]['Kernel_Name'] and 'copyBuffer' not in rows[j-1]['Kernel_Name']: j-=1
    return j
s=start_of(idx[-2]); e=start_of(idx[-1])
t0=int(rows[s]['Start_Timestamp'])
qs={}
for r in rows[s:e]:
    st=int(r['Start_Timestamp']); en=int(r['End_Timestamp'])
    q=qs.setdefault(r['Queue_Id'],len(qs))
    if (en-st)/1e3>thr:
        name=r['Kernel_Name'].replace('void zk::','').replace('zk::','')
        print("%8.3f -> %8.3f (%6.3f) q%d %s" % ((st-t0)/1e6,(en-t0)/1e6,(en-st)/1e6,q,name[:46]+(' G2' if 'Fp2T' in name else '')))
print("proof span ms", (int(rows[e-1]['End_Timestamp'])-t0)/1e6, " next proof starts", (int(rows[e]['Start_Timestamp'])-t0)/1e6)
