R=$GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_prove.py tests/test_gpu_synth.py -x -q -m gpu -k "precomp" 2>&1 | tail -2
cd /tmp; export TMPDIR=/tmp
for cfg in "12 32" "11 32" "11 16"; do
  set -- $cfg
  rm -rf /tmp/pp; ZKHIP_SERIAL=1 ZKHIP_BIN_SHIFT=$1 ZKHIP_BIN_SLICES=$2 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o x -- python $R/bench.py --steps 3 --warmup 1 --no-cpu > /dev/null 2>&1
  echo "shift $1 slices $2"; python - <<'PY'
import csv,glob
f=glob.glob('/tmp/pp/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'k_bin' in r['Name']:
        print("   %-50s calls %4s avg_us %9.1f" % (r['Name'][:50], r['Calls'], float(r['AverageNs'])/1e3))
PY
  cd $R; ZKHIP_BIN_SHIFT=$1 ZKHIP_BIN_SLICES=$2 python bench.py --steps 8 --warmup 2 --no-cpu 2>&1 | tail -1 | cut -c1-120; cd /tmp
done
