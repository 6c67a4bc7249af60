#!/bin/bash
# Profiles of the DRIVER'S command for profiles/<tag>_* (run on the GPU box, from the repo root):  tools/profile_bench.sh <tag> [bench args]
# 1. python bench.py                                        -> <tag>_bench.json   (the line itself, unprofiled; its counters are measured
#    IN the run by the two rocprofv3 --pmc passes bench.py wraps around a child of itself: rapidsnark_old_amd/counters.py) and
#    <tag>_counters.json (the per-leg counter summaries of that run: what a later run replays when it cannot measure)
# 2. rocprofv3 --kernel-trace --stats -- python bench.py    -> <tag>_kernel_stats.csv   (the whole run, rocprofv3's own table); with
#    ZK_BENCH_LEG_MARKERS=1 a marker launch separates bench.py's legs, so the trace is also cut into one table per leg
#    -> <tag>_kernel_stats_<leg>.csv  (tools/leg_stats.py): 2p22_headline (host witnesses, six in flight: roofline.launch_ms is
#    the mean of k_msm_accum_l1<Fq> per MSM HERE), 2p22_other_witness_placement, 2p22_lone_resident (launch_ms_one_in_flight),
#    2p22_lone_host_witness (ms_per_proof_sync), 2p22_after (the CPU leg's checker proof), 2p22_circuit_*, 2p20_*.
#    (under the tracer bench.py does not start its own counter passes: that line replays <tag>_counters.json)
set -u
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out/profiles
export TMPDIR=/tmp
ZK_BENCH_SAVE_COUNTERS=$tag python bench.py "$@" > $out/bench.json 2> $out/bench.err
ZK_BENCH_LEG_MARKERS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python bench.py --no-server "$@" > $out/stats.json 2> $out/stats.err
cp "$(find $out/stats -name '*kernel_stats.csv' | head -1)" $out/profiles/${tag}_kernel_stats.csv
python tools/leg_stats.py $out/stats $out/stats.err $out/profiles/${tag}_kernel_stats > $out/profiles/${tag}_legs.txt 2>&1
cp $out/bench.json $out/profiles/${tag}_bench.json
cp profiles/${tag}_counters.json $out/profiles/ 2>/dev/null
grep '^{' $out/stats.json | tail -1 > $out/profiles/${tag}_bench_under_kernel_trace.json
find $out -name '*kernel_trace.csv' -size +30M -delete
ls -la $out/profiles
