#!/bin/bash
# Profiles of the DRIVER'S command for profiles/<tag>_* (run on the GPU box, from the repo root):  tools/profile_bench.sh <tag>
# 1. python bench.py                                        -> <tag>_bench.json            (the line itself, unprofiled)
# 2. rocprofv3 --kernel-trace --stats -- python bench.py    -> <tag>_kernel_stats.csv      (the whole run, rocprofv3's own table)
#    the SAME command, ZK_BENCH_LEG_MARKERS=1 in its environment: a marker launch between bench.py's legs, so the trace is
#    also cut into one table per leg  -> <tag>_kernel_stats_<leg>.csv  (tools/leg_stats.py): 2p22_headline (host witnesses,
#    six in flight: roofline.launch_ms is the mean of k_msm_accum_l1<Fq> HERE), 2p22_other_witness_placement (resident),
#    2p22_lone_resident, 2p22_lone_host_witness (ms_per_proof_sync), 2p22_after (the CPU leg's checker proof), 2p20_*.
# 3./4. separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (TCC slots do not hold both) of `python bench.py --no-cpu`
#    (= the 2^22 legs only; the CPU leg and the 2^20 leg launch no 2^22 kernel) -> <tag>_pmc_traffic.json, per launch
# 5. --pmc SQ_INSTS_VALU pass of the same -> <tag>_valu_instruction_budget.{txt,json}  (bench.py replays the json as
#    roofline.issue_bound.valu_instructions_per_proof, the traffic json as roofline.traffic)
set -u
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out/profiles
export TMPDIR=/tmp
python bench.py "$@" > $out/bench.json 2> $out/bench.err
ZK_BENCH_LEG_MARKERS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python bench.py "$@" > $out/stats.json 2> $out/stats.err
args="--no-cpu"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o f -- python bench.py $args > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o w -- python bench.py $args > $out/write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT64 SQ_WAVES --output-format csv -d $out/valu -o v -- python bench.py $args > $out/valu.log 2>&1
sd=$(dirname $(find $out/stats -name '*kernel_stats.csv' | head -1))
fd=$(dirname $(find $out/fetch -name '*counter_collection.csv' | head -1))
wd=$(dirname $(find $out/write -name '*counter_collection.csv' | head -1))
python tools/summarize_profiles.py $tag $sd $fd $wd $out/bench.json && cp profiles/${tag}_* $out/profiles/
python tools/leg_stats.py $out/stats $out/stats.err $out/profiles/${tag}_kernel_stats > $out/profiles/${tag}_legs.txt 2>&1
cp $out/bench.json $out/profiles/${tag}_bench.json
grep '^{' $out/stats.json | tail -1 > $out/profiles/${tag}_bench_under_kernel_trace.json
python tools/instr_budget.py $out/valu 0 $out/bench.json $out/profiles/${tag}_valu_instruction_budget.json > $out/profiles/${tag}_valu_instruction_budget.txt 2>&1 || true
find $out -name '*kernel_trace.csv' -size +30M -delete; find $out -name '*counter_collection.csv' -size +30M -delete
ls -la $out/profiles
