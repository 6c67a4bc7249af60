#!/bin/bash
# Profiles of the default bench command for profiles/<tag>_* (run on the GPU box, from the repo root):
#   tools/profile_bench.sh <tag> [bench args]
# 1. plain bench run -> <tag>_bench.json
# 2. rocprofv3 --kernel-trace --stats of the same command -> kernel_stats.csv
# 3./4. separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (TCC slots do not hold both)
# 5. VALU instruction budget pass
# then tools/summarize_profiles.py condenses them under profiles/ (copied back through gpurun_out/).
set -u
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python bench.py "$@" > $out/bench.json 2> $out/bench.err
args="--steps 6 --warmup 2 --no-cpu"   # (tools/instr_budget.py counts the proofs itself: one k_spmv_abc launch each)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python bench.py $args > $out/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o f -- python bench.py $args > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o w -- python bench.py $args > $out/write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT64 SQ_WAVES --output-format csv -d $out/valu -o v -- python bench.py $args > $out/valu.log 2>&1
sd=$(dirname $(find $out/stats -name '*kernel_stats.csv' | head -1))
fd=$(dirname $(find $out/fetch -name '*counter_collection.csv' | head -1))
wd=$(dirname $(find $out/write -name '*counter_collection.csv' | head -1))
mkdir -p $out/profiles
python tools/summarize_profiles.py $tag $sd $fd $wd $out/bench.json && cp profiles/${tag}_* $out/profiles/
cp $out/bench.json $out/profiles/${tag}_bench_2p22.json
vd=$(find $out/valu -name '*counter_collection.csv' | head -1)
python tools/instr_budget.py $out/valu > $out/profiles/${tag}_valu_instruction_budget.txt 2>&1 || true
ls -la $out/profiles
