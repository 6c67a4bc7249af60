#!/bin/bash
# round-3 fourth GPU pass: whole suite with the new shape tests, default bench line (2^22 + 2^20 leg), REST front end rate,
# server throughput at Semaphore-class sizes over both routes, stall counters of the level-1 kernels
out=gpurun_out/r03d
mkdir -p $out
( timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 2>&1 | tail -30 ) > $out/pytest.txt
timeout 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err
# REST front end alone: status polls over keep-alive connections
mkdir -p /tmp/fe/build && cp tests/golden/r1cs_n8/circuit.zkey /tmp/fe/r1cs_n8.zkey
( cd /tmp/fe && ZKHIP_QUEUE=64 ZKHIP_WORKERS=0 $OLDPWD/rapidsnark-old_amd/proverServer 9471 r1cs_n8.zkey > /dev/null 2>&1 & echo $! > /tmp/fe/pid )
sleep 4
for t in 1 4 8 16; do tools/http_load 9471 $t 2 /status; done > $out/http_rate.txt 2>&1
tools/http_load 9471 8 2 /status/1 >> $out/http_rate.txt 2>&1
kill $(cat /tmp/fe/pid)
for k in 14 16; do
  for route in input witness; do timeout 300 python tools/server_bench.py $k 256 0 $route 2>/dev/null; done
  timeout 300 python bench.py --log2n $k --batch 4 --steps 256 --warmup 8 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C-ABI loop 2^$k batch 4:', d['value'], 'proofs/s')"
done > $out/server.txt 2>&1
timeout 1500 bash tools/stall_probe.sh r03d_stall > /dev/null 2>&1
cp gpurun_out/r03d_stall/stall.txt $out/stall.txt 2>/dev/null
cat $out/pytest.txt $out/http_rate.txt $out/server.txt; head -c 3000 $out/bench_default.json; tail -5 $out/bench_default.err
