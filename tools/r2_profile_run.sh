#!/bin/bash
# GPU-box script: default bench, single-stream eval profile, PMC traffic passes, 2-rank rehearsal.  Each GPU step has
# its own time limit; the script stops at the first failure.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2prof
mkdir -p $O
if [ "$1" != "noprof-bench" ]; then
timeout -k 10 400 python3 bench.py --steps 3 --warmup 1 --launch-table $O/launch_table_B1_b4096.csv > $O/bench_default.json 2> $O/bench_default.err || { echo "bench failed"; tail -5 $O/bench_default.err; exit 1; }
echo "bench ok"; cut -c1-300 $O/bench_default.json
fi
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/eval_stats -o eval -- python3 tools/eval_once.py 4096 B1_weather96 3 > $O/eval_stats.log 2>&1 || { echo "eval stats failed"; tail -5 $O/eval_stats.log; exit 1; }
echo "stats ok"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 tools/eval_once.py 4096 B1_weather96 1 > $O/pmc_fetch.log 2>&1 || { echo "pmc fetch failed"; tail -5 $O/pmc_fetch.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 tools/eval_once.py 4096 B1_weather96 1 > $O/pmc_write.log 2>&1 || { echo "pmc write failed"; tail -5 $O/pmc_write.log; exit 1; }
echo "pmc ok"
find $O -name "*.csv" | head -20
F=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_write -name "*counter_collection.csv" | head -1)
python3 tools/traffic_summary.py $F $W $O/r02_traffic_B1_weather96_b4096.json B1_weather96 4096 > $O/traffic.log 2>&1 || { echo "traffic summary failed"; tail -5 $O/traffic.log; exit 1; }
cat $O/r02_traffic_B1_weather96_b4096.json
# multi-rank rehearsal: 2 ranks on this box's one GPU, collectives staged through the host (gloo); then the refusal of
# the RCCL launch with more ranks than devices (must exit non-zero at once, not hang)
timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --steps 2 --warmup 1 --batch 512 --dist-backend gloo-host --dist-timeout 120 --cpu-series 0 > $O/rehearsal2.json 2> $O/rehearsal2.err || { echo "rehearsal failed"; tail -20 $O/rehearsal2.err; exit 1; }
echo "rehearsal ok"; cut -c1-400 $O/rehearsal2.json
timeout -k 10 200 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29532 bench.py --gpus 2 --steps 1 --warmup 0 --batch 64 --cpu-series 0 > $O/refuse.json 2> $O/refuse.err; echo "refusal rc=$?"; grep -h "bench.py: WORLD_SIZE" $O/refuse.err | head -2
