#!/bin/bash
# GPU-box script (round 3 records): single-stream evaluation profile + PMC traffic passes for B1 (4096 series) and D
# (1024), graph train-step kernel stats at two step counts (their difference is the steady-state step), launch tables.
# Each GPU step has its own time limit; the script stops at the first failure.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3prof
mkdir -p $O
for CASE in "B1_weather96 4096" "D_etth1_768 1024"; do
  set -- $CASE; WL=$1; B=$2
  timeout -k 10 200 python3 bench.py --workload $WL --steps 2 --warmup 1 --extra none --train-steps 0 --cpu-series 0 --launch-table $O/launch_table_${WL}_b$B.csv > $O/bench_$WL.json 2> $O/bench_$WL.err || { echo "bench $WL failed"; tail -5 $O/bench_$WL.err; exit 1; }
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/eval_$WL -o eval -- python3 tools/eval_once.py $B $WL 3 > $O/eval_$WL.log 2>&1 || { echo "eval stats $WL failed"; tail -5 $O/eval_$WL.log; exit 1; }
  timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$WL -o f -- python3 tools/eval_once.py $B $WL 1 > $O/pmc_fetch_$WL.log 2>&1 || { echo "pmc fetch $WL failed"; tail -5 $O/pmc_fetch_$WL.log; exit 1; }
  timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$WL -o w -- python3 tools/eval_once.py $B $WL 1 > $O/pmc_write_$WL.log 2>&1 || { echo "pmc write $WL failed"; tail -5 $O/pmc_write_$WL.log; exit 1; }
  F=$(find $O/pmc_fetch_$WL -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_write_$WL -name "*counter_collection.csv" | head -1)
  python3 tools/traffic_summary.py $F $W $O/r03_traffic_${WL}_b$B.json $WL $B $O/launch_table_${WL}_b$B.csv > $O/traffic_$WL.log 2>&1 || { echo "traffic summary $WL failed"; tail -5 $O/traffic_$WL.log; exit 1; }
  echo "$WL ok"; cut -c1-600 $O/r03_traffic_${WL}_b$B.json
done
for WL in B1_weather96 D_etth1_768; do
  for ST in 10 60; do
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_${WL}_$ST -o t -- python3 bench.py --mode train --workload $WL --steps $ST --warmup 3 > $O/train_${WL}_$ST.json 2> $O/train_${WL}_$ST.err || { echo "train $WL $ST failed"; tail -5 $O/train_${WL}_$ST.err; exit 1; }
  done
  echo "train $WL ok"; cut -c1-200 $O/train_${WL}_60.json
done
