#!/bin/bash
# GPU-box script: shape D's 1024-series step as one graph (the default: the two-stream form starts at 2048 series) and as
# two 512-series graphs on two streams (I2R_DUAL_STREAM_MIN=1024), alternating on one box.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_dual_d.log
: > $O
for DS in 2048 1024 2048 1024; do
  export I2R_DUAL_STREAM_MIN=$DS
  timeout -k 10 300 python3 bench.py --gpus 1 --workload D_etth1_768 --steps 6 --warmup 2 --extra none --train-steps 0 --cpu-series 0 > gpurun_out/_s.json 2> gpurun_out/_s.err || { echo "bench failed"; tail -5 gpurun_out/_s.err; exit 1; }
  python3 -c "import json;d=json.loads(open('gpurun_out/_s.json').read().strip().splitlines()[-1]);print('D dual_stream_min=$DS', round(d['value'],1),'series/s')" >> $O
done
cat $O
