#!/bin/bash
# GPU-box script: B1 sampler at 4096 (the bench default), 6144 and 8192 series per step (two half-batch graphs each).
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_batch_ab.log
: > $O
for B in ${BATCHES:-4096 8192 6144 4096 8192}; do
  timeout -k 10 400 python3 bench.py --gpus 1 --workload B1_weather96 --batch $B --steps 4 --warmup 2 --extra none --train-steps 0 --cpu-series 0 > gpurun_out/_s.json 2> gpurun_out/_s.err || { echo "bench $B failed"; tail -5 gpurun_out/_s.err; exit 1; }
  python3 -c "import json;d=json.loads(open('gpurun_out/_s.json').read().strip().splitlines()[-1]);print('B1 series per step=$B', round(d['value'],1),'series/s', round(d['ms_per_step'],1), 'ms/step')" >> $O
done
cat $O
