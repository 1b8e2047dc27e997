#!/bin/bash
# GPU-box script: shape A's sampler with the 3x3 convs of its 1x1 level run as 1x1 convs of the centre tap (default) and
# as full 3x3 launches (I2R_CENTER_TAP=0), alternating on one box.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_center_ab.log
: > $O
for W in 1 0 1 0; do
  export I2R_CENTER_TAP=$W
  timeout -k 10 300 python3 bench.py --gpus 1 --workload A_sine24 --steps 6 --warmup 2 --extra none --train-steps 0 --cpu-series 0 > gpurun_out/_s.json 2> gpurun_out/_s.err || { echo "bench failed"; tail -5 gpurun_out/_s.err; exit 1; }
  python3 -c "import json;d=json.loads(open('gpurun_out/_s.json').read().strip().splitlines()[-1]);print('A center_tap=$W', round(d['value'],1),'series/s')" >> $O
done
cat $O
