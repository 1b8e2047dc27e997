#!/bin/bash
# GPU-box script: the weight-stationary 1x1 kernel from 65536 positions on (default) against from 32768 on (the 4x4-level
# qkv convs of a 2048-series half batch; shape D's 8x8 level at 512 series), alternating on one box.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_wsmin_ab.log
: > $O
for WL in B1_weather96 D_etth1_768; do
  for MP in 65536 32768 65536 32768; do
    export I2R_H2_WS_MINPOS=$MP
    timeout -k 10 300 python3 bench.py --gpus 1 --workload $WL --steps 6 --warmup 2 --extra none --train-steps 0 --cpu-series 0 > gpurun_out/_s.json 2> gpurun_out/_s.err || { echo "bench failed"; tail -5 gpurun_out/_s.err; exit 1; }
    python3 -c "import json;d=json.loads(open('gpurun_out/_s.json').read().strip().splitlines()[-1]);print('$WL ws_minpos=$MP', round(d['value'],1),'series/s')" >> $O
  done
done
cat $O
