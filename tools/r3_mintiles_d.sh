#!/bin/bash
# GPU-box script: shape D (two 512-series graphs) with the 16x16x32 3x3 kernel taken from 256 tiles on (default), from 64
# (its 4x4 level: 64 tiles per half batch) and from 32, alternating on one box.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_mintiles_d.log
: > $O
for MT in 256 64 256 64 32; do
  export I2R_H2_M16_MINTILES=$MT
  timeout -k 10 300 python3 bench.py --gpus 1 --workload D_etth1_768 --steps 6 --warmup 2 --extra none --train-steps 0 --cpu-series 0 > gpurun_out/_s.json 2> gpurun_out/_s.err || { echo "bench failed"; tail -5 gpurun_out/_s.err; exit 1; }
  python3 -c "import json;d=json.loads(open('gpurun_out/_s.json').read().strip().splitlines()[-1]);print('D m16_mintiles=$MT', round(d['value'],1),'series/s')" >> $O
done
cat $O
