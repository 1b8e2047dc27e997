#!/bin/bash
# GPU-box script: shape A's sampler with the tiled 1x1 kernel fetching the channels of 1x1-image positions with 16-byte
# loads (default) and with eight dword loads (I2R_H2_HW1_SCALAR=1), alternating on one box.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_hw1vec_ab.log
: > $O
for W in 0 1 0 1; do
  if [ $W = 1 ]; then export I2R_H2_HW1_SCALAR=1; else unset I2R_H2_HW1_SCALAR; fi
  timeout -k 10 300 python3 bench.py --gpus 1 --workload A_sine24 --steps 6 --warmup 2 --extra none --train-steps 0 --cpu-series 0 > gpurun_out/_s.json 2> gpurun_out/_s.err || { echo "bench failed"; tail -5 gpurun_out/_s.err; exit 1; }
  python3 -c "import json;d=json.loads(open('gpurun_out/_s.json').read().strip().splitlines()[-1]);print('A hw1 scalar loads=$W', round(d['value'],1),'series/s k1 ms', round(d['kernels']['conv_fwd_h2_k1']['ms'],2))" >> $O
done
cat $O
