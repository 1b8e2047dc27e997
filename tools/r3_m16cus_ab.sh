#!/bin/bash
# GPU-box script: conv3x3_h2_m16_kernel on fewer than all CUs (I2R_H2_M16_CUS workgroups per XCD), so that the other
# stream's memory-bound kernels find free CUs while it runs; B1 sampler, two-stream form, one box.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_m16cus_ab.log
: > $O
for CU in 32 28 24 32 28 24 20; do
  export I2R_H2_M16_CUS=$CU
  timeout -k 10 300 python3 bench.py --gpus 1 --workload B1_weather96 --steps 6 --warmup 2 --extra none --train-steps 0 --cpu-series 0 > gpurun_out/_s.json 2> gpurun_out/_s.err || { echo "bench failed"; tail -5 gpurun_out/_s.err; exit 1; }
  python3 -c "import json;d=json.loads(open('gpurun_out/_s.json').read().strip().splitlines()[-1]);print('B1 m16 workgroups per XCD=$CU', round(d['value'],1),'series/s')" >> $O
done
cat $O
