#!/bin/bash
# GPU-box script: conv3x3_h2_m16_kernel with at least I2R_H2_M16_TPW tiles per workgroup (short launches on fewer CUs,
# pipelined over their tiles, beside the other stream's kernels); B1 sampler, two-stream form, one box.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_m16tpw_ab.log
: > $O
for T in 1 2 1 2 4; do
  export I2R_H2_M16_TPW=$T
  timeout -k 10 300 python3 bench.py --gpus 1 --workload B1_weather96 --steps 6 --warmup 2 --extra none --train-steps 0 --cpu-series 0 > gpurun_out/_s.json 2> gpurun_out/_s.err || { echo "bench failed"; tail -5 gpurun_out/_s.err; exit 1; }
  python3 -c "import json;d=json.loads(open('gpurun_out/_s.json').read().strip().splitlines()[-1]);print('B1 min tiles per workgroup=$T', round(d['value'],1),'series/s')" >> $O
done
cat $O
