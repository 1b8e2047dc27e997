#!/bin/bash
# GPU-box script: sampler with the four-groups-per-slot GroupNorm forward (I2R_GN_SMALL, now also for 256-element groups)
# on and off; A (its top level: 23 launches per evaluation) and B1, alternating on one box.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_gnsmall_ab.log
: > $O
for WL in A_sine24 B1_weather96; do
for W in 1 0 1 0; do
  export I2R_GN_SMALL=$W
  timeout -k 10 300 python3 bench.py --gpus 1 --workload $WL --steps 6 --warmup 2 --extra none --train-steps 0 --cpu-series 0 > gpurun_out/_s.json 2> gpurun_out/_s.err || { echo "bench failed"; tail -5 gpurun_out/_s.err; exit 1; }
  python3 -c "import json;d=json.loads(open('gpurun_out/_s.json').read().strip().splitlines()[-1]);print('$WL gn_small=$W', round(d['value'],1),'series/s gn ms', round(d['kernels']['groupnorm_fwd']['ms'],2))" >> $O
done
done
cat $O
