#!/bin/bash
# GPU-box script: B1 / D / A train step with the GroupNorm backward's dgamma / dbeta sums gathered in LDS per workgroup
# (I2R_GN_BWD_LDS, default on) and without, alternating on one box.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_train_ab.log
: > $O
for WL in ${WLS:-B1_weather96 D_etth1_768 A_sine24}; do
  for W in 1 0 1 0; do
    export ${SW:-I2R_GN_BWD_LDS}=$W
    timeout -k 10 200 python3 bench.py --mode train --workload $WL --steps 60 --warmup 3 > gpurun_out/_t.json 2> gpurun_out/_t.err || { echo "train $WL failed"; tail -5 gpurun_out/_t.err; exit 1; }
    python3 -c "import json;d=json.loads(open('gpurun_out/_t.json').read().strip().splitlines()[-1]);print('$WL ${SW:-I2R_GN_BWD_LDS}=$W', round(d['ms_per_step'],3),'ms', round(d['value'],1))" >> $O
  done
done
cat $O
