#!/bin/bash
# GPU-box script: what the dgamma / dbeta atomics of the GroupNorm backward cost in the B1 / D train step (ablation:
# results are wrong with I2R_GN_ABL set, only the time is read).
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_gnabl_ab.log
: > $O
for WL in B1_weather96 D_etth1_768; do
  for W in 0 1 0 1; do
    if [ $W = 1 ]; then export I2R_GN_ABL=1; else unset I2R_GN_ABL; fi
    timeout -k 10 200 python3 bench.py --mode train --workload $WL --steps 60 --warmup 3 > gpurun_out/_t.json 2> gpurun_out/_t.err || { echo "train $WL failed"; tail -5 gpurun_out/_t.err; exit 1; }
    python3 -c "import json;d=json.loads(open('gpurun_out/_t.json').read().strip().splitlines()[-1]);print('$WL atomics ablated=$W', round(d['ms_per_step'],3),'ms', round(d['value'],1))" >> $O
  done
done
cat $O
