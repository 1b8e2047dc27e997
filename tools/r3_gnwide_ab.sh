#!/bin/bash
# GPU-box script: the B1 / D train step with the wide (256-lane) GroupNorm forms for few groups on and off, same box.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_gnwide_ab.log
: > $O
for WL in B1_weather96 D_etth1_768; do
  for W in 1 0 1 0; do
    export I2R_GN_WIDE=$W
    timeout -k 10 200 python3 bench.py --mode train --workload $WL --steps 60 --warmup 3 > gpurun_out/_t.json 2> gpurun_out/_t.err || { echo "train $WL failed"; tail -5 gpurun_out/_t.err; exit 1; }
    python3 -c "import json;d=json.loads(open('gpurun_out/_t.json').read().strip().splitlines()[-1]);print('$WL I2R_GN_WIDE=$W', round(d['ms_per_step'],3),'ms', round(d['value'],1))" >> $O
  done
done
cat $O
