#!/bin/bash
# GPU-box script: A/B of a training knob given as ENV=VALUE pairs ("base" = none), graph-mode train step.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2ab
mkdir -p $O
for WL in B1_weather96 D_etth1_768 C_electricity96 A_sine24; do
  for CFG in "$@"; do
    if [ "$CFG" = "base" ]; then E=""; else E="$CFG"; fi
    R=$(env $E timeout -k 10 200 python3 bench.py --mode train --workload $WL --steps 30 --warmup 5 2> $O/err.log) || { echo "$WL $CFG failed"; tail -5 $O/err.log; exit 1; }
    echo "$WL [$CFG] $(echo $R | python3 -c 'import sys,json; r=json.loads(sys.stdin.read()); print(round(r["value"],1), "samples/s", round(r["ms_per_step"],3), "ms/step")')"
  done
done
