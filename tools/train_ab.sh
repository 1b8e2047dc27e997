#!/bin/bash
# GPU-box script: graph-replayed train step of the given workloads with an environment switch off and on.
#   tools/train_ab.sh SWITCH "OFF ON" WORKLOAD...   (a value "unset" removes the variable)     e.g.  tools/train_ab.sh I2R_WGRAD_TILE "0 1" B1_weather96 D_etth1_768
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
SW=$1; VALS=$2; shift 2
mkdir -p gpurun_out/ab
for WL in "$@"; do
  for V in $VALS; do
    if [ "$V" = unset ]; then E="-u $SW"; else E="$SW=$V"; fi
    T=$(echo "$V" | tr '/' '_')
    env $E timeout -k 10 200 python3 bench.py --mode train --workload $WL --steps 30 --warmup 5 > gpurun_out/ab/$WL.$T.json 2> gpurun_out/ab/$WL.$T.err || { echo "$WL $SW=$V failed"; tail -5 gpurun_out/ab/$WL.$T.err; exit 1; }
    python3 - "$WL" "$SW=$V" gpurun_out/ab/$WL.$T.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
print("%-18s %-22s %8.1f %s  %.2f ms/step" % (sys.argv[1], sys.argv[2], r["value"], r["unit"], r["ms_per_step"]), flush=True)
PY
  done
done
