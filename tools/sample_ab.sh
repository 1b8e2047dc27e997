#!/bin/bash
# GPU-box script: the sampler line of bench.py (no train legs, no CPU baseline) with an environment switch off and on.
#   tools/sample_ab.sh SWITCH "OFF ON" WORKLOAD...      (a value "unset" removes the variable)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
SW=$1; VALS=$2; shift 2
mkdir -p gpurun_out/sab
for WL in "$@"; do
  for V in $VALS; do
    if [ "$V" = unset ]; then E="-u $SW"; else E="$SW=$V"; fi
    T=$(echo "$V" | tr '/' '_')
    env $E timeout -k 10 300 python3 bench.py --workload $WL --steps 4 --warmup 1 --cpu-series 0 --train-steps 0 --extra none > gpurun_out/sab/$WL.$T.json 2> gpurun_out/sab/$WL.$T.err || { echo "$WL $SW=$V failed"; tail -5 gpurun_out/sab/$WL.$T.err; exit 1; }
    python3 - "$WL" "$SW=$V" gpurun_out/sab/$WL.$T.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
more = {m["kernel"]: round(m["achieved"]) for m in r.get("roofline_more", [])}
print("%-16s %-18s %8.1f %s  %.1f ms/step  frac %.4f  %s" % (sys.argv[1], sys.argv[2], r["value"], r["unit"], r["ms_per_step"],
                                                          r["roofline"]["frac"], more), flush=True)
PY
  done
done
