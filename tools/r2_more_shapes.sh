#!/bin/bash
# GPU-box script: sampler + train lines of the non-BASELINE architecture families (F, H, I of workloads.py).
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/more_shapes
mkdir -p $O
for WL in F_mujoco24 I_weather96_32 H_energy768; do
  timeout -k 10 280 python3 bench.py --workload $WL --steps 2 --warmup 1 --cpu-series 0 --extra none > $O/$WL.json 2> $O/$WL.err || { echo "$WL failed"; tail -5 $O/$WL.err; exit 1; }
  python3 - $O/$WL.json <<'PY'
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
t=r.get("train",{})
print(r["config"]["workload"], "batch", r["config"].get("series_per_step"), round(r["value"],1), "series/s", "frac", round(r["roofline"]["frac"],3), "| train", {k:(round(v,1) if isinstance(v,float) else v) for k,v in t.items() if k in ("value","ms_per_step","batch")})
PY
done
