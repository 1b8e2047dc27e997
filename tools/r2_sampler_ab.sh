#!/bin/bash
# GPU-box script: A/B of sampler knobs given as ENV=VALUE configs ("base" = none) on B1 at the bench batch.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2ab
mkdir -p $O
WL=${WL:-B1_weather96}
for CFG in "$@"; do
  if [ "$CFG" = "base" ]; then E=""; else E="$CFG"; fi
  R=$(env $E timeout -k 10 300 python3 bench.py --workload $WL --steps 3 --warmup 1 --train-steps 0 --cpu-series 0 --extra none 2> $O/err.log) || { echo "$CFG failed"; tail -5 $O/err.log; exit 1; }
  echo "$WL [$CFG] $(echo $R | python3 -c 'import sys,json; r=json.loads(sys.stdin.read()); print(round(r["value"],1), "series/s", round(r["eager_eval_ms"],2), "ms eager eval;", {k: round(v["ms"],2) for k,v in r["kernels"].items()})')"
done
