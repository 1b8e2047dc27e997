#!/bin/bash
# GPU-box script: rocprofv3 kernel stats of the graph-replayed train step for the given workloads.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2train
mkdir -p $O
for WL in "$@"; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$WL -o t -- python3 bench.py --mode train --workload $WL --steps 20 --warmup 3 > $O/$WL.json 2> $O/$WL.err || { echo "$WL failed"; tail -5 $O/$WL.err; exit 1; }
  cut -c1-300 $O/$WL.json
  python3 tools/kstats.py $O/$WL/t_kernel_stats.csv 25 | head -32
done
