#!/bin/bash
# GPU-box script: steady-state kernel statistics of shape A's train step (batch 128, 8x8 images): 60-step minus 10-step run.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3prof_a
mkdir -p $O
WL=${1:-A_sine24}
for ST in 10 60; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_${WL}_$ST -o t -- python3 bench.py --mode train --workload $WL --steps $ST --warmup 3 > $O/train_${WL}_$ST.json 2> $O/train_${WL}_$ST.err || { echo "train $WL $ST failed"; tail -5 $O/train_${WL}_$ST.err; exit 1; }
done
S10=$(find $O/train_${WL}_10 -name "*kernel_stats.csv" | head -1); S60=$(find $O/train_${WL}_60 -name "*kernel_stats.csv" | head -1)
python3 tools/kstats_diff.py $S10 $S60 10 60 $O/steady_$WL.csv > $O/steady_$WL.txt
head -40 $O/steady_$WL.txt | cut -c1-135
