#!/bin/bash
# GPU-box script: kernel statistics of the sampler bench itself (two half-batch graphs on two streams), i.e. of the
# timed region's own kernel mix -- the per-evaluation tables elsewhere are single-stream at the full batch.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3prof_timed
mkdir -p $O
WL=${1:-B1_weather96}
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$WL -o s -- python3 bench.py --workload $WL --steps 4 --warmup 1 --extra none --train-steps 0 --cpu-series 0 > $O/$WL.json 2> $O/$WL.err || { echo "failed"; tail -5 $O/$WL.err; exit 1; }
F=$(find $O/$WL -name "*kernel_stats.csv" | head -1)
python3 - "$F" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)/1e6
print('total kernel ms', round(tot,1))
for r in rows[:32]:
    print(r['Name'][:84].ljust(84), r['Calls'].rjust(7), str(round(float(r['TotalDurationNs'])/1e6,1)).rjust(8), 'ms', str(round(float(r['AverageNs'])/1e3,1)).rjust(8),'us', r['Percentage'])
PY
