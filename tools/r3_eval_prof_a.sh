#!/bin/bash
# GPU-box script: kernel statistics of three single-stream sampler evaluations of shape A (8x8 images) at 4096 series.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3prof_a
mkdir -p $O
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/eval_A -o eval -- python3 tools/eval_once.py 4096 A_sine24 3 > $O/eval_A.log 2>&1 || { echo "eval failed"; tail -5 $O/eval_A.log; exit 1; }
F=$(find $O/eval_A -name "*kernel_stats.csv" | head -1)
python3 - "$F" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)/3e6
print('ms per evaluation', round(tot,2))
for r in rows[:22]:
    print(r['Name'][:78].ljust(78), int(r['Calls'])//3, round(float(r['TotalDurationNs'])/3e6,3), 'ms', round(float(r['AverageNs'])/1e3,1),'us')
PY
