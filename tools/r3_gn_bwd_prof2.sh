#!/bin/bash
# GPU-box script: GroupNorm backward kernel times at the training batch, atomics form, 64-lane forms (I2R_GN_WIDE=0:
# four samples of a group per workgroup, their dgamma / dbeta sums pre-reduced in LDS) for comparison with the
# whole-workgroup forms of tools/r3_gn_bwd_prof.sh.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3_gn_bwd
mkdir -p $O
export I2R_GN_WIDE=0
for B in 32 128; do
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/narrow$B -o g -- python3 tools/gn_bench.py --batch $B --iters 50 --bwd --add --acc > $O/narrow$B.log 2>&1 || { echo "failed"; tail -5 $O/narrow$B.log; exit 1; }
F=$(find $O/narrow$B -name "*kernel_trace.csv" | head -1)
python3 - "$F" <<'PY'
import csv,sys,collections
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'gn_bwd' in r['Kernel_Name']]
per=collections.defaultdict(list)
for i,r in enumerate(rows):
    per[i//53].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in per.items():
    v=sorted(v); print('shape',k, rows[k*53]['Kernel_Name'][:40], 'median us', v[len(v)//2])
PY
done
