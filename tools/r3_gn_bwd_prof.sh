#!/bin/bash
# GPU-box script: kernel times of the GroupNorm backward at the training batch: slot accumulation with the
# channel sums of a workgroup gathered in LDS first (lds), with one atomic per sample and channel (acc), per-sample
# partials + row reduction (part).
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3_gn_bwd
mkdir -p $O
for M in lds acc part; do
  F="--acc"; [ $M = part ] && F=""
  if [ $M = acc ]; then export I2R_GN_BWD_LDS=0; else unset I2R_GN_BWD_LDS; fi
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$M -o g -- python3 tools/gn_bench.py --batch 32 --iters 50 --bwd --add $F > $O/$M.log 2>&1 || { echo "$M failed"; tail -5 $O/$M.log; exit 1; }
  echo "== $M"; grep -E "gn_bwd|reduce_rows" $(find $O/$M -name "*kernel_stats.csv" | head -1) | cut -c1-120
done
for M in lds acc; do
echo "== per shape, $M"
F=$(find $O/$M -name "*kernel_trace.csv" | head -1)
python3 - "$F" <<'PY'
import csv,sys,collections
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'gn_bwd' in r['Kernel_Name']]
# per shape: 53 launches each in order
per=collections.defaultdict(list)
for i,r in enumerate(rows):
    per[i//53].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in per.items():
    v=sorted(v); print('shape',k, rows[k*53]['Kernel_Name'][:40], 'grid', rows[k*53].get('Grid_Size_X', rows[k*53].get('Grid_Size')), 'median us', v[len(v)//2])
PY
done
