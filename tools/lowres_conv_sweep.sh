#!/bin/bash
# GPU-box script: kernel times (rocprofv3) of the fp32 conv forward on the low-resolution training layers at batch 32 as
# a function of the split-K depth (I2R_CONV_NSPLIT; unset = the plan's own choice).
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/lowres
mkdir -p $O
for SH in 32,512,512,2,3 32,256,256,4,3 32,1024,512,2,3; do
  for NS in plan 4 8 16 64; do
    if [ $NS = plan ]; then unset I2R_CONV_NSPLIT; else export I2R_CONV_NSPLIT=$NS; fi
    timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$SH.$NS -o t -- python3 tools/conv_bench.py --op fwd --iters 50 --shape $SH > $O/$SH.$NS.log 2>&1 || { echo "failed $SH $NS"; tail -3 $O/$SH.$NS.log; exit 1; }
    python3 - $O/$SH.$NS/t_kernel_stats.csv "$SH nsplit=$NS" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
out = []
for r in rows:
    if "conv_fwd_kernel" in r["Name"] or "splitk_reduce" in r["Name"]:
        out.append("%s %.1f us" % (r["Name"][:28], float(r["AverageNs"]) / 1e3))
print(sys.argv[2], " | ".join(out), flush=True)
PY
  done
done
