#!/bin/bash
# GPU-box script: what the fp32 conv kernel costs before and after its K loop on a low-resolution layer (batch 32, 2x2,
# 512 output channels): few-chunk problems, with and without the epilogue (I2R_CONV_ABLATE=8).
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/lowres_fixed
mkdir -p $O
for SH in 32,4,512,2,3 32,32,512,2,3 32,128,512,2,3; do
  for AB in 0 8; do
    I2R_CONV_NSPLIT=1 I2R_CONV_ABLATE=$AB timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$SH.$AB -o t -- python3 tools/conv_bench.py --op fwd --iters 50 --shape $SH > $O/$SH.$AB.log 2>&1 || { echo "failed $SH $AB"; tail -3 $O/$SH.$AB.log; exit 1; }
    python3 - $O/$SH.$AB/t_kernel_stats.csv "$SH ablate=$AB" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(sys.argv[2], " | ".join("%s %.1f us" % (r["Name"][:28], float(r["AverageNs"]) / 1e3) for r in rows if "conv_fwd_kernel" in r["Name"]), flush=True)
PY
  done
done
