#!/bin/bash
# GPU-box script: ONE sampler graph (bit-reproducible alone) beside a second stream that runs one family of this
# library's kernels in a loop: which family's company makes the sampler's results vary from run to run?
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_det_own.log
: > $O
for F in gn conv3 conv1 attn; do
  echo "== beside $F" >> $O
  I2R_DUAL_STREAM_MIN=0 DET_OWN=$F timeout -k 10 300 python3 tools/r3_determinism_check.py B1_weather96 2048 3 2>&1 | grep "^run\|Error\|error" >> $O
done
cat $O
