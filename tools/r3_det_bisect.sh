#!/bin/bash
# GPU-box script: which kernel switch makes the two-stream sampler bit-reproducible again (tools/r3_determinism_check.py).
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_det_bisect.log
: > $O
for CFG in "none=1" "I2R_H2_M16_W2OFF=1" "I2R_H2_M16=0" "I2R_H2_WS=0" "I2R_H2_M16_K1=0" "I2R_GN_SMALL=0" "I2R_H2_PERSIST=0" "I2R_ATTN_CHUNK_MB=0"; do
  echo "== $CFG" >> $O
  env $CFG timeout -k 10 200 python3 tools/r3_determinism_check.py B1_weather96 4096 3 2>&1 | grep "^run" >> $O || { echo "failed $CFG" >> $O; }
done
cat $O
