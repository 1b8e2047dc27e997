#!/bin/bash
# GPU-box script: where the time of the 1x1 (qkv / proj) conv goes -- ablation builds of the kernel via env knobs.
cd "$GRAFT_REPO_ROOT" || exit 1
for SH in "512,128,384,16,1 --split-in --qkv-out" "512,128,128,16,1 --res" "1024,256,768,8,1 --split-in --qkv-out"; do
  for AB in 0 1 2 3; do
    echo "shape [$SH] ablate=$AB  $(I2R_H2_ABLATE=$AB timeout -k 10 100 python3 tools/conv_bench.py --op h2 --iters 50 --shape $SH 2>&1 | grep 'h2 B')"
  done
  echo "shape [$SH] BN=256   $(I2R_H2_BN=256 timeout -k 10 100 python3 tools/conv_bench.py --op h2 --iters 50 --shape $SH 2>&1 | grep 'h2 B')"
done
