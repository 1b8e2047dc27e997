#!/bin/bash
# conv1x1_h2_ws_kernel on the qkv shapes of the bench batch: I2R_H2_ABLATE variants
for A in 0 64 1; do
  for ARGS in "--qkv-out --shape 4096,256,768,8,1" "--qkv-out --shape 2048,128,384,16,1" "--qkv-out --shape 4096,256,768,4,1"; do
    R=$(I2R_H2_ABLATE=$A timeout -k 10 100 python3 tools/conv_bench.py --op h2 --split-in --iters 20 $ARGS 2>&1 | grep 'h2 B'); rc=$?
    echo "ablate=$A [$ARGS] $R"
    if [ $rc -gt 1 ]; then echo "stopping"; exit 1; fi
  done
done
