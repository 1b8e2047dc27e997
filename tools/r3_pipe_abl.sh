#!/bin/bash
# conv3x3_h2_m16 (pipelined form): what the epilogue's stores cost (I2R_H2_ABLATE=1 sends them out of range)
for SH in "4096,128,128,16,3" "4096,256,256,8,3" "4096,512,256,8,3"; do
  for A in 0 32; do
    R=$(I2R_H2_ABLATE=$A timeout -k 10 100 python3 tools/conv_bench.py --op h2 --split-in --iters 20 --shape $SH 2>&1 | grep 'h2 B')
    rc=$?
    echo "ablate=$A $R"
    if [ $rc -gt 1 ]; then echo "stopping"; exit 1; fi
  done
done
