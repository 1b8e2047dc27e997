#!/bin/bash
# A/B of two library builds: 3x3 bench shapes (with and without residual) and the 1x1 shapes the lock-step 16x16x32
# kernel takes; $1 = the other library
for ARGS in "--shape 4096,128,128,16,3" "--res --shape 4096,128,128,16,3" "--shape 4096,256,256,8,3" "--res --shape 4096,256,256,8,3" "--shape 4096,512,256,8,3" "--res --shape 4096,256,256,4,3" "--res --shape 4096,512,512,2,3" "--res --shape 2048,128,128,16,1" "--res --shape 4096,256,256,8,1" "--res --shape 4096,256,256,4,1"; do
  for L in "$1" ""; do
    R=$(I2R_LIB_PATH=$L timeout -k 10 100 python3 tools/conv_bench.py --op h2 --split-in --iters 20 $ARGS 2>&1 | grep 'h2 B'); rc=$?
    echo "${L:+base} ${L:-tree} [$ARGS] $R"
    if [ $rc -gt 1 ]; then echo "stopping"; exit 1; fi
  done
done
