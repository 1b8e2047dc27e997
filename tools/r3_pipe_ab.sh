#!/bin/bash
# A/B of two builds of the library on the bench batch's 3x3 shapes (tools/conv_bench.py): $1 = other library
for SH in "4096,128,128,16,3" "4096,256,256,8,3" "4096,256,128,16,3" "4096,512,256,8,3" "4096,256,256,4,3" "4096,512,512,2,3" "1024,128,128,32,3" "1024,256,256,16,3"; do
  for L in "$1" ""; do
    R=$(I2R_LIB_PATH=$L timeout -k 10 100 python3 tools/conv_bench.py --op h2 --split-in --iters 20 --shape $SH 2>&1 | grep 'h2 B')
    rc=$?
    echo "${L:-tree} $R"
    if [ $rc -gt 1 ]; then echo "stopping"; exit 1; fi
  done
done
