#!/bin/bash
# conv1x1_h2_m16 on the 256->768 qkv conv and the 256->256 proj conv: whole / no stores (1) / no in-loop epilogue (128)
for A in 0 1 128; do
  R=$(I2R_H2_ABLATE=$A timeout -k 10 100 python3 tools/conv_bench.py --op h2 --split-in --qkv-out --iters 20 --shape 1024,256,768,8,1 2>&1 | grep 'h2 B'); rc=$?
  echo "qkv ablate=$A $R"; if [ $rc -gt 1 ]; then exit 1; fi
  R=$(I2R_H2_ABLATE=$A timeout -k 10 100 python3 tools/conv_bench.py --op h2 --split-in --iters 20 --shape 1024,256,768,8,1 2>&1 | grep 'h2 B'); rc=$?
  echo "plain-out ablate=$A $R"; if [ $rc -gt 1 ]; then exit 1; fi
  R=$(I2R_H2_ABLATE=$A timeout -k 10 100 python3 tools/conv_bench.py --op h2 --split-in --res --iters 20 --shape 1024,256,256,8,1 2>&1 | grep 'h2 B'); rc=$?
  echo "proj+res ablate=$A $R"; if [ $rc -gt 1 ]; then exit 1; fi
done
