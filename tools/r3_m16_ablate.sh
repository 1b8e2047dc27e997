#!/bin/bash
# Ablation builds of conv3x3_h2_m16_kernel (-DM16_ABL=mask compiles parts of its main loop out; results are garbage,
# only the time counts): which non-MFMA part costs what.
#   tools/r3_m16_ablate.sh build   (in the build container: build/libi2r_abl<mask>.so)
#   tools/r3_m16_ablate.sh run     (on the GPU box, from the repo root)
set -e
MASKS="0 1 2 4 8 16 3 7 15 31"
SRC=imageni2r_amd/csrc
if [ "$1" = "build" ]; then
  mkdir -p build
  OL=""; for o in common delay_embed elementwise groupnorm linear optim conv_fwd conv_wgrad attention; do OL="$OL $SRC/$o.o"; done
  for M in $MASKS; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DM16_ABL=$M -c $SRC/conv_fwd_h2.hip -o build/h2_abl$M.o &
  done
  wait
  for M in $MASKS; do
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OL build/h2_abl$M.o -o build/libi2r_abl$M.so
    rm -f build/h2_abl$M.o
  done
  exit 0
fi
for M in $MASKS; do
  for SH in "1024,256,256,16,3" "2048,128,128,16,3" "2048,256,256,8,3"; do
    echo "M16_ABL=$M shape [$SH] $(I2R_LIB_PATH=$PWD/build/libi2r_abl$M.so timeout -k 10 100 python3 tools/conv_bench.py --op h2 --split-in --iters 30 --shape $SH 2>&1 | grep 'h2 B')"
  done
done
for SH in "1024,256,256,16,3" "2048,128,128,16,3" "2048,256,256,8,3"; do
  echo "32x32x16 kernel shape [$SH] $(I2R_H2_M16=0 timeout -k 10 100 python3 tools/conv_bench.py --op h2 --split-in --iters 30 --shape $SH 2>&1 | grep 'h2 B')"
done
