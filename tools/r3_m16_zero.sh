for SH in "1024,256,256,16,3" "2048,128,128,16,3"; do
 for Z in "" "--zeros"; do
  for M in 0 1 31; do
   echo "ABL=$M $Z [$SH] $(I2R_LIB_PATH=$PWD/build/libi2r_abl$M.so timeout -k 10 100 python3 tools/conv_bench.py --op h2 --split-in --iters 30 $Z --shape $SH 2>&1 | grep 'h2 B' | cut -c40-)"
  done
 done
done
