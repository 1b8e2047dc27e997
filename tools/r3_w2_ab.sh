#!/bin/bash
# GPU-box script: the sampler with the rows-of-2 form of conv3x3_h2_m16_kernel (the all-padding MFMAs of 2x2 images not
# issued; taken from 128 tiles on) and with I2R_H2_M16_W2OFF=1 (the general form from 256 tiles on, as before),
# alternating on one box.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_w2_ab.log
: > $O
for WL in ${WLS:-B1_weather96 C_electricity96 A_sine24}; do
for W in 0 1 0 1; do
  if [ $W = 1 ]; then export I2R_H2_M16_W2OFF=1; else unset I2R_H2_M16_W2OFF; fi
  timeout -k 10 300 python3 bench.py --workload $WL --steps 6 --warmup 2 --extra none --train-steps 0 --cpu-series 0 > gpurun_out/_s.json 2> gpurun_out/_s.err || { echo "bench failed"; tail -5 gpurun_out/_s.err; exit 1; }
  python3 -c "import json;d=json.loads(open('gpurun_out/_s.json').read().strip().splitlines()[-1]);print('$WL W2OFF=$W', round(d['value'],1),'series/s frac', round(d['roofline']['frac'],4), 'k3 ms', round(d['kernels']['conv_fwd_h2_k3']['ms'],2))" >> $O
done
done
cat $O
