#!/bin/bash
# GPU-box script: the sampler as two half-batch graphs on two streams (I2R_DUAL_STREAM_MIN=2048, the default so far)
# against one graph (0), after the rows-of-2 form of conv3x3_h2_m16_kernel (which a 2048-series half batch does not
# reach: 128 tiles at the 2x2 level); and the two-stream form with the 16x16x32 kernel taken from 128 tiles on.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_dual_ab2.log
: > $O
for WL in ${WLS:-B1_weather96 C_electricity96 D_etth1_768}; do
  for CFG in "2048 256" "0 256" "2048 128" "2048 256" "0 256" "2048 128"; do
    set -- $CFG
    export I2R_DUAL_STREAM_MIN=$1 I2R_H2_M16_MINTILES=$2
    timeout -k 10 300 python3 bench.py --gpus 1 --workload $WL --steps 6 --warmup 2 --extra none --train-steps 0 --cpu-series 0 > gpurun_out/_s.json 2> gpurun_out/_s.err || { echo "bench failed"; tail -5 gpurun_out/_s.err; exit 1; }
    python3 -c "import json;d=json.loads(open('gpurun_out/_s.json').read().strip().splitlines()[-1]);print('$WL dual_stream_min=$1 m16_mintiles=$2', round(d['value'],1),'series/s')" >> $O
  done
done
cat $O
