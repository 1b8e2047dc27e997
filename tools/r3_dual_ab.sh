#!/bin/bash
# two half-batch graphs on two streams (I2R_DUAL_STREAM_MIN=2048, the default of rounds 1-2) against one graph (0)
for WL in A_sine24 C_electricity96 B1_weather96; do
  for DS in 0 2048; do
    I2R_DUAL_STREAM_MIN=$DS timeout -k 10 250 python3 bench.py --gpus 1 --workload $WL --steps 6 --warmup 2 --extra none --train-steps 0 --cpu-series 0 > gpurun_out/r3_dual_${WL}_$DS.json 2> gpurun_out/r3_dual_${WL}_$DS.err
    rc=$?
    echo "$WL dual_stream_min=$DS rc=$rc $(python3 -c "import json,sys; d=json.loads(open('gpurun_out/r3_dual_${WL}_$DS.json').read().strip().splitlines()[-1]); print(round(d['value'],1), 'series/s', round(d['ms_per_step'],1), 'ms/step', d['config'].get('global_batch'))")"
    if [ $rc -ne 0 ]; then exit 1; fi
  done
done
