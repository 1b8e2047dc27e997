#!/bin/bash
# attention-tail chunk size (I2R_ATTN_CHUNK_MB; 0 = whole batch per launch) on the bench workloads
for WL in B1_weather96 D_etth1_768; do
  for MB in 256 0 1024; do
    I2R_ATTN_CHUNK_MB=$MB timeout -k 10 250 python3 bench.py --gpus 1 --workload $WL --steps 6 --warmup 2 --extra none --train-steps 0 --cpu-series 0 > gpurun_out/r3_chunk_${WL}_$MB.json 2> gpurun_out/r3_chunk_${WL}_$MB.err
    rc=$?
    echo "$WL chunk_mb=$MB rc=$rc $(python3 -c "import json,sys; d=json.loads(open('gpurun_out/r3_chunk_${WL}_$MB.json').read().strip().splitlines()[-1]); print(round(d['value'],1), 'series/s', round(d['ms_per_step'],1), 'ms/step eager eval', round(d['eager_eval_ms'],2))")"
    if [ $rc -ne 0 ]; then exit 1; fi
  done
done
