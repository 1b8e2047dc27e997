#!/bin/bash
# SQ counters of the 16x16x32 conv kernels on two bench shapes (one pass per counter set, each its own run)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3pmc
mkdir -p $O
SHAPES="--shape 4096,128,128,16,3 --shape 4096,512,256,8,3 --shape 1024,256,768,8,1"
timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/a -o a -- python3 tools/conv_bench.py --op h2 --split-in --iters 3 $SHAPES > $O/a.log 2>&1 || { echo "pass a failed"; tail -5 $O/a.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VALU --output-format csv -d $O/b -o b -- python3 tools/conv_bench.py --op h2 --split-in --iters 3 $SHAPES > $O/b.log 2>&1 || { echo "pass b failed"; tail -5 $O/b.log; exit 1; }
for P in a b; do
  F=$(find $O/$P -name "*counter_collection.csv" | head -1)
  echo "## pass $P"; python3 tools/pmc_summary.py $F m16
done > $O/r03_pmc_m16_summary.txt
cat $O/r03_pmc_m16_summary.txt
