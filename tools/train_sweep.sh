set -e
run() { echo "== $*"; env "$@" timeout -k 10 120 python bench.py --mode train --steps 20 --warmup 2 --train-mode graph $ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['last_loss'])"; }
ARGS=""
run X=1
run I2R_WGRAD_BK=32
run I2R_WGRAD_BK=32 I2R_WGRAD_TARGET=2048
run I2R_WGRAD_BK=32 I2R_WGRAD_TARGET=512
ARGS="--workload D_etth1_768"
run X=1
run I2R_WGRAD_BK=32
