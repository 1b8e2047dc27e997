set -e
run() { echo "== $*"; env "$@" timeout -k 10 120 python bench.py --mode train --steps 20 --warmup 2 --train-mode graph $ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['last_loss'])"; }
ARGS=""
run X=1
run I2R_WGRAD_MAXSPLIT=64
run I2R_WGRAD_MAXSPLIT=32
run I2R_WGRAD_MAXSPLIT=16
run I2R_WGRAD_MAXSPLIT=8
