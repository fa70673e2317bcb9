cd /tmp
F="--steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-dense --no-prof --no-isolated"
run() { python3 $GRAFT_REPO_ROOT/bench.py $F "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for l in 3 2 4 3; do echo "lanes $l: $(DLV_LANES=$l run)"; done
for b in 16 24 32 16 48 12; do echo "lanes 3 sw_batch $b: $(run --sw-batch $b)"; done
for cfg in "2 32" "4 12" "2 24"; do set -- $cfg; echo "lanes $1 sw_batch $2: $(DLV_LANES=$1 run --sw-batch $2)"; done
