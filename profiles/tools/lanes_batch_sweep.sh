cd /tmp
F="--steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-dense --no-prof --no-isolated"
for l in 3 2 4 1 3; do echo "lanes $l: $(DLV_LANES=$l python3 $GRAFT_REPO_ROOT/bench.py $F 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")"; done
for b in 8 24 32; do echo "lanes 3 sw_batch $b: $(python3 $GRAFT_REPO_ROOT/bench.py $F --sw-batch $b 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")"; done
