#!/bin/bash
# developer sweep: register budget of the pose kernels vs overlapped throughput
TAG=${1:-sweep}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT" || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1; tail -3 "$OUT/pytest.log"
for W in 1 2 4; do
    touch visual_odom_amd/csrc/pnp.hip
    VO_PNP_WAVES=$W python -m visual_odom_amd.build > "$OUT/build_$W.log" 2>&1 || { echo "build $W failed"; tail -5 "$OUT/build_$W.log"; continue; }
    timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline > "$OUT/bench_w$W.json" 2> "$OUT/bench_w$W.err"
    VO_SERIAL_POSE=1 timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline > "$OUT/bench_serial_w$W.json" 2> "$OUT/bench_serial_w$W.err"
    python - "$OUT/bench_w$W.json" "$OUT/bench_serial_w$W.json" $W <<'PY'
import json, sys
for f in sys.argv[1:3]:
    try:
        b = json.loads(open(f).read().strip().splitlines()[-1])
        print("W=%s %-28s fps %.0f ms %.3f" % (sys.argv[3], f.split('/')[-1], b["value"], b["ms_per_step"]), {k: round(v, 3) for k, v in b["config"]["stage_ms"].items()})
    except Exception as e:
        print(f, "ERR", e)
PY
done
