#!/bin/bash
# developer A/B: register caps of lk_circular_kernel vs throughput
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-lksweep}
mkdir -p "$OUT"; cd "$ROOT" || exit 1
run() {
    touch visual_odom_amd/csrc/lk.hip
    VO_LK_ATTRS="$2" python -m visual_odom_amd.build > "$OUT/build_$1.log" 2>&1 || { echo "build $1 failed"; tail -3 "$OUT/build_$1.log"; return; }
    for i in 1 2; do
        VO_SERIAL_POSE=1 timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline > "$OUT/serial_$1_$i.json" 2>/dev/null
        timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline > "$OUT/overlap_$1_$i.json" 2>/dev/null
    done
    python - "$OUT" "$1" <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/*_" + sys.argv[2] + "_*.json")):
    b = json.loads(open(f).read().strip().splitlines()[-1])
    print("%-22s fps %.0f ms %.3f lk %.3f" % (f.split("/")[-1], b["value"], b["ms_per_step"], b["config"]["stage_ms"]["lk"]))
PY
}
run C "__launch_bounds__(64)"
run A "__launch_bounds__(64) __attribute__((amdgpu_num_sgpr(96)))"
run B "__launch_bounds__(64,8) __attribute__((amdgpu_num_sgpr(80)))"
touch visual_odom_amd/csrc/lk.hip
