#!/bin/bash
# Round-5 GPU session driver.  Usage (from the authoring container):
#   gpurun --timeout 1500 -- 'bash tools/gpu_r5.sh <tag> <part> [<part> ...]'
# parts: tests latency phases timeline bench prof pyrab quads pmclegs
TAG=${1:-r5}
shift
PARTS="$*"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT" || exit 1
export TMPDIR=/tmp
t0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a "$OUT/timeline.log"; }
has() { [[ " $PARTS " == *" $1 "* ]]; }
LEAN="--no-cpu-baseline --sustain 0 --no-replay-leg --no-configs"
line() { python -c "import json,sys; b=json.loads(open('$1').read().strip().splitlines()[-1]); print('  $2 %.0f fps %.3f ms/step lk %.3f' % (b['value'], b['ms_per_step'], b['roofline']['launch_ms']), {k: round(v,2) for k,v in b['config'].get('stage_ms',{}).items()}, b['config'].get('schedule'), 'val', b.get('validated_frames'))" 2>&1 | tee -a "$OUT/summary.txt"; }

if has tests; then
    stamp "pytest -m gpu $PYTEST_K"
    timeout 1500 python -m pytest tests -m gpu -q --durations=10 ${PYTEST_K:+-k "$PYTEST_K"} > "$OUT/pytest.log" 2>&1
    stamp "pytest rc=$?"
    tail -15 "$OUT/pytest.log"
fi
if has latency; then
    stamp "latency mode of the drop-in boundary"
    timeout 300 python tools/latency_mode.py 200 > "$OUT/latency.log" 2>&1
    cat "$OUT/latency.log"
fi
if has phases; then
    stamp "pose phases (developer build time stamps)"
    VO_HIP_LIB=$ROOT/visual_odom_amd/libvo_hip_dev.so timeout 300 python tools/pose_phases.py 6 14 > "$OUT/pose_phases.txt" 2>&1
    cat "$OUT/pose_phases.txt"
    VO_HIP_LIB=$ROOT/visual_odom_amd/libvo_hip_dev.so timeout 300 python tools/pose_phases.py 1 14 > "$OUT/pose_phases_340.txt" 2>&1
    tail -3 "$OUT/pose_phases_340.txt"
fi
if has timeline; then
    stamp "kernel timeline of vo_track_frame"
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/tf" -- python "$ROOT/tools/latency_mode.py" trackonly 6 60 > "$OUT/tf.log" 2>&1)
    python tools/kernel_timeline.py "$OUT/tf" 52 > "$OUT/timeline.txt" 2>&1
    rm -rf "$OUT/tf"
    tail -40 "$OUT/timeline.txt"
fi
if has bench; then
    stamp "bench (default: headline + exact replay + configs)"
    timeout 900 python bench.py --steps 20 --warmup 3 > "$OUT/bench.json" 2> "$OUT/bench.err"
    tail -c 600 "$OUT/bench.json"; tail -3 "$OUT/bench.err"
fi
if has prof; then
    stamp "rocprofv3 --kernel-trace --stats of the default headline (lean)"
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- python "$ROOT/bench.py" --steps 20 --warmup 3 $LEAN --validate 0 > "$OUT/prof.log" 2>&1)
    find "$OUT/prof" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats_batch.csv"
    rm -rf "$OUT/prof"
    head -12 "$OUT/kernel_stats_batch.csv" | cut -c1-200
fi
stamp "done"
