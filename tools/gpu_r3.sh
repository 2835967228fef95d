#!/bin/bash
# Round-3 GPU session driver.  Usage (from the authoring container):
#   gpurun --timeout 2400 -- 'bash tools/gpu_r3.sh <tag> <part> [<part> ...]'
# parts: tests bench seq seqhost prof pmc latency timeline ingest sweep
TAG=${1:-r3}
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
LEAN="--no-cpu-baseline --validate 0 --sustain 0 --no-replay-leg --no-configs"

if has tests; then
    stamp "pytest -m gpu"
    timeout 1500 python -m pytest tests -m gpu -q --durations=10 > "$OUT/pytest.log" 2>&1
    stamp "pytest rc=$?"
    tail -15 "$OUT/pytest.log"
fi
if has bench; then
    stamp "bench (default: headline + exact replay + configs)"
    timeout 900 python bench.py --steps 20 --warmup 3 > "$OUT/bench.json" 2> "$OUT/bench.err"
    tail -c 600 "$OUT/bench.json"; tail -3 "$OUT/bench.err"
    stamp "bench (detect+full)"
    timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --stages detect+full --no-configs --no-replay-leg > "$OUT/bench_detect.json" 2> "$OUT/bench_detect.err"
    tail -c 300 "$OUT/bench_detect.json"
fi
if has seq; then
    for S in 256 64 16 8 1; do
        stamp "bench --mode sequences --seqs $S (reference-default bucketing, pairs resident in HBM)"
        timeout 600 python bench.py --mode sequences --workload kitti374 --seqs $S --steps 40 --warmup 4 --no-cpu-baseline --validate $([ $S = 256 ] && echo 3 || echo 0) > "$OUT/bench_seq_${S}.json" 2> "$OUT/bench_seq_${S}.err"
        python -c "import json; b=json.loads(open('$OUT/bench_seq_${S}.json').read().strip().splitlines()[-1]); print('  S=%-4d %.0f fps %.3f ms/step' % ($S, b['value'], b['ms_per_step']), b['config']['schedule'])"
    done
fi
if has seqhost; then
    for ING in pinned host; do
        for S in 256 8; do
            stamp "bench --mode sequences --seqs $S --ingest $ING (PCIe-inclusive)"
            timeout 600 python bench.py --mode sequences --workload kitti374 --seqs $S --steps 40 --warmup 4 --no-cpu-baseline --validate 0 --ingest $ING > "$OUT/bench_seq_${S}_${ING}.json" 2> "$OUT/bench_seq_${S}_${ING}.err"
            python -c "import json; b=json.loads(open('$OUT/bench_seq_${S}_${ING}.json').read().strip().splitlines()[-1]); print('  S=%-4d $ING %.0f fps %.3f ms/step' % ($S, b['value'], b['ms_per_step']))"
        done
    done
fi
if has latency; then
    stamp "latency mode of the drop-in boundary"
    timeout 300 python tools/latency_mode.py 200 > "$OUT/latency.log" 2>&1
    cat "$OUT/latency.log"
fi
if has timeline; then
    stamp "kernel timeline of vo_track_frame"
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/tf" -- python "$ROOT/tools/latency_mode.py" trackonly 6 60 > "$OUT/tf.log" 2>&1)
    python tools/kernel_timeline.py "$OUT/tf" 52 > "$OUT/timeline.txt" 2>&1
    rm -rf "$OUT/tf"
    tail -30 "$OUT/timeline.txt"
fi
if has ingest; then
    stamp "decode-inclusive ingest (C++ host + python front end, PNG / PGM, 1 and 8 sequences)"
    timeout 900 python tools/ingest_bench.py 600 > "$OUT/ingest.json" 2> "$OUT/ingest.err"
    python -c "
import json
o=json.load(open('$OUT/ingest.json'))
for r in o['runs']: print('  %-34s %s S=%d threads=%-2d %s frames/s' % (r['host'], r['format'], r['sequences'], r['decode_threads'], r['frames_per_s']))"
    tail -3 "$OUT/ingest.err"
fi
if has sweep; then
    stamp "schedule probe against every pinned schedule"
    timeout 1200 python tools/schedule_sweep.py > "$OUT/sweep.jsonl" 2> "$OUT/sweep.err"
    python -c "
import json
for l in open('$OUT/sweep.jsonl'):
    r=json.loads(l); pk=r['probe_pick']
    print('  %-22s probe %7d (w%d s%d p%d)  best %-6s %7d  worst %7d  probe/best %.3f' % (r['config'], r['probe_fps'], pk['pose_waves'], pk['pose_streams'], pk['prepare'], r['best_pinned'], r['best_pinned_fps'], r['worst_pinned_fps'], r['probe_over_best']))"
fi
if has prof; then
    cd /tmp
    stamp "rocprofv3 kernel trace (overlapped, as benched)"
    timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_overlap" -- python "$ROOT/bench.py" --steps 5 --warmup 1 $LEAN > "$OUT/prof_overlap.log" 2>&1
    stamp "rocprofv3 kernel trace (sequence mode)"
    timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_seq" -- python "$ROOT/bench.py" --mode sequences --workload kitti374 --seqs 256 --steps 10 --warmup 2 --no-cpu-baseline --validate 0 > "$OUT/prof_seq.log" 2>&1
    cd "$ROOT"
    # the traces themselves are large: keep the stats and one trace per run for the register / LDS columns
    find "$OUT/prof_overlap" "$OUT/prof_seq" -name "*_kernel_trace.csv" -size +20M -delete 2>/dev/null
fi
if has wide; then
    stamp "pose phases (developer build)"
    VO_HIP_LIB=$ROOT/visual_odom_amd/libvo_hip_dev.so timeout 200 python tools/pose_phases.py 6 14 > "$OUT/pose_phases.txt" 2>&1
    VO_HIP_LIB=$ROOT/visual_odom_amd/libvo_hip_dev.so VO_EPNP_WIDE_MAX=0 timeout 200 python tools/pose_phases.py 6 14 > "$OUT/pose_phases_narrow.txt" 2>&1
    cat "$OUT/pose_phases.txt" "$OUT/pose_phases_narrow.txt"
    for WM in 0 16 64; do
        for S in 1 8 16 32 64; do
            VO_HIP_LIB=$ROOT/visual_odom_amd/libvo_hip_dev.so VO_EPNP_WIDE_MAX=$WM timeout 300 python bench.py --mode sequences --workload kitti374 --seqs $S --steps 60 --warmup 4 --no-cpu-baseline --validate 0 > "$OUT/wide_${WM}_seq_${S}.json" 2> "$OUT/wide_${WM}_seq_${S}.err"
            python -c "import json; b=json.loads(open('$OUT/wide_${WM}_seq_${S}.json').read().strip().splitlines()[-1]); print('  wide_max=%-3d S=%-4d %.0f fps %.3f ms/step' % ($WM, $S, b['value'], b['ms_per_step']), {k: b['config']['schedule'][k] for k in ('pose_waves','pose_streams','prepare')})"
        done
    done
    for WM in 0 16; do
        for B in 8 16 32; do
            VO_HIP_LIB=$ROOT/visual_odom_amd/libvo_hip_dev.so VO_EPNP_WIDE_MAX=$WM timeout 300 python bench.py --workload kitti374 --frames $B --steps 30 --warmup 3 $LEAN > "$OUT/wide_${WM}_batch_${B}.json" 2> "$OUT/wide_${WM}_batch_${B}.err"
            python -c "import json; b=json.loads(open('$OUT/wide_${WM}_batch_${B}.json').read().strip().splitlines()[-1]); print('  wide_max=%-3d batch B=%-4d %.0f fps %.3f ms/step' % ($WM, $B, b['value'], b['ms_per_step']))"
        done
    done
fi
if has pmc; then
    cd /tmp
    stamp "rocprofv3 pmc SQ pass (LK instruction counts)"
    timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d "$OUT/pmc_sq" -- python "$ROOT/bench.py" --steps 3 --warmup 1 $LEAN > "$OUT/pmc_sq.log" 2>&1
    stamp "rocprofv3 pmc FETCH_SIZE"
    timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python "$ROOT/bench.py" --steps 3 --warmup 1 $LEAN > "$OUT/pmc_fetch.log" 2>&1
    stamp "rocprofv3 pmc WRITE_SIZE"
    timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python "$ROOT/bench.py" --steps 3 --warmup 1 $LEAN > "$OUT/pmc_write.log" 2>&1
    cd "$ROOT"
fi
stamp "done"
du -sh "$OUT"
