#!/bin/bash
# Round-2 GPU session driver.  Usage (from the authoring container):
#   gpurun --timeout 1500 -- 'bash tools/gpu_r2.sh <tag> <part> [<part> ...]'
# parts: tests ubench bench seq seqhost prof pmc latency
TAG=${1:-r2}
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

if has tests; then
    stamp "pytest -m gpu"
    timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > "$OUT/pytest.log" 2>&1
    stamp "pytest rc=$?"
    tail -25 "$OUT/pytest.log"
fi
if has newtests; then
    stamp "pytest (new files only)"
    timeout ${NEWTESTS_TIMEOUT:-600} python -m pytest ${NEWTESTS:-tests/test_gpu_sequences.py} -m gpu -x -q --durations=10 > "$OUT/pytest_new.log" 2>&1
    stamp "pytest rc=$?"
    tail -40 "$OUT/pytest_new.log"
fi
if has ubench; then
    stamp "VALU issue-cost micro-benchmark"
    (cd tools/ubench && timeout 120 /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w valu_rate.hip -o /tmp/valu_rate && timeout 120 /tmp/valu_rate) > "$OUT/valu_rate.log" 2>&1
    cat "$OUT/valu_rate.log"
    (cd /tmp && timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d "$OUT/ubench_pmc" -- /tmp/valu_rate --quick > "$OUT/ubench_pmc.log" 2>&1)
    python tools/ubench_table.py "$OUT/ubench_pmc" > "$OUT/valu_issue_cost_pmc.txt" 2>&1
    cat "$OUT/valu_issue_cost_pmc.txt"
fi
if has lkprobe; then
    stamp "LK probe: parity subset, two bench lines, VALU count"
    timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_config.py -m gpu -x -q -k "lk or circular or bench_configuration or track_frame" 2>&1 | tail -2
    for i in 1 2; do
        timeout 300 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --validate 0 --sustain 0 --no-replay-leg > "$OUT/probe_bench_$i.json" 2>/dev/null
        python -c "import json; b=json.load(open('$OUT/probe_bench_$i.json')); print('  %.0f fps %.3f ms/step' % (b['value'], b['ms_per_step']), {k: round(v,3) for k,v in b['config']['stage_ms'].items()})"
    done
    (cd /tmp && timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d "$OUT/probe_pmc" -- python "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --validate 0 --sustain 0 --no-replay-leg > /dev/null 2>&1)
    python - <<PY
import glob, csv
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob("$OUT/probe_pmc/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "lk_circular" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
if m:
    print("LK per feature: VALU %.0f SALU %.0f LDS %.1f; cycles/VALU %.2f; launch cycles %.0f" % (
        m["SQ_INSTS_VALU"] / m["SQ_WAVES"], m["SQ_INSTS_SALU"] / m["SQ_WAVES"], m["SQ_INSTS_LDS"] / m["SQ_WAVES"],
        m["GRBM_GUI_ACTIVE"] / 8 * 1024 / m["SQ_INSTS_VALU"], m["GRBM_GUI_ACTIVE"] / 8))
PY
fi
if has bench; then
    stamp "bench (default)"
    timeout 900 python bench.py --steps 20 --warmup 3 > "$OUT/bench.json" 2> "$OUT/bench.err"
    cat "$OUT/bench.json"; tail -3 "$OUT/bench.err"
fi
if has bench374; then
    stamp "bench (reference-default load)"
    timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --workload kitti374 > "$OUT/bench_kitti374.json" 2> "$OUT/bench_kitti374.err"
    cat "$OUT/bench_kitti374.json"; tail -3 "$OUT/bench_kitti374.err"
fi
if has config4; then
    for WL in hd4000 hd4000l4; do
        stamp "bench (config 4: 1920x1080, 4000 points, $WL)"
        timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload $WL --frames 16 --validate 0 > "$OUT/bench_$WL.json" 2> "$OUT/bench_$WL.err"
        cat "$OUT/bench_$WL.json"; tail -3 "$OUT/bench_$WL.err"
    done
fi
if has stageslk; then
    stamp "bench (config 2: circularMatching only on the device)"
    timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --stages lk > "$OUT/bench_stages_lk.json" 2> "$OUT/bench_stages_lk.err"
    cat "$OUT/bench_stages_lk.json"; tail -3 "$OUT/bench_stages_lk.err"
    stamp "bench (detect+full)"
    timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --stages detect+full > "$OUT/bench_detect.json" 2> "$OUT/bench_detect.err"
    cat "$OUT/bench_detect.json"; tail -3 "$OUT/bench_detect.err"
fi
if has seq; then
    for S in 256 64 16 8 1; do
        stamp "bench --mode sequences --seqs $S (reference-default bucketing, pairs resident in HBM)"
        timeout 600 python bench.py --mode sequences --workload kitti374 --seqs $S --steps 40 --warmup 4 --no-cpu-baseline --validate $([ $S = 256 ] && echo 3 || echo 0) > "$OUT/bench_seq_${S}.json" 2> "$OUT/bench_seq_${S}.err"
        cat "$OUT/bench_seq_${S}.json"; tail -3 "$OUT/bench_seq_${S}.err"
    done
    stamp "bench --mode sequences, ~2000 points per frame"
    timeout 600 python bench.py --mode sequences --workload kitti2000 --seqs 256 --steps 30 --warmup 4 --no-cpu-baseline --validate 2 > "$OUT/bench_seq2000_256.json" 2> "$OUT/bench_seq2000_256.err"
    cat "$OUT/bench_seq2000_256.json"; tail -3 "$OUT/bench_seq2000_256.err"
fi
if has seqab; then
    for S in 1 8 64 256; do
        stamp "A/B seq S=$S: second pose stream off"
        VO_POSE2_FRAMES=0 timeout 300 python bench.py --mode sequences --workload kitti374 --seqs $S --steps 40 --warmup 4 --no-cpu-baseline --validate 0 > "$OUT/ab_seq_${S}_onepose.json" 2>/dev/null
        python -c "import json,sys; b=json.load(open('$OUT/ab_seq_${S}_onepose.json')); print('  one pose stream: %.0f fps %.3f ms/step' % (b['value'], b['ms_per_step']), {k: round(v,3) for k,v in b['config']['stage_ms'].items()})"
        stamp "A/B seq S=$S: 512-register pose kernels"
        VO_SEQ_CROWDED_MIN=100000 timeout 300 python bench.py --mode sequences --workload kitti374 --seqs $S --steps 40 --warmup 4 --no-cpu-baseline --validate 0 > "$OUT/ab_seq_${S}_bigpose.json" 2>/dev/null
        python -c "import json,sys; b=json.load(open('$OUT/ab_seq_${S}_bigpose.json')); print('  512-reg pose:    %.0f fps %.3f ms/step' % (b['value'], b['ms_per_step']), {k: round(v,3) for k,v in b['config']['stage_ms'].items()})"
    done
    stamp "A/B batch kitti374: 128-register pose kernels forced"
    VO_CROWDED_MIN=1 VO_CROWDED_MIN_PTS=1 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --workload kitti374 --validate 0 --sustain 0 --no-replay-leg > "$OUT/ab_kitti374_crowded.json" 2>/dev/null
    python -c "import json,sys; b=json.load(open('$OUT/ab_kitti374_crowded.json')); print('  kitti374 crowded: %.0f fps %.3f ms/step' % (b['value'], b['ms_per_step']), {k: round(v,3) for k,v in b['config']['stage_ms'].items()})"
fi
if has posewaves; then
    for WV in 1 2 4; do
        stamp "pose kernels at $WV waves per SIMD"
        VO_POSE_WAVES=$WV timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --workload kitti374 --validate 0 --sustain 0 --no-replay-leg > "$OUT/pw_kitti374_$WV.json" 2>/dev/null
        python -c "import json; b=json.load(open('$OUT/pw_kitti374_$WV.json')); print('  batch kitti374 : %.0f fps %.3f ms/step' % (b['value'], b['ms_per_step']), {k: round(v,3) for k,v in b['config']['stage_ms'].items()})"
        for S in 16 64 256; do
            VO_POSE_WAVES=$WV timeout 300 python bench.py --mode sequences --workload kitti374 --seqs $S --steps 40 --warmup 4 --no-cpu-baseline --validate 0 > "$OUT/pw_seq_${S}_$WV.json" 2>/dev/null
            python -c "import json; b=json.load(open('$OUT/pw_seq_${S}_$WV.json')); print('  seq S=%-4d      : %.0f fps %.3f ms/step' % ($S, b['value'], b['ms_per_step']), {k: round(v,3) for k,v in b['config']['stage_ms'].items()})"
        done
        VO_POSE_WAVES=$WV timeout 300 python bench.py --mode sequences --workload kitti2000 --seqs 256 --steps 30 --warmup 4 --no-cpu-baseline --validate 0 > "$OUT/pw_seq2000_$WV.json" 2>/dev/null
        python -c "import json; b=json.load(open('$OUT/pw_seq2000_$WV.json')); print('  seq2000 S=256   : %.0f fps %.3f ms/step' % (b['value'], b['ms_per_step']), {k: round(v,3) for k,v in b['config']['stage_ms'].items()})"
        VO_POSE_WAVES=$WV timeout 300 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --validate 0 --sustain 0 --no-replay-leg > "$OUT/pw_default_$WV.json" 2>/dev/null
        python -c "import json; b=json.load(open('$OUT/pw_default_$WV.json')); print('  batch kitti2000: %.0f fps %.3f ms/step' % (b['value'], b['ms_per_step']), {k: round(v,3) for k,v in b['config']['stage_ms'].items()})"
    done
fi
if has prepab; then
    for PREP in 0 1; do
        stamp "lock-step loop, VO_SEQ_PREP=$PREP"
        for S in 1 8 64 256; do
            VO_SEQ_PREP=$PREP timeout 300 python bench.py --mode sequences --workload kitti374 --seqs $S --steps 40 --warmup 4 --no-cpu-baseline --validate $([ $S = 256 ] && echo 3 || echo 0) > "$OUT/prep${PREP}_seq_$S.json" 2>"$OUT/prep${PREP}_seq_$S.err" || tail -3 "$OUT/prep${PREP}_seq_$S.err"
            python -c "import json; b=json.load(open('$OUT/prep${PREP}_seq_$S.json')); print('  seq S=%-4d      : %.0f fps %.3f ms/step validated %d' % ($S, b['value'], b['ms_per_step'], b['validated_frames']), {k: round(v,3) for k,v in b['config']['stage_ms'].items()})"
        done
        VO_SEQ_PREP=$PREP timeout 300 python bench.py --mode sequences --workload kitti2000 --seqs 256 --steps 30 --warmup 4 --no-cpu-baseline --validate 2 > "$OUT/prep${PREP}_seq2000.json" 2>/dev/null
        python -c "import json; b=json.load(open('$OUT/prep${PREP}_seq2000.json')); print('  seq2000 S=256   : %.0f fps %.3f ms/step validated %d' % (b['value'], b['ms_per_step'], b['validated_frames']), {k: round(v,3) for k,v in b['config']['stage_ms'].items()})"
        VO_SEQ_PREP=$PREP timeout 300 python bench.py --mode sequences --workload kitti374 --seqs 256 --steps 40 --warmup 4 --no-cpu-baseline --validate 0 --ingest pinned > "$OUT/prep${PREP}_seq_256_pinned.json" 2>/dev/null
        python -c "import json; b=json.load(open('$OUT/prep${PREP}_seq_256_pinned.json')); print('  S=256 pinned    : %.0f fps %.3f ms/step' % (b['value'], b['ms_per_step']))"
    done
fi
if has seqhost; then
    for ING in pinned host; do
        for S in 256 8; do
            stamp "bench --mode sequences --seqs $S --ingest $ING (PCIe-inclusive)"
            timeout 600 python bench.py --mode sequences --workload kitti374 --seqs $S --steps 40 --warmup 4 --no-cpu-baseline --validate 0 --ingest $ING > "$OUT/bench_seq_${S}_${ING}.json" 2> "$OUT/bench_seq_${S}_${ING}.err"
            cat "$OUT/bench_seq_${S}_${ING}.json"; tail -3 "$OUT/bench_seq_${S}_${ING}.err"
        done
    done
fi
if has latency; then
    stamp "latency mode of the drop-in boundary"
    timeout 300 python tools/latency_mode.py 40 > "$OUT/latency.log" 2>&1
    cat "$OUT/latency.log"
fi
if has prof; then
    cd /tmp
    stamp "rocprofv3 kernel trace (overlapped, as benched)"
    timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_overlap" -- python "$ROOT/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --validate 0 --sustain 0 --no-replay-leg > "$OUT/prof_overlap.log" 2>&1
    stamp "rocprofv3 kernel trace (sequence mode)"
    timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_seq" -- python "$ROOT/bench.py" --mode sequences --workload kitti374 --seqs 256 --steps 10 --warmup 2 --no-cpu-baseline --validate 0 > "$OUT/prof_seq.log" 2>&1
    cd "$ROOT"
fi
if has pmc; then
    cd /tmp
    stamp "rocprofv3 pmc SQ pass (LK instruction counts)"
    timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d "$OUT/pmc_sq" -- python "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --validate 0 --sustain 0 --no-replay-leg > "$OUT/pmc_sq.log" 2>&1
    stamp "rocprofv3 pmc FETCH_SIZE"
    timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --validate 0 --sustain 0 --no-replay-leg > "$OUT/pmc_fetch.log" 2>&1
    stamp "rocprofv3 pmc WRITE_SIZE"
    timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --validate 0 --sustain 0 --no-replay-leg > "$OUT/pmc_write.log" 2>&1
    cd "$ROOT"
fi
stamp "done"
du -sh "$OUT"
