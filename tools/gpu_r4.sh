#!/bin/bash
# Round-4 GPU session driver.  Usage (from the authoring container):
#   gpurun --timeout 1500 -- 'bash tools/gpu_r4.sh <tag> <part> [<part> ...]'
# parts: tests slimab slimdev seqab pyrab bench prof latency seqinline constants wideprobe pyrstore pyrrows xcdab timeline pyrpmc pyrprof
# (PYTEST_K=<expr> narrows `tests`; pyrpmc keeps to SQ_* / GRBM_* counters -- TCP_* / TCC_* derived counters hang rocprofv3 on this pool)
TAG=${1:-r4}
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
if has slimab; then   # pinned schedules, product library: the slim pose chain (pose_waves 4) against the round-3 ones
    for WL in kitti2000 kitti374; do
        for SCHED in 2,1,0 2,2,0 4,1,0 4,2,0; do
            stamp "bench $WL --schedule $SCHED"
            timeout 300 python bench.py --workload $WL --steps 20 --warmup 3 $LEAN --validate 2 --schedule $SCHED > "$OUT/ab_${WL}_${SCHED}.json" 2> "$OUT/ab_${WL}_${SCHED}.err"
            line "$OUT/ab_${WL}_${SCHED}.json" "$WL $SCHED"
        done
    done
fi
if has slimdev; then  # register budget of the slim EPnP (developer build)
    export VO_HIP_LIB=$ROOT/visual_odom_amd/libvo_hip_dev.so
    for WL in kitti2000 kitti374; do
        for SW in 4 5 7; do
            stamp "dev: VO_SLIM_WAVES=$SW bench $WL --schedule 4,1,0"
            VO_SLIM_WAVES=$SW timeout 300 python bench.py --workload $WL --steps 20 --warmup 3 $LEAN --validate 0 --schedule 4,1,0 > "$OUT/dev_${WL}_sw${SW}.json" 2> "$OUT/dev_${WL}_sw${SW}.err"
            line "$OUT/dev_${WL}_sw${SW}.json" "$WL slim_waves=$SW"
        done
        for CH in 128; do
            stamp "dev: VO_RANSAC_CHUNK=$CH bench $WL --schedule 4,1,0"
            VO_RANSAC_CHUNK=$CH timeout 300 python bench.py --workload $WL --steps 20 --warmup 3 $LEAN --validate 0 --schedule 4,1,0 > "$OUT/dev_${WL}_ch${CH}.json" 2> "$OUT/dev_${WL}_ch${CH}.err"
            line "$OUT/dev_${WL}_ch${CH}.json" "$WL chunk=$CH"
        done
    done
    unset VO_HIP_LIB
fi
if has seqab; then    # lock-step loop, 256 sequences: does the prepare stream pay once the pose chain is slim?
    for WL in kitti374 kitti2000; do
        for SCHED in 2,2,0 2,2,1 4,2,0 4,2,1 4,1,1; do
            stamp "bench --mode sequences $WL --seqs 256 --schedule $SCHED"
            timeout 300 python bench.py --mode sequences --workload $WL --seqs 256 --steps 40 --warmup 4 --no-cpu-baseline --validate 2 --schedule $SCHED > "$OUT/seq_${WL}_${SCHED}.json" 2> "$OUT/seq_${WL}_${SCHED}.err"
            python -c "import json; b=json.loads(open('$OUT/seq_${WL}_${SCHED}.json').read().strip().splitlines()[-1]); print('  seq256 $WL $SCHED %.0f fps %.3f ms/step' % (b['value'], b['ms_per_step']), b.get('validated_frames'))" 2>&1 | tee -a "$OUT/summary.txt"
        done
    done
fi
if has pyrab; then    # fused pyramid passes against the three-kernel chain (developer build), rows per work item
    for WL in kitti2000 kitti374; do
        for V in "dev 0" "dev 1" "pf4 1" "pf16 1"; do
            set -- $V
            stamp "lib=$1 VO_PYR_FUSED=$2 bench $WL --schedule 2,1,0"
            VO_HIP_LIB=$ROOT/visual_odom_amd/libvo_hip_$1.so VO_PYR_FUSED=$2 timeout 300 python bench.py --workload $WL --steps 20 --warmup 3 $LEAN --validate 2 --schedule 2,1,0 > "$OUT/pyr_${WL}_$1_$2.json" 2> "$OUT/pyr_${WL}_$1_$2.err"
            line "$OUT/pyr_${WL}_$1_$2.json" "$WL lib=$1 fused=$2"
        done
    done
    for V in "dev 0" "dev 1"; do
        set -- $V
        stamp "lib=$1 VO_PYR_FUSED=$2 bench --stages lk"
        VO_HIP_LIB=$ROOT/visual_odom_amd/libvo_hip_$1.so VO_PYR_FUSED=$2 timeout 300 python bench.py --stages lk --steps 20 --warmup 3 $LEAN --validate 2 > "$OUT/pyr_lk_$1_$2.json" 2> "$OUT/pyr_lk_$1_$2.err"
        line "$OUT/pyr_lk_$1_$2.json" "stages=lk lib=$1 fused=$2"
        stamp "lib=$1 VO_PYR_FUSED=$2 latency mode"
        VO_HIP_LIB=$ROOT/visual_odom_amd/libvo_hip_$1.so VO_PYR_FUSED=$2 timeout 300 python tools/latency_mode.py 200 > "$OUT/latency_$1_$2.log" 2>&1
        grep -i "track_frame\|ms" "$OUT/latency_$1_$2.log" | head -12
    done
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
    find "$OUT/prof" -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_trace_batch.csv"
    rm -rf "$OUT/prof"
    head -12 "$OUT/kernel_stats_batch.csv" | cut -c1-200
fi
if has latency; then
    stamp "latency mode of the drop-in boundary"
    timeout 300 python tools/latency_mode.py 200 > "$OUT/latency.log" 2>&1
    cat "$OUT/latency.log"
fi
if has seqinline; then  # (round-4 experiment, removed from the code: filter + carry on the tracking stream -- no gain at 1 sequence, -1..-10 % at 8 / 64; profiles/r04_experiments.md)
    export VO_HIP_LIB=$ROOT/visual_odom_amd/libvo_hip_dev.so
    for INL in 0 1; do
        for S in 1 8 64 256; do
            for WL in kitti374 kitti2000; do
                stamp "VO_SEQ_INLINE_FILTER=$INL bench --mode sequences $WL --seqs $S"
                VO_SEQ_INLINE_FILTER=$INL timeout 300 python bench.py --mode sequences --workload $WL --seqs $S --steps $([ $S -le 8 ] && echo 300 || echo 60) --warmup 4 --no-cpu-baseline --validate 0 > "$OUT/inl${INL}_${WL}_${S}.json" 2> "$OUT/inl${INL}_${WL}_${S}.err"
                python -c "import json; b=json.loads(open('$OUT/inl${INL}_${WL}_${S}.json').read().strip().splitlines()[-1]); print('  inline=$INL $WL S=%-4d %.0f fps %.3f ms/step' % ($S, b['value'], b['ms_per_step']), {k: b['config']['schedule'][k] for k in ('pose_waves','pose_streams','prepare')})" 2>&1 | tee -a "$OUT/summary.txt"
            done
        done
    done
    unset VO_HIP_LIB
fi
if has constants; then  # the two launch-size constants left in the pose chain, at the other camera shapes (developer build switches)
    export VO_HIP_LIB=$ROOT/visual_odom_amd/libvo_hip_dev.so
    for WL in zed374 rgbd374 hd4000; do
        FR=$([ $WL = hd4000 ] && echo 128 || echo 256)
        for CH in 128 64 32; do
            stamp "VO_RANSAC_CHUNK=$CH bench $WL --frames $FR"
            VO_RANSAC_CHUNK=$CH timeout 300 python bench.py --workload $WL --frames $FR --quads 4 --steps 12 --warmup 2 $LEAN --validate 0 > "$OUT/chunk${CH}_${WL}.json" 2> "$OUT/chunk${CH}_${WL}.err"
            python -c "import json; b=json.loads(open('$OUT/chunk${CH}_${WL}.json').read().strip().splitlines()[-1]); print('  first chunk $CH  $WL x $FR frames  %.0f fps %.3f ms/step' % (b['value'], b['ms_per_step']))" 2>&1 | tee -a "$OUT/summary.txt"
        done
    done
    for WL in zed374 rgbd374; do
        for SM in 4 16; do
            for S in 2 4 8 16; do
                stamp "VO_EPNP_SPLIT_MAX=$SM bench --mode sequences $WL --seqs $S"
                VO_EPNP_SPLIT_MAX=$SM timeout 300 python bench.py --mode sequences --workload $WL --seqs $S --quads 4 --steps 200 --warmup 4 --no-cpu-baseline --validate 0 > "$OUT/split${SM}_${WL}_${S}.json" 2> "$OUT/split${SM}_${WL}_${S}.err"
                python -c "import json; b=json.loads(open('$OUT/split${SM}_${WL}_${S}.json').read().strip().splitlines()[-1]); print('  four-kernel EPnP up to $SM frames  $WL S=%-3d %.0f fps %.3f ms/step' % ($S, b['value'], b['ms_per_step']))" 2>&1 | tee -a "$OUT/summary.txt"
            done
        done
    done
    unset VO_HIP_LIB
fi
if has wideprobe; then  # the probe with the four-kernel EPnP as one of its knobs, where the constant had lost (r4_11)
    for WL in rgbd374 zed374 kitti374; do
        for S in 8 16; do
            stamp "bench --mode sequences $WL --seqs $S (probed)"
            timeout 300 python bench.py --mode sequences --workload $WL --seqs $S --quads 4 --steps 200 --warmup 4 --no-cpu-baseline --validate 2 > "$OUT/wp_${WL}_${S}.json" 2> "$OUT/wp_${WL}_${S}.err"
            python -c "import json; b=json.loads(open('$OUT/wp_${WL}_${S}.json').read().strip().splitlines()[-1]); print('  probed  $WL S=%-3d %.0f fps %.3f ms/step' % ($S, b['value'], b['ms_per_step']), b['config']['schedule'], 'val', b['validated_frames'])" 2>&1 | tee -a "$OUT/summary.txt"
        done
    done
fi
if has pyrstore; then  # what bounds the level-0 pass: Scharr stores non-temporal / ordinary / none (developer build, no pose chain)
    for SM in 0 1 2; do
        stamp "VO_PYR_STORE=$SM kernel trace, bench --stages lk"
        (cd /tmp && VO_PYR_STORE=$SM VO_HIP_LIB=$ROOT/visual_odom_amd/libvo_hip_dev.so timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/ps$SM" -- python "$ROOT/bench.py" --stages lk --steps 6 --warmup 2 $LEAN --validate 0 > "$OUT/ps$SM.log" 2>&1)
        python - "$OUT/ps$SM" $SM <<'PYEOF2' | tee -a "$OUT/summary.txt"
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
by = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "pyr_pass" in r["Kernel_Name"]:
        by[int(r["Grid_Size_X"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("  store mode %s: " % sys.argv[2] + "  ".join("grid %d: avg %.0f min %.0f us" % (g, sum(v) / len(v), min(v)) for g, v in sorted(by.items(), reverse=True)))
PYEOF2
        rm -rf "$OUT/ps$SM"
    done
fi
if has xcdab; then   # XCD-aware workgroup order of the pyramid pass against dispatch order, in the pipeline (developer build: VO_PYR_XCD)
    for WL in kitti2000 kitti374 rgbd374 zed374 hd4000; do
        for X in 0 1; do
            stamp "VO_PYR_XCD=$X bench $WL"
            FR=""; [ $WL = hd4000 ] && FR="--frames 128"
            VO_HIP_LIB=$ROOT/visual_odom_amd/libvo_hip_dev.so VO_PYR_XCD=$X timeout 300 python bench.py --workload $WL $FR --steps 20 --warmup 3 $LEAN --validate 0 --schedule 1,1,0 > "$OUT/xcd_${WL}_$X.json" 2> "$OUT/xcd_${WL}_$X.err"
            line "$OUT/xcd_${WL}_$X.json" "$WL xcd=$X"
        done
    done
fi
if has pyrrows; then   # rows per work item of the fused pass (libvo_hip_pf4 / _dev (8) / _pf16: pyramid.hip built with -DVO_PF_ROWS=)
    for LIB in pf4 dev pf16; do
        [ -f "$ROOT/visual_odom_amd/libvo_hip_$LIB.so" ] || continue
        stamp "lib=$LIB kernel trace, bench --stages lk"
        (cd /tmp && VO_HIP_LIB=$ROOT/visual_odom_amd/libvo_hip_$LIB.so timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/pr$LIB" -- python "$ROOT/bench.py" --stages lk --steps 6 --warmup 2 $LEAN --validate 2 > "$OUT/pr$LIB.log" 2>&1)
        python - "$OUT/pr$LIB" $LIB <<'PYEOF3' | tee -a "$OUT/summary.txt"
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
by = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "pyr_pass" in r["Kernel_Name"]:
        by[int(r["Grid_Size_X"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("  lib %s: " % sys.argv[2] + "  ".join("grid %d: avg %.0f min %.0f us" % (g, sum(v) / len(v), min(v)) for g, v in sorted(by.items(), reverse=True)))
PYEOF3
        rm -rf "$OUT/pr$LIB"
        tail -c 300 "$OUT/pr$LIB.log" | grep -o '"validated_frames": [0-9]*' | tee -a "$OUT/summary.txt"
    done
fi
if has timeline; then
    stamp "kernel timeline of vo_track_frame"
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/tf" -- python "$ROOT/tools/latency_mode.py" trackonly 6 60 > "$OUT/tf.log" 2>&1)
    python tools/kernel_timeline.py "$OUT/tf" 52 > "$OUT/timeline.txt" 2>&1
    rm -rf "$OUT/tf"
    tail -40 "$OUT/timeline.txt"
    stamp "kernel timeline of the one-sequence lock-step loop"
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/sq" -- python "$ROOT/tools/latency_mode.py" pipelined 6 30 > "$OUT/sq.log" 2>&1)
    python tools/kernel_timeline.py "$OUT/sq" 120 > "$OUT/timeline_seq.txt" 2>&1
    rm -rf "$OUT/sq"
    tail -60 "$OUT/timeline_seq.txt"
fi
if has pyrpmc; then   # what the fused pass waits for: SQ counters per level (bench --stages lk: no pose chain beside
                      # it; developer build, VO_PYR_STORE 0 = product stores, 2 = no Scharr stores)
    for SM in ${PYR_SM:-0 2}; do
    n=0
    for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
               "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"; do
        n=$((n + 1))
        stamp "store mode $SM pmc set $n: $SET"
        (cd /tmp && VO_PYR_STORE=$SM VO_HIP_LIB=$ROOT/visual_odom_amd/libvo_hip_dev.so timeout 300 rocprofv3 --pmc $SET --output-format csv -d "$OUT/pp$n" -- python "$ROOT/bench.py" --stages lk --steps 3 --warmup 1 $LEAN --validate 0 > "$OUT/pp${SM}_$n.log" 2>&1)
        python - "$OUT/pp$n" <<'PYEOF4' | tee -a "$OUT/summary.txt"
import csv, glob, sys, collections
fs = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not fs:
    print("  (no counter file)")
    sys.exit(0)
by = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fs[0])):
    if "pyr_pass" in r["Kernel_Name"]:
        by[int(r["Grid_Size"]) if "Grid_Size" in r else 0][r["Counter_Name"]].append(float(r["Counter_Value"]))
for g in sorted(by, reverse=True)[:2]:
    print("  grid %d: " % g + "  ".join("%s %.4g" % (k, sum(v) / len(v)) for k, v in sorted(by[g].items())))
PYEOF4
        rm -rf "$OUT/pp$n"
    done
    done
fi
if has pyrprof; then  # kernel-level split of the pyramid stage (developer build; VO_PYR_FUSED from the environment)
    stamp "rocprofv3 kernel stats, dev lib, VO_PYR_FUSED=${VO_PYR_FUSED:-default}"
    (cd /tmp && VO_HIP_LIB=$ROOT/visual_odom_amd/libvo_hip_dev.so timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/pyrprof" -- python "$ROOT/bench.py" --workload kitti374 --steps 20 --warmup 3 $LEAN --validate 0 --schedule 2,1,0 > "$OUT/pyrprof.log" 2>&1)
    find "$OUT/pyrprof" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats_pyr.csv"
    rm -rf "$OUT/pyrprof"
    head -14 "$OUT/kernel_stats_pyr.csv" | cut -c1-60,100-
fi
stamp "done"
