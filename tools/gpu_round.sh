#!/bin/bash
# THE GPU session driver (one script; rounds 2-5 had one generation each).  Usage (from the authoring container):
#   gpurun --timeout 1500 -- 'bash tools/gpu_round.sh <tag> <part> [<part> ...]'
# parts: tests latency phases timeline bench prof seq pmclegs posepmc quads ranks8 hunt ingestab ingestdev schedab splitab probereps runahead
# Everything lands in gpurun_out/<tag>/; tools/profile_summary.py / tools/pmc_legs.py / tools/pose_pmc.py turn it into profiles/.
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
if has seq; then      # the lock-step loop at the reference-default load, pairs resident in HBM, and PCIe-inclusive at 256 / 8 sequences
    for S in 256 64 16 8 1; do
        stamp "bench --mode sequences --seqs $S"
        timeout 600 python bench.py --mode sequences --workload kitti374 --seqs $S --steps 40 --warmup 4 --no-cpu-baseline ${QUADS:+--quads $QUADS} --validate $([ $S = 256 ] && echo 3 || echo 0) > "$OUT/bench_seq_${S}.json" 2> "$OUT/bench_seq_${S}.err"
        python -c "import json; b=json.loads(open('$OUT/bench_seq_${S}.json').read().strip().splitlines()[-1]); print('  S=%-4d %.0f fps %.3f ms/step' % ($S, b['value'], b['ms_per_step']), b['config']['schedule'])" 2>&1 | tee -a "$OUT/summary.txt"
    done
    for ING in pinned host; do
        for S in 256 8; do
            stamp "bench --mode sequences --seqs $S --ingest $ING (PCIe-inclusive)"
            timeout 600 python bench.py --mode sequences --workload kitti374 --seqs $S --steps 40 --warmup 4 --no-cpu-baseline --validate 0 --ingest $ING > "$OUT/bench_seq_${S}_${ING}.json" 2> "$OUT/bench_seq_${S}_${ING}.err"
            python -c "import json; b=json.loads(open('$OUT/bench_seq_${S}_${ING}.json').read().strip().splitlines()[-1]); print('  S=%-4d $ING %.0f fps %.3f ms/step' % ($S, b['value'], b['ms_per_step']))" 2>&1 | tee -a "$OUT/summary.txt"
        done
    done
fi
if has pmclegs; then  # PMC passes of every bench leg's LK launch (tools/pmc_legs.py turns them into profiles/lk_traffic.json / lk_issue.json)
    for WL in ${PMC_WL:-kitti2000 kitti374 hd4000 hd4000l4 replay2000}; do
        FR=256; Q=""
        case $WL in hd4000*) FR=128; Q="--quads 4";; esac
        CMD="--workload $WL --frames $FR $Q --steps 3 --warmup 1 $LEAN --validate 0"
        case $WL in replay*) CMD="--mode sequences --workload kitti${WL#replay} --seqs 256 --steps 20 --warmup 4 --no-cpu-baseline --validate 0";; esac   # the exact replay (lock-step loop, pairs resident)
        stamp "bench $WL x $FR (plain: the leg's line)"
        timeout 300 python bench.py $CMD > "$OUT/pmc_$WL.json" 2> "$OUT/pmc_$WL.err"
        for SET in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "sq:SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES"; do
            NAME=${SET%%:*}; CNT=${SET#*:}
            stamp "pmc $NAME: $WL"
            (cd /tmp && timeout 400 rocprofv3 --pmc $CNT --output-format csv -d "$OUT/pmc_${WL}_$NAME" -- python "$ROOT/bench.py" $CMD > "$OUT/pmc_${WL}_$NAME.log" 2>&1)
            # keep only the rows of the two kernels the summary reads (the raw files are tens of MB)
            for f in $(find "$OUT/pmc_${WL}_$NAME" -name "*_counter_collection.csv"); do
                (head -1 "$f"; grep -E "lk_circular_kernel|pyr_pass_kernel" "$f") > "$f.tmp" && mv "$f.tmp" "$f"
            done
            find "$OUT/pmc_${WL}_$NAME" -type f ! -name "*_counter_collection.csv" -delete
        done
    done
    python tools/pmc_legs.py "$OUT" 2>&1 | tee -a "$OUT/summary.txt"
    cp profiles/lk_traffic.json profiles/lk_issue.json "$OUT/"
fi
if has posepmc; then  # counters of every kernel behind LK in the config-4 legs -> tools/pose_pmc.py -> profiles/r06_pose_pmc.md
    for WL in ${POSE_WL:-hd4000 kitti2000}; do
        FR=256; Q=""
        case $WL in hd4000*) FR=128; Q="--quads 4";; esac
        stamp "bench $WL x $FR (plain)"
        timeout 300 python bench.py --workload $WL --frames $FR $Q --steps 3 --warmup 1 $LEAN --validate 0 > "$OUT/pose_$WL.json" 2> "$OUT/pose_$WL.err"
        (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/pose_${WL}_trace" -- python "$ROOT/bench.py" --workload $WL --frames $FR $Q --steps 3 --warmup 1 $LEAN --validate 0 > "$OUT/pose_${WL}_trace.log" 2>&1)
        find "$OUT/pose_${WL}_trace" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/pose_${WL}_kernel_stats.csv"
        find "$OUT/pose_${WL}_trace" -name "*kernel_trace.csv" | head -1 | xargs -I{} python tools/pose_pmc.py --trace {} "$OUT/pose_${WL}_resources.csv"
        rm -rf "$OUT/pose_${WL}_trace"
        for SET in "a:SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
                   "b:SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_FLAT SQ_INSTS_SMEM" \
                   "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
            NAME=${SET%%:*}; CNT=${SET#*:}
            stamp "pose pmc $NAME: $WL"
            (cd /tmp && timeout 400 rocprofv3 --pmc $CNT --output-format csv -d "$OUT/posepmc_${WL}_$NAME" -- python "$ROOT/bench.py" --workload $WL --frames $FR $Q --steps 3 --warmup 1 $LEAN --validate 0 > "$OUT/posepmc_${WL}_$NAME.log" 2>&1)
            for f in $(find "$OUT/posepmc_${WL}_$NAME" -name "*_counter_collection.csv"); do
                (head -1 "$f"; grep -v -E "lk_circular_kernel|pyr_pass_kernel|fast_|bucket_kernel" "$f" | tail -n +2) > "$f.tmp" && mv "$f.tmp" "$f"
            done
            find "$OUT/posepmc_${WL}_$NAME" -type f ! -name "*_counter_collection.csv" -delete
        done
    done
    python tools/pose_pmc.py "$OUT" 2>&1 | tee "$OUT/pose_pmc.md"
fi
if has ranks8; then   # eight REAL ranks on this one GPU (gloo): the N > 1 code path incl. the config-5 leg
    stamp "bench --gpus 8 on GPU 0"
    VO_ALLOW_SHARED_GPU=1 VO_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 8 --frames 32 --steps 3 --warmup 1 --no-cpu-baseline --sustain 0 > "$OUT/bench_ranks8.json" 2> "$OUT/bench_ranks8.err"
    tail -c 1500 "$OUT/bench_ranks8.json"; tail -3 "$OUT/bench_ranks8.err"
fi
if has quads; then   # the headline against the number of distinct rendered quadruples and the world seed (one process: tools/quads_table.py)
    stamp "quads table"
    timeout 1500 python tools/quads_table.py ${QMAX:-128} 2> "$OUT/quads_table.err" | tee "$OUT/quads_table.txt"
fi
if has hunt; then    # longer, differently seeded runs of every fuzz of the GPU suite (HUNT_SEEDS, default "1 2 3")
    for SEED in ${HUNT_SEEDS:-1 2 3}; do
        for T in "tests/test_gpu_round6.py track_frame_fuzz 2000" "tests/test_gpu_round6.py detect_bucket_fuzz 5000" "tests/test_gpu_round6.py pnp_ransac_fuzz 5000" \
                 "tests/test_gpu_round6.py essential_pose_fuzz 3000" "tests/test_gpu_batch_fuzz.py random_batches 2000" \
                 "tests/test_gpu_round6.py kept_pair_call_fuzz 20000"; do
            [ -n "$HUNT_ONLY" ] && [[ "$T" != *"$HUNT_ONLY"* ]] && continue
            set -- $T
            stamp "hunt $2 seed $SEED x $3"
            VO_FUZZ_EXAMPLES=$3 VO_FUZZ_SEED=$SEED timeout 900 python -m pytest $1 -m gpu -q -x -s -k "$2" > "$OUT/hunt_$2_$SEED.log" 2>&1
            grep -E "fuzz:|passed|failed" "$OUT/hunt_$2_$SEED.log" | tee -a "$OUT/summary.txt"
        done
        stamp "hunt random feeds of the lock-step loop, seed $SEED x 40"
        VO_FEED_HUNT=40 VO_FEED_SEED=$SEED timeout 900 python -m pytest tests/test_gpu_sequences.py -m gpu -q -x -k random_feed_hunt > "$OUT/hunt_feed_$SEED.log" 2>&1
        tail -1 "$OUT/hunt_feed_$SEED.log" | tee -a "$OUT/summary.txt"
    done
fi
if has ingestab; then   # lock-step loop: page-locked / pageable / resident pairs side by side (profiles/r06_ingest_ab.txt)
    for WL in kitti2000 kitti374; do for ING in pinned host device; do
        python bench.py --mode sequences --workload $WL --seqs 256 --steps 30 --warmup 4 --no-cpu-baseline --validate 2 --ingest $ING > "$OUT/b_${WL}_${ING}.json" 2>/dev/null
        python -c "import json; b=json.loads(open('$OUT/b_${WL}_${ING}.json').read().strip().splitlines()[-1]); print('$WL $ING  %.0f fps %.2f ms val %d' % (b['value'], b['ms_per_step'], b['validated_frames']), {k: round(v,2) for k,v in b['config']['stage_ms'].items()}, b['config']['schedule']['prepare'])" | tee -a "$OUT/summary.txt"
    done; done
    for S in 8 64; do for ING in pinned host device; do
        python bench.py --mode sequences --workload kitti374 --seqs $S --steps 60 --warmup 4 --no-cpu-baseline --validate 0 --ingest $ING > "$OUT/b_S${S}_${ING}.json" 2>/dev/null
        python -c "import json; b=json.loads(open('$OUT/b_S${S}_${ING}.json').read().strip().splitlines()[-1]); print('kitti374 S=$S $ING  %.0f fps %.3f ms' % (b['value'], b['ms_per_step']))" | tee -a "$OUT/summary.txt"
    done; done
fi
if has ingestdev; then  # developer-build A/B of the PCIe ingest inside the loop, schedule pinned (profiles/r06_experiments.md section 1):
                        # INGEST_AB=wait: VO_INGEST_WAIT 0 / 1 (at once / behind the running step's detection); else VO_INGEST_WAVES = the persistent grid
    devrun() { # name, env..., bench args in $ARGS
        local name=$1; shift
        env VO_HIP_LIB=$ROOT/visual_odom_amd/libvo_hip_dev.so "$@" python bench.py --mode sequences --steps 30 --warmup 4 --no-cpu-baseline --validate 0 $ARGS > "$OUT/b_$name.json" 2>/dev/null
        python -c "import json; b=json.loads(open('$OUT/b_$name.json').read().strip().splitlines()[-1]); print('$name  %.0f fps %.3f ms' % (b['value'], b['ms_per_step']), {k: round(v,2) for k,v in b['config']['stage_ms'].items()})" | tee -a "$OUT/summary.txt"
    }
    if [ "$INGEST_AB" = "wait" ]; then
        for SCH in 2,1,0 2,2,0 1,1,0; do for WL in kitti2000 kitti374; do for S in 256 64 8; do for W in 0 1; do
            ARGS="--workload $WL --seqs $S --ingest pinned --schedule $SCH"; devrun "wait${W}_${WL}_S${S}_${SCH}" VO_INGEST_WAIT=$W
        done; done; done; done
    else
        for G in 64 128 192 256 384 512; do for WL in kitti2000 kitti374; do
            ARGS="--workload $WL --seqs 256 --ingest pinned --schedule 2,1,0"; devrun "G${G}_${WL}" VO_INGEST_WAVES=$G
        done; done
    fi
fi
if has schedab; then    # pinned schedules against the probe's pick on ONE box: SCHED_S sequences, SCHED_LIST "w,s,p ... probe", SCHED_WL, SCHED_REPS, SCHED_INGEST
    S=${SCHED_S:-256}; WL=${SCHED_WL:-kitti374}
    for rep in $(seq 1 ${SCHED_REPS:-3}); do for sc in ${SCHED_LIST:-1,2,1 2,2,1 probe}; do
        f="$OUT/ab_${S}_${sc//,/}_$rep.json"
        timeout 300 python bench.py --mode sequences --workload $WL --seqs $S --steps 60 --warmup 6 --no-cpu-baseline --validate 0 ${SCHED_INGEST:+--ingest $SCHED_INGEST} $([ "$sc" = probe ] || echo --schedule $sc) > "$f" 2> "$f.err"
        python -c "import json; b=json.loads(open('$f').read().strip().splitlines()[-1]); s=b['config']['schedule']; print('S=$S $WL rep $rep %-6s %8.0f fps %.3f ms/step  ran %s,%s,%s' % ('$sc', b['value'], b['ms_per_step'], s['pose_waves'], s['pose_streams'], s['prepare']))" | tee -a "$OUT/summary.txt"
    done; done
fi
if has splitab; then    # developer-build A/B of the synchronous calls' split LK chain (hop 0 under the t1 pair's PCIe pull + pyramids): VO_SYNC_SPLIT 0 / 1
    for rep in 1 2 3; do for SP in 0 1; do
        VO_HIP_LIB=$ROOT/visual_odom_amd/libvo_hip_dev.so VO_SYNC_SPLIT=$SP timeout 300 python tools/latency_mode.py 200 > "$OUT/latency_split${SP}_$rep.log" 2>&1
        echo "== VO_SYNC_SPLIT=$SP rep $rep" | tee -a "$OUT/summary.txt"
        grep -E "track_frame|adapter calls|stateless|kept pair" "$OUT/latency_split${SP}_$rep.log" | tee -a "$OUT/summary.txt"
    done; done
fi
if has probereps; then   # the schedule the loop settles on, run after run on ONE box: PROBE_REPS runs of PROBE_LIST entries "S ingest workload"
    for rep in $(seq 1 ${PROBE_REPS:-4}); do
        while read -r S ING WL; do
            [ -z "$S" ] && continue
            f="$OUT/pr_${S}_${ING}_${WL}_$rep.json"
            timeout 300 python bench.py --mode sequences --workload $WL --seqs $S --steps 60 --warmup 6 --no-cpu-baseline --validate 0 --ingest $ING > "$f" 2> "$f.err"
            python -c "import json; b=json.loads(open('$f').read().strip().splitlines()[-1]); s=b['config']['schedule']; print('S=$S $ING $WL rep $rep %8.0f fps %.3f ms/step  ran %s,%s,%s' % (b['value'], b['ms_per_step'], s['pose_waves'], s['pose_streams'], s['prepare']), {k: round(v, 2) for k, v in s.get('probe_ms', {}).items() if 'real' in k})" | tee -a "$OUT/summary.txt"
        done <<< "${PROBE_LIST:-256 pinned kitti374}"
    done
fi
if has runahead; then   # developer build: the host's run-ahead in the lock-step loop (VO_SEQ_RUNAHEAD steps instead of 8), schedules pinned
    while read -r S ING WL SC; do
        [ -z "$S" ] && continue
        for RA in ${RUNAHEAD_LIST:-8 4 3 2}; do for rep in 1 2; do
            f="$OUT/ra_${S}_${ING}_${WL}_${RA}_$rep.json"
            VO_HIP_LIB=$ROOT/visual_odom_amd/libvo_hip_dev.so VO_SEQ_RUNAHEAD=$RA timeout 300 python bench.py --mode sequences --workload $WL --seqs $S --steps 80 --warmup 8 --no-cpu-baseline --validate 0 --ingest $ING --schedule $SC > "$f" 2> "$f.err"
            python -c "import json; b=json.loads(open('$f').read().strip().splitlines()[-1]); print('S=$S $ING $WL sched $SC run-ahead $RA rep $rep %8.0f fps %.3f ms/step' % (b['value'], b['ms_per_step']))" | tee -a "$OUT/summary.txt"
        done; done
    done <<< "${RUNAHEAD_LOADS:-64 device kitti374 1,2,1}"
fi
stamp "done"
