#!/bin/bash
# Developer-build A/B of the lock-step loop's PCIe ingest in the loop itself, schedule pinned (profiles/r06_experiments.md
# section 1): VO_INGEST_WAVES = the persistent grid, VO_INGEST_WAIT = 0 / 1 the ingest starts at once / behind the running
# step's detection.   gpurun -- 'bash tools/ingest_waves.sh <tag> wait|waves'
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/${1:-ingw}; mkdir -p $OUT
export VO_HIP_LIB=$GRAFT_REPO_ROOT/visual_odom_amd/libvo_hip_dev.so
run() { # name, env..., -- bench args
    local name=$1; shift
    env "$@" python bench.py --mode sequences --steps 30 --warmup 4 --no-cpu-baseline --validate 0 $ARGS > $OUT/b_$name.json 2>/dev/null
    python -c "import json; b=json.loads(open('$OUT/b_$name.json').read().strip().splitlines()[-1]); print('$name  %.0f fps %.3f ms' % (b['value'], b['ms_per_step']), {k: round(v,2) for k,v in b['config']['stage_ms'].items()})" | tee -a $OUT/summary.txt
}
if [ "$2" = "wait" ]; then
    for SCH in 2,1,0 2,2,0 1,1,0; do for WL in kitti2000 kitti374; do for S in 256 64 8; do for W in 0 1; do
        ARGS="--workload $WL --seqs $S --ingest pinned --schedule $SCH"
        run "wait${W}_${WL}_S${S}_${SCH}" VO_INGEST_WAIT=$W
    done; done; done; done
else
    for G in 64 128 192 256 384 512; do for WL in kitti2000 kitti374; do
        ARGS="--workload $WL --seqs 256 --ingest pinned --schedule 2,1,0"
        run "G${G}_${WL}" VO_INGEST_WAVES=$G
    done; done
fi
