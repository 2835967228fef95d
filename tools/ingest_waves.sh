#!/bin/bash
# Sweep of the ingest kernel's persistent grid (developer build: VO_INGEST_WAVES / VO_INGEST_WAVES_DEV) in the loop itself,
# schedule pinned (profiles/r06_experiments.md section 1).   gpurun -- 'bash tools/ingest_waves.sh'
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6_ingw
export VO_HIP_LIB=$GRAFT_REPO_ROOT/visual_odom_amd/libvo_hip_dev.so
for G in 64 128 192 256 384 512; do for WL in kitti2000 kitti374; do
VO_INGEST_WAVES=$G python bench.py --mode sequences --workload $WL --seqs 256 --steps 30 --warmup 4 --no-cpu-baseline --validate 0 --ingest pinned --schedule 2,1,0 > gpurun_out/r6_ingw/b_${WL}_${G}.json 2>/dev/null
python -c "import json; b=json.loads(open('gpurun_out/r6_ingw/b_${WL}_${G}.json').read().strip().splitlines()[-1]); print('pinned G=$G $WL  %.0f fps %.2f ms' % (b['value'], b['ms_per_step']), {k: round(v,2) for k,v in b['config']['stage_ms'].items()})" | tee -a gpurun_out/r6_ingw/summary.txt
done; done
for G in 1024 2048 8192 32768; do
VO_INGEST_WAVES_DEV=$G python bench.py --mode sequences --workload kitti374 --seqs 256 --steps 30 --warmup 4 --no-cpu-baseline --validate 0 --ingest device --schedule 2,1,0 > gpurun_out/r6_ingw/b_dev_${G}.json 2>/dev/null
python -c "import json; b=json.loads(open('gpurun_out/r6_ingw/b_dev_${G}.json').read().strip().splitlines()[-1]); print('device G=$G kitti374  %.0f fps %.2f ms' % (b['value'], b['ms_per_step']), {k: round(v,2) for k,v in b['config']['stage_ms'].items()})" | tee -a gpurun_out/r6_ingw/summary.txt
done
