#!/bin/bash
# developer build: the one-sequence lock-step loop on the partitioned stream set (half of the CUs for the post-LK streams) against
# VO_POSE_CUS=0 (no partition) and other splits; 8 sequences never use it.  gpurun -- 'bash tools/cu_mask_ab.sh'
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT" || exit 1
export VO_HIP_LIB=$ROOT/visual_odom_amd/libvo_hip_dev.so
for ING in device host; do
  for N in 0 64 128 160; do
    for SCHED in probe 1,2,0 1,1,0; do
      S=""; [ $SCHED != probe ] && S="--schedule $SCHED"
      VO_POSE_CUS=$N timeout 300 python bench.py --mode sequences --workload kitti374 --seqs 1 --steps 300 --warmup 10 --no-cpu-baseline --validate 2 $S --quads 8 --ingest $ING > /tmp/cm.json 2>/tmp/cm.err
      python -c "import json; b=json.loads(open('/tmp/cm.json').read().strip().splitlines()[-1]); s=b['config']['schedule']; print('S=1 %-6s pose CUs %3d sched %-6s -> %s,%s,%s  %.0f fps %.3f ms/step val %d' % ('$ING', $N, '$SCHED', s['pose_waves'], s['pose_streams'], s['prepare'], b['value'], b['ms_per_step'], b['validated_frames']))" 2>&1 || tail -2 /tmp/cm.err
    done
  done
done
