#!/bin/bash
# round-5 library against the round-4 library on ONE box (visual_odom_amd/libvo_hip_r04.so: `git worktree add /tmp/r04 7069a8a && (cd /tmp/r04 && python -m visual_odom_amd.build) && cp /tmp/r04/visual_odom_amd/libvo_hip.so visual_odom_amd/libvo_hip_r04.so`; the C ABI is unchanged), pinned
# schedules: the lock-step loop at 8 / 16 / 64 sequences and the batch legs.  gpurun -- 'bash tools/ab_r04.sh'
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT" || exit 1
for S in 8 16 64; do
  for SCHED in 1,2,1 1,1,0 2,2,0; do
    for LIB in r04 r05; do
      L=$ROOT/visual_odom_amd/libvo_hip.so; [ $LIB = r04 ] && L=$ROOT/visual_odom_amd/libvo_hip_r04.so
      VO_HIP_LIB=$L timeout 300 python bench.py --mode sequences --workload kitti374 --seqs $S --steps 60 --warmup 6 --no-cpu-baseline --validate 0 --schedule $SCHED --quads 8 > /tmp/ab.json 2>/tmp/ab.err
      python -c "import json; b=json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1]); print('S=%-3d sched %s %s  %.0f fps %.3f ms/step' % ($S, '$SCHED', '$LIB', b['value'], b['ms_per_step']), {k: round(v,2) for k,v in b['config']['stage_ms'].items()})" 2>&1 || tail -2 /tmp/ab.err
    done
  done
done
