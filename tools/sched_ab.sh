#!/bin/bash
# Pinned-schedule A/B of the lock-step loop on ONE box: bash tools/sched_ab.sh <tag> <seqs> "<sched> <sched> ..." [workload] [reps]
# (a schedule is pose_waves,pose_streams,prepare as bench.py --schedule takes it; "probe" = the library's own pick)
TAG=$1; S=$2; SCHEDS=$3; WL=${4:-kitti374}; REPS=${5:-3}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$ROOT" || exit 1
for rep in $(seq 1 $REPS); do
  for sc in $SCHEDS; do
    f="$OUT/ab_${S}_${sc//,/}_$rep.json"
    timeout 300 python bench.py --mode sequences --workload $WL --seqs $S --steps 60 --warmup 6 --no-cpu-baseline --validate 0 \
        $([ "$sc" = probe ] || echo --schedule $sc) > "$f" 2> "$f.err"
    python -c "import json; b=json.loads(open('$f').read().strip().splitlines()[-1]); s=b['config']['schedule']; print('S=$S $WL rep $rep %-6s %8.0f fps %.3f ms/step  ran %s,%s,%s' % ('$sc', b['value'], b['ms_per_step'], s['pose_waves'], s['pose_streams'], s['prepare']))" | tee -a "$OUT/summary.txt"
  done
done
