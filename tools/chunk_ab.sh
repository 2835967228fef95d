# A/B of the first RANSAC chunk in big launches (developer build)
cd ${GRAFT_REPO_ROOT:-.}
export VO_HIP_LIB=$PWD/visual_odom_amd/libvo_hip_dev.so
LEAN="--no-cpu-baseline --validate 0 --sustain 0 --no-replay-leg --no-configs"
for CH in 128 64 32; do
  for WL in kitti2000 kitti374; do
    VO_RANSAC_CHUNK=$CH python bench.py --workload $WL --steps 20 --warmup 3 $LEAN 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chunk=$CH batch $WL %.0f fps %.3f ms' % (b['value'], b['ms_per_step']), {k: round(v,2) for k,v in b['config']['stage_ms'].items()})"
  done
  for S in 64 256; do
    VO_RANSAC_CHUNK=$CH python bench.py --mode sequences --workload kitti374 --seqs $S --steps 40 --warmup 4 --no-cpu-baseline --validate 0 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chunk=$CH seq S=$S %.0f fps %.3f ms' % (b['value'], b['ms_per_step']))"
  done
done
