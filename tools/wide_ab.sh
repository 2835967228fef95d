# A/B of the two EPnP forms (developer build): VO_EPNP_SPLIT_MAX = largest launch (frames) the four-kernel form takes
cd ${GRAFT_REPO_ROOT:-.}
export VO_HIP_LIB=$PWD/visual_odom_amd/libvo_hip_dev.so
for WM in ${*:-4 0}; do
  echo "=== VO_EPNP_SPLIT_MAX=$WM"
  VO_EPNP_SPLIT_MAX=$WM python -c "import tools.latency_mode as l; l.run('track',6,300); l.run('track',1,300)"
  for S in 1 4 8 16 32; do
    VO_EPNP_SPLIT_MAX=$WM python bench.py --mode sequences --workload kitti374 --seqs $S --steps 60 --warmup 4 --no-cpu-baseline --validate 0 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  S=$S %.0f fps %.3f ms/step' % (b['value'], b['ms_per_step']), {k: b['config']['schedule'][k] for k in ('pose_waves','pose_streams','prepare')})"
  done
done
