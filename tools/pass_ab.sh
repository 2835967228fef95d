#!/bin/bash
# the fused pyramid pass standalone (tools/ubench/pass_bench.hip): product and the knock-out builds (-DVO_PASS_X=bits: 1 no
# border items, 2 no next-level stores, 4 no edge items), KITTI and the other camera shapes.  gpurun -- 'bash tools/pass_ab.sh'
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT" || exit 1
for X in ${PASS_X:-0 1 2 4 7}; do
    hipcc --offload-arch=gfx950 -O3 -DVO_DEV_VARIANTS -DVO_PASS_X=$X -Iinclude -Ivisual_odom_amd/csrc tools/ubench/pass_bench.hip -o /tmp/pass_bench_$X 2>/dev/null || { echo "build failed ($X)"; exit 1; }
done
for rep in 1 2; do
    for X in ${PASS_X:-0 1 2 4 7}; do
        /tmp/pass_bench_$X 1241 376 514 | sed "s/^/[x$X] /"
    done
done
for SHAPE in "640 480 512" "1280 720 512" "1920 1080 256"; do
    /tmp/pass_bench_0 $SHAPE | sed "s/^/[x0] /"
done
