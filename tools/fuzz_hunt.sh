#!/bin/bash
# A longer, differently seeded run of the vo_track_frame fuzz (tests/test_gpu_round6.py): 6 x 2 000 random cases.
#   gpurun -- 'bash tools/fuzz_hunt.sh'   -> gpurun_out/r6_fuzz/seed*.log
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6_fuzz
for SEED in 1 2 3 4 5 6; do
VO_FUZZ_EXAMPLES=2000 VO_FUZZ_SEED=$SEED timeout 900 python -m pytest tests/test_gpu_round6.py -m gpu -q -x -k "track_frame_fuzz" -s > gpurun_out/r6_fuzz/seed$SEED.log 2>&1
tail -3 gpurun_out/r6_fuzz/seed$SEED.log; grep "fuzz:" gpurun_out/r6_fuzz/seed$SEED.log
done
