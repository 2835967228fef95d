#!/bin/bash
# Lock-step loop, 256 / 64 / 8 sequences: page-locked / pageable / resident pairs side by side (profiles/r06_ingest_ab.txt).
#   gpurun -- 'bash tools/ingest_ab.sh <tag>'   -> gpurun_out/<tag>/summary.txt
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/${1:-ingab}; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q -x -k "sequence or seq or lockstep or ingest or run_command or soak" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for WL in kitti2000 kitti374; do for ING in pinned host device; do
python bench.py --mode sequences --workload $WL --seqs 256 --steps 30 --warmup 4 --no-cpu-baseline --validate 2 --ingest $ING > $OUT/b_${WL}_${ING}.json 2>/dev/null
python -c "import json; b=json.loads(open('$OUT/b_${WL}_${ING}.json').read().strip().splitlines()[-1]); print('$WL $ING  %.0f fps %.2f ms val %d' % (b['value'], b['ms_per_step'], b['validated_frames']), {k: round(v,2) for k,v in b['config']['stage_ms'].items()}, b['config']['schedule']['prepare'])" | tee -a $OUT/summary.txt
done; done
for S in 8 64; do for ING in pinned host device; do
python bench.py --mode sequences --workload kitti374 --seqs $S --steps 60 --warmup 4 --no-cpu-baseline --validate 0 --ingest $ING > $OUT/b_S${S}_${ING}.json 2>/dev/null
python -c "import json; b=json.loads(open('$OUT/b_S${S}_${ING}.json').read().strip().splitlines()[-1]); print('kitti374 S=$S $ING  %.0f fps %.3f ms' % (b['value'], b['ms_per_step']))" | tee -a $OUT/summary.txt
done; done
