#!/bin/bash
# Lock-step loop, 256 / 64 / 8 sequences: page-locked / pageable / resident pairs side by side (profiles/r06_ingest_ab.txt).
#   gpurun -- 'bash tools/ingest_ab.sh'   -> gpurun_out/<dir>/summary.txt
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6_ingab4
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r6_ingab4/pytest.log 2>&1; tail -4 gpurun_out/r6_ingab4/pytest.log
for WL in kitti2000 kitti374; do for ING in pinned host; do
python bench.py --mode sequences --workload $WL --seqs 256 --steps 30 --warmup 4 --no-cpu-baseline --validate 2 --ingest $ING > gpurun_out/r6_ingab4/b_${WL}_${ING}.json 2>/dev/null
python -c "import json; b=json.loads(open('gpurun_out/r6_ingab4/b_${WL}_${ING}.json').read().strip().splitlines()[-1]); print('$WL $ING  %.0f fps %.2f ms val %d' % (b['value'], b['ms_per_step'], b['validated_frames']), {k: round(v,2) for k,v in b['config']['stage_ms'].items()}, b['config']['schedule']['prepare'])" | tee -a gpurun_out/r6_ingab4/summary.txt
done; done
for S in 8 64; do for ING in host; do
python bench.py --mode sequences --workload kitti374 --seqs $S --steps 60 --warmup 4 --no-cpu-baseline --validate 0 --ingest $ING > gpurun_out/r6_ingab4/b_S${S}_${ING}.json 2>/dev/null
python -c "import json; b=json.loads(open('gpurun_out/r6_ingab4/b_S${S}_${ING}.json').read().strip().splitlines()[-1]); print('kitti374 S=$S $ING  %.0f fps %.3f ms' % (b['value'], b['ms_per_step']))" | tee -a gpurun_out/r6_ingab4/summary.txt
done; done
