#!/bin/bash
# One GPU-box session: parity tests, bench (both LDS read variants), rocprofv3 kernel trace and the
# two PMC passes.  Usage (from the authoring container):
#   gpurun --timeout 1500 -- 'bash tools/gpu_round.sh <tag>'
# Everything lands in gpurun_out/<tag>/.
TAG=${1:-r}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT" || exit 1
export TMPDIR=/tmp
t0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a "$OUT/timeline.log"; }

stamp "pytest -m gpu"
timeout 1200 python -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1
rc=$?
stamp "pytest rc=$rc"
tail -5 "$OUT/pytest.log"
if [ $rc -ne 0 ]; then
    stamp "pytest failed: detailed first-contact diffs"
    VO_PNP=1 timeout 400 python tools/dev_gpu_check.py > "$OUT/dev.log" 2>&1
    tail -40 "$OUT/dev.log"
fi
stamp "VALU issue-rate micro-benchmark"
(cd tools/ubench && timeout 120 /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w valu_rate.hip -o /tmp/valu_rate && timeout 60 /tmp/valu_rate) > "$OUT/valu_rate.log" 2>&1
cat "$OUT/valu_rate.log"

stamp "latency mode of the drop-in boundary (host images, PCIe-inclusive)"
timeout 300 python tools/latency_mode.py 40 > "$OUT/latency.log" 2>&1
cat "$OUT/latency.log"
stamp "bench (default)"
timeout 600 python bench.py --steps 20 --warmup 3 > "$OUT/bench.json" 2> "$OUT/bench.err"
cat "$OUT/bench.json"
stamp "bench (reference-default load, 374 points per frame)"
timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --workload kitti374 > "$OUT/bench_kitti374.json" 2> "$OUT/bench_kitti374.err"
cat "$OUT/bench_kitti374.json"

stamp "bench (FAST + bucketing on the device feed LK)"
timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --stages detect+full > "$OUT/bench_detect.json" 2> "$OUT/bench_detect.err"
cat "$OUT/bench_detect.json"
stamp "bench (mono_rotation: essential matrix + recoverPose next to the PnP solve)"
timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --mono-rotation > "$OUT/bench_mono.json" 2> "$OUT/bench_mono.err"
cat "$OUT/bench_mono.json"
stamp "bench (1920x1080, 4000 points per frame)"
timeout 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload hd4000 --frames 16 > "$OUT/bench_hd4000.json" 2> "$OUT/bench_hd4000.err"
cat "$OUT/bench_hd4000.json"
stamp "bench (pose solve serialised on the tracking stream)"
VO_SERIAL_POSE=1 timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/bench_serial.json" 2> "$OUT/bench_serial.err"
cat "$OUT/bench_serial.json"
cd /tmp
stamp "rocprofv3 kernel trace (serialised pose solve: stand-alone kernel durations)"
VO_SERIAL_POSE=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- python "$ROOT/bench.py" --steps 5 --warmup 1 --no-cpu-baseline > "$OUT/prof.log" 2>&1
stamp "rocprofv3 kernel trace (overlapped, as benched)"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_overlap" -- python "$ROOT/bench.py" --steps 5 --warmup 1 --no-cpu-baseline > "$OUT/prof_overlap.log" 2>&1
stamp "rocprofv3 pmc FETCH_SIZE"
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/pmc_fetch.log" 2>&1
stamp "rocprofv3 pmc WRITE_SIZE"
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/pmc_write.log" 2>&1
stamp "rocprofv3 pmc SQ pass A"
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sqa" -- python "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/pmc_sqa.log" 2>&1
stamp "rocprofv3 pmc SQ pass B"
timeout 400 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM --output-format csv -d "$OUT/pmc_sqb" -- python "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/pmc_sqb.log" 2>&1
stamp "done"
find "$OUT" -type f | head -50
du -sh "$OUT"
