#!/bin/bash
# L2 / memory-side counters of the fused pyramid pass, level 0, XCD-pinned against dispatch order (why does KITTI's level 0 prefer
# dispatch order?  VERDICT r04 item 4).  Raw TCC counters only, every rocprofv3 call under its own short timeout: the DERIVED
# TCP_* / TCC_* groups hung rocprofv3 on this pool in round 4 (gpurun_out/r4_23).  gpurun -- 'bash tools/pass_pmc.sh'
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT" || exit 1
OUT=$ROOT/gpurun_out/pass_pmc; mkdir -p "$OUT"; export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -DVO_DEV_VARIANTS -DVO_PASS_X=0 -DPASS_BENCH_REPS=3 -Iinclude -Ivisual_odom_amd/csrc tools/ubench/pass_bench.hip -o /tmp/pass_bench_pmc || exit 1
(cd /tmp && timeout 60 rocprofv3 --list-avail 2>/dev/null | grep -o "TCC_[A-Z0-9_]*" | sort -u | tr '\n' ' ' | cut -c1-3000) > "$OUT/tcc_counters.txt"
echo "available TCC counters: $(wc -w < "$OUT/tcc_counters.txt")"
for SET in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_WRREQ_STALL_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_WRITE_sum TCC_READ_sum"; do
    TAG=$(echo $SET | tr ' ' '+')
    echo "== $SET"
    (cd /tmp && timeout 90 rocprofv3 --pmc $SET --output-format csv -d "$OUT/$TAG" -- /tmp/pass_bench_pmc 1241 376 514 > "$OUT/$TAG.log" 2>&1; echo "rc=$?")
    python - "$OUT/$TAG" <<'PY'
import csv, glob, sys, collections
fs = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not fs:
    print("  (no counter file)"); sys.exit(0)
by = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fs[0])):
    if "pyr_pass_kernel" in r["Kernel_Name"] and "sm_kernel" not in r["Kernel_Name"]:
        by[int(r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for g in sorted(by, reverse=True)[:2]:   # the two largest grids = level 0: pinned order (padded to 520 images) and dispatch order (514)
    print("  grid %9d: " % g + "  ".join("%s %.4g" % (k, sum(v) / len(v)) for k, v in sorted(by[g].items())))
PY
    rm -rf "$OUT/$TAG"
done
