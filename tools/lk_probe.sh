#!/bin/bash
# Quick LK A/B on the GPU box: parity tests, two bench lines and the VALU / SALU instruction counts of the LK
# launch (one rocprofv3 --pmc pass).  Usage: gpurun -- 'bash tools/lk_probe.sh <tag>'
TAG=${1:-probe}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT" || exit 1
export TMPDIR=/tmp
timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -1
for i in 1 2; do timeout 300 python bench.py --steps 12 --warmup 2 --no-cpu-baseline > "$OUT/bench_$i.json" 2>/dev/null; done
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d "$OUT/pmc" -- python "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1)
python - <<PY
import glob, json, csv
from collections import defaultdict
for f in sorted(glob.glob("$OUT/bench_*.json")):
    b = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], "fps %.0f ms/step %.3f" % (b["value"], b["ms_per_step"]), {k: round(v, 3) for k, v in b["config"]["stage_ms"].items()})
acc = defaultdict(list)
for f in glob.glob("$OUT/pmc/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "lk_circular" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
if m:
    print("LK per feature: VALU %.0f SALU %.0f LDS %.1f; cycles/VALU %.2f" % (
        m["SQ_INSTS_VALU"] / m["SQ_WAVES"], m["SQ_INSTS_SALU"] / m["SQ_WAVES"], m["SQ_INSTS_LDS"] / m["SQ_WAVES"],
        m["GRBM_GUI_ACTIVE"] / 8 * 1024 / m["SQ_INSTS_VALU"]))
PY
