#!/bin/bash
# SQ counters of one kernel (mean per dispatch), each counter group in its own rocprofv3 run.
#   gpurun -- 'bash tools/pmc_kernel.sh <tag> <kernel substring> -- <command ...>'
# (TCP_* / TCC_* derived counters hang rocprofv3 on this pool: SQ_* and GRBM_* only.)
TAG=$1; KERNEL=$2; shift 3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
n=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
           "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC"; do
    n=$((n + 1))
    (cd /tmp && timeout 300 rocprofv3 --pmc $SET --output-format csv -d "$OUT/pmc$n" -- "$@" > "$OUT/pmc$n.log" 2>&1)
    python - "$OUT/pmc$n" "$KERNEL" <<'PYEOF' | tee -a "$OUT/pmc_summary.txt"
import csv, glob, sys, collections
fs = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not fs:
    print("  (no counter file)")
    sys.exit(0)
by = collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    if sys.argv[2] in r["Kernel_Name"]:
        by[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("  " + sys.argv[2] + ": " + "  ".join("%s %.4g" % (k, sum(v) / len(v)) for k, v in sorted(by.items())) + "  (dispatches %d)" % (len(next(iter(by.values()))) if by else 0))
PYEOF
    rm -rf "$OUT/pmc$n"
done
