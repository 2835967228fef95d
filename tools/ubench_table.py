"""rocprofv3 --pmc pass of tools/ubench/valu_rate -> issue cost per opcode in SIMD cycles, from the counters alone:
cycles per instruction per SIMD = GRBM_GUI_ACTIVE (shader-engine clock cycles the dispatch was busy) x 1024 SIMDs /
SQ_INSTS_VALU (wave-instructions, summed over the chip)."""
import csv
import glob
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("%-40s %14s %14s %10s %10s" % ("kernel", "SQ_INSTS_VALU", "GUI_ACTIVE", "cyc/inst", "busy/inst"))
for k, m in acc.items():
    v = {c: sum(x) / len(x) for c, x in m.items()}
    if "SQ_INSTS_VALU" not in v or v["SQ_INSTS_VALU"] == 0:
        continue
    name = k.split("(")[0]
    gui = v.get("GRBM_GUI_ACTIVE", 0.0)
    # GRBM_GUI_ACTIVE is reported summed over the 8 XCDs' counters on this part (profiles/r01: / 8 per die)
    cyc = gui / 8 * 1024 / v["SQ_INSTS_VALU"]
    busy = v.get("SQ_ACTIVE_INST_VALU", 0.0) * 4 / v["SQ_INSTS_VALU"]
    print("%-40s %14.0f %14.0f %10.2f %10.2f" % (name, v["SQ_INSTS_VALU"], gui, cyc, busy))
