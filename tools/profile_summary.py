"""Turn one tools/gpu_round.sh output directory (gpurun_out/<tag>/) into the committed evidence under
profiles/: per-kernel statistics of the rocprofv3 --kernel-trace --stats run, the PMC passes
(FETCH_SIZE / WRITE_SIZE in their own runs, SQ counters in two more), the bench JSON lines, and
profiles/lk_traffic.json (HBM bytes per LK launch, read back by bench.py as roofline.traffic).

    python tools/profile_summary.py gpurun_out/r2 profiles/r01_v2
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def main(src, prefix):
    os.makedirs(os.path.dirname(prefix) or ".", exist_ok=True)
    lines = ["# %s -- rocprofv3 summaries of `python bench.py` on one MI355X" % os.path.basename(prefix), ""]
    stats = glob.glob(os.path.join(src, "prof", "*", "*_kernel_stats.csv"))
    trace = glob.glob(os.path.join(src, "prof", "*", "*_kernel_trace.csv"))
    res = {}
    if trace:
        for r in csv.DictReader(open(trace[0])):
            res.setdefault(short(r["Kernel_Name"]), (r["VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"],
                                                    r["Workgroup_Size_X"], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"]))
    if stats:
        shutil.copy(stats[0], prefix + "_kernel_stats.csv")
        lines += ["## kernel trace (`rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 1`)", "",
                  "| kernel | calls | avg us | min us | max us | % | VGPR | SGPR | LDS B | scratch B | wg | grid |", "|---|---|---|---|---|---|---|---|---|---|---|---|"]
        for r in csv.DictReader(open(stats[0])):
            n = short(r["Name"])
            x = res.get(n, ("",) * 8)
            lines.append("| %s | %s | %.1f | %.1f | %.1f | %.2f | %s | %s | %s | %s | %s | %s x %s x %s |" % (
                n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3,
                float(r["Percentage"]), x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7]))
        lines.append("")
    pmc = defaultdict(lambda: defaultdict(list))
    for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
        if not os.path.isdir(d):
            continue
        for f in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                pmc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if pmc:
        counters = sorted({c for k in pmc.values() for c in k})
        lines += ["## PMC passes (each `--pmc` group in its own run; mean per dispatch)", "",
                  "| kernel | " + " | ".join(counters) + " |", "|---|" + "---|" * len(counters)]
        for k in sorted(pmc, key=lambda k: -sum(pmc[k].get("SQ_WAVE_CYCLES", [0]))):
            if k.startswith("__amd"):
                continue
            lines.append("| %s | " % k + " | ".join("%.4g" % (sum(pmc[k][c]) / len(pmc[k][c])) if pmc[k].get(c) else "" for c in counters) + " |")
        lines += ["", "FETCH_SIZE / WRITE_SIZE are in KB.  SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_* count quad-cycles.",
                  "GRBM_GUI_ACTIVE is summed over the 8 XCDs.", ""]
    for name in ("bench.json", "bench_kitti374.json", "bench_detect.json", "bench_mono.json", "bench_hd4000.json", "bench_serial.json"):
        p = os.path.join(src, name)
        if os.path.exists(p) and os.path.getsize(p):
            shutil.copy(p, prefix + "_" + name)
            lines += ["## %s" % name, "", "```json", open(p).read().strip(), "```", ""]
    for name, title in (("valu_rate.log", "VALU issue-rate micro-benchmark (tools/ubench/valu_rate.hip)"),
                        ("latency.log", "latency mode of the drop-in calls (tools/latency_mode.py: host images, PCIe-inclusive)")):
        p = os.path.join(src, name)
        if os.path.exists(p):
            lines += ["## " + title, "", "```", open(p).read().strip(), "```", ""]
    open(prefix + ".md", "w").write("\n".join(lines))
    lk = pmc.get("vo::lk_circular_kernel")
    bench = os.path.join(src, "bench.json")
    if lk and lk.get("FETCH_SIZE") and lk.get("WRITE_SIZE") and os.path.exists(bench):
        b = json.loads(open(bench).read().strip().splitlines()[-1])
        fetch = sum(lk["FETCH_SIZE"]) / len(lk["FETCH_SIZE"]) * 1024.0
        write = sum(lk["WRITE_SIZE"]) / len(lk["WRITE_SIZE"]) * 1024.0
        rec = {"workload": "kitti2000", "frames_per_step": b["config"]["frames_per_step_per_gpu"],
               "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
               # MI355X_MICROARCH.md: gfx950 FETCH_SIZE tallies 128-byte requests as 64 -> doubled.  Calibrated on
               # this box with tools/ubench/fetch_calib.hip (1 GiB read once): 16 B/lane stream 0.500, 8 B/lane
               # stream 0.500, LK-like 21-row x 16 B gather 0.566 of the true bytes (profiles/r01_fetch_calibration.txt)
               "hbm_bytes_per_launch": 2 * fetch + write,
               "hbm_bytes_per_launch_uncorrected": fetch + write,
               "source": os.path.basename(prefix) + ": rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate runs of "
                         "`python bench.py --steps 3 --warmup 1 --no-cpu-baseline`; KB -> bytes; FETCH_SIZE doubled (gfx950 "
                         "correction, calibrated in profiles/r01_fetch_calibration.txt); WRITE_SIZE as reported"}
        json.dump(rec, open(os.path.join(os.path.dirname(prefix), "lk_traffic.json"), "w"), indent=1)
    if lk and lk.get("SQ_INSTS_VALU") and lk.get("GRBM_GUI_ACTIVE") and lk.get("SQ_WAVES") and os.path.exists(bench):
        b = json.loads(open(bench).read().strip().splitlines()[-1])
        mean = lambda k: sum(lk[k]) / len(lk[k])
        valu, waves, cyc = mean("SQ_INSTS_VALU"), mean("SQ_WAVES"), mean("GRBM_GUI_ACTIVE") / 8.0
        rec = {"workload": "kitti2000", "frames_per_step": b["config"]["frames_per_step_per_gpu"],
               "valu_instructions_per_launch": valu, "waves_per_launch": waves,
               "valu_instructions_per_feature": valu / waves,
               "salu_instructions_per_feature": mean("SQ_INSTS_SALU") / waves if lk.get("SQ_INSTS_SALU") else None,
               "shader_cycles_per_launch": cyc,
               # 1024 SIMDs; one VALU instruction of a wave64 occupies a SIMD's issue slot for 4 cycles
               "simd_cycles_per_valu_instruction": cyc * 1024.0 / valu,
               "valu_issue_utilisation": 4.0 * valu / (cyc * 1024.0),
               "source": os.path.basename(prefix) + ": rocprofv3 --pmc SQ_INSTS_VALU ... GRBM_GUI_ACTIVE (its own run of "
                         "`python bench.py --steps 3 --warmup 1 --no-cpu-baseline`), means over the lk_circular_kernel dispatches"}
        json.dump(rec, open(os.path.join(os.path.dirname(prefix), "lk_issue.json"), "w"), indent=1)
    print(open(prefix + ".md").read()[:3000])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
