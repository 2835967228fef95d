"""profiles/r04.md and profiles/r04_* from one tools/gpu_r3.sh output directory (gpurun_out/<tag>): the two rocprofv3
--kernel-trace --stats runs (batch mode as benched, lock-step loop with 256 sequences), the PMC passes, every bench line,
the latency-mode log.  Also rewrites profiles/lk_issue.json / lk_traffic.json (what bench.py imports as
roofline.valu_issue_imported / roofline.traffic) from the PMC passes.

    python tools/profile_summary_r4.py gpurun_out/r4_final
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def trace_table(d):
    stats = glob.glob(os.path.join(d, "*", "*_kernel_stats.csv")) + glob.glob(os.path.join(d, "*_kernel_stats.csv"))
    trace = glob.glob(os.path.join(d, "*", "*_kernel_trace.csv")) + glob.glob(os.path.join(d, "*_kernel_trace.csv"))
    if not stats:
        return None, []
    res = {}
    if trace:
        for r in csv.DictReader(open(trace[0])):
            res.setdefault(short(r["Kernel_Name"]), (r["VGPR_Count"], r.get("Accum_VGPR_Count", "0"), r["SGPR_Count"], r["LDS_Block_Size"],
                                                    r["Scratch_Size"], r["Workgroup_Size_X"], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"]))
    lines = ["| kernel | calls | avg us | min us | max us | % | VGPR | AGPR | SGPR | LDS B | scratch B | wg | grid |",
             "|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for r in csv.DictReader(open(stats[0])):
        n = short(r["Name"])
        x = res.get(n, ("",) * 9)
        lines.append("| %s | %s | %.1f | %.1f | %.1f | %.2f | %s | %s | %s | %s | %s | %s | %s x %s x %s |" % (
            n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3,
            float(r["Percentage"]), *x))
    return stats[0], lines


def main(src):
    tag = os.path.basename(os.path.normpath(src))
    out = ["# r04 -- rocprofv3 summaries and bench lines of round 4 (one MI355X, gpurun_out/%s = the final code of the round)" % tag, ""]
    for sub, title, cmd, dst in (
            ("prof_overlap", "kernel trace: batch mode as benched (streams overlapped)",
             "python bench.py --steps 5 --warmup 1 --no-cpu-baseline --validate 0 --sustain 0 --no-replay-leg --no-configs", "r04_kernel_stats_batch.csv"),
            ("prof_seq", "kernel trace: lock-step sequence loop, 256 sequences, reference-default bucketing",
             "python bench.py --mode sequences --workload kitti374 --seqs 256 --steps 10 --warmup 2 --no-cpu-baseline --validate 0", "r04_kernel_stats_seq.csv")):
        stats, lines = trace_table(os.path.join(src, sub))
        if stats:
            shutil.copy(stats, os.path.join(ROOT, "profiles", dst))
            out += ["## " + title, "", "`rocprofv3 --kernel-trace --stats -- %s`" % cmd, ""] + lines + [
                "", "(`epnp_kernel` asks for its 78 KB of LDS dynamically; rocprofv3 lists static LDS only.)", ""]
    pmc = defaultdict(lambda: defaultdict(list))
    for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
        if os.path.isdir(d):
            for f in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
                for r in csv.DictReader(open(f)):
                    pmc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if pmc:
        counters = sorted({c for k in pmc.values() for c in k})
        out += ["## PMC passes (each `--pmc` group in its own run of `python bench.py --steps 3 --warmup 1 --no-cpu-baseline --validate 0 --sustain 0 --no-replay-leg --no-configs`; mean per dispatch)", "",
                "| kernel | " + " | ".join(counters) + " |", "|---|" + "---|" * len(counters)]
        for k in sorted(pmc, key=lambda k: -sum(pmc[k].get("SQ_INSTS_VALU", [0]))):
            if not k.startswith("__amd"):
                out.append("| %s | " % k + " | ".join("%.4g" % (sum(pmc[k][c]) / len(pmc[k][c])) if pmc[k].get(c) else "" for c in counters) + " |")
        out += ["", "FETCH_SIZE / WRITE_SIZE are in KB (FETCH_SIZE undercounts by 2x on gfx950, profiles/r01_fetch_calibration.txt).  "
                "GRBM_GUI_ACTIVE is summed over the 8 XCDs.", ""]
        lk = pmc.get("vo::lk_circular_kernel")
        bench = os.path.join(src, "bench.json")
        b = json.loads(open(bench).read().strip().splitlines()[-1]) if os.path.exists(bench) and os.path.getsize(bench) else None
        mean = lambda v: sum(v) / len(v)
        if lk and lk.get("SQ_INSTS_VALU") and lk.get("GRBM_GUI_ACTIVE"):
            old = json.load(open(os.path.join(ROOT, "profiles", "lk_issue.json")))
            valu, waves = mean(lk["SQ_INSTS_VALU"]), mean(lk["SQ_WAVES"])
            cycles = mean(lk["GRBM_GUI_ACTIVE"]) / 8.0
            old.update(valu_instructions_per_launch=valu, waves_per_launch=waves, valu_instructions_per_feature=valu / waves,
                       salu_instructions_per_feature=mean(lk["SQ_INSTS_SALU"]) / waves, lds_instructions_per_feature=mean(lk["SQ_INSTS_LDS"]) / waves,
                       shader_cycles_per_launch=cycles, simd_cycles_per_valu_instruction=cycles * 1024.0 / valu)
            old["measured_over_bound"] = old["simd_cycles_per_valu_instruction"] / old["issue_cost_bound_cycles_per_valu_instruction"]
            old["source"] = ("r04 (gpurun_out/%s): rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVES "
                             "SQ_BUSY_CYCLES (its own run of `python bench.py --steps 3 --warmup 1 --no-cpu-baseline --validate 0 --sustain 0 --no-replay-leg "
                             "--no-configs`), means over the lk_circular_kernel dispatches; the bound = sum over the hot loop's opcodes of count x measured "
                             "issue cost (tools/isa_histogram.py with profiles/r02_valu_issue_cost.txt: 80 VALU / 324 cycles per iteration), "
                             "profiles/r02_lk_issue_bound.md" % tag)
            json.dump(old, open(os.path.join(ROOT, "profiles", "lk_issue.json"), "w"), indent=1)
        if lk and lk.get("FETCH_SIZE") and lk.get("WRITE_SIZE"):
            old = json.load(open(os.path.join(ROOT, "profiles", "lk_traffic.json")))
            old["fetch_bytes_per_launch"] = mean(lk["FETCH_SIZE"]) * 1024.0
            old["write_bytes_per_launch"] = mean(lk["WRITE_SIZE"]) * 1024.0
            old["hbm_bytes_per_launch"] = 2.0 * old["fetch_bytes_per_launch"] + old["write_bytes_per_launch"]
            old["hbm_bytes_per_launch_uncorrected"] = old["fetch_bytes_per_launch"] + old["write_bytes_per_launch"]
            old["source"] = ("r04 (gpurun_out/%s): rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate runs of `python bench.py --steps 3 --warmup 1 "
                             "--no-cpu-baseline --validate 0 --sustain 0 --no-replay-leg --no-configs`; KB -> bytes; FETCH_SIZE doubled (gfx950 correction, "
                             "calibrated in profiles/r01_fetch_calibration.txt); WRITE_SIZE as reported" % tag)
            pp_ = pmc.get("vo::pyr_pass_kernel")
            if pp_ and pp_.get("FETCH_SIZE") and pp_.get("WRITE_SIZE"):
                # four dispatches per step (levels 0 .. 3): per-step totals = 4 x the mean per dispatch
                pf, pw = 4 * mean(pp_["FETCH_SIZE"]) * 1024.0, 4 * mean(pp_["WRITE_SIZE"]) * 1024.0
                old["pyramid_stage"] = {"images_per_step": 514, "fetch_bytes_per_step": 2.0 * pf, "write_bytes_per_step": pw,
                                        "hbm_bytes_per_step": 2.0 * pf + pw, "kernel": "pyr_pass_kernel (4 launches per step)"}
            json.dump(old, open(os.path.join(ROOT, "profiles", "lk_traffic.json"), "w"), indent=1)
    # the fused pyramid pass level by level (consecutive launches of a step = levels 0 .. L-1) and its HBM traffic
    tr = glob.glob(os.path.join(src, "prof_overlap", "*", "*_kernel_trace.csv")) + glob.glob(os.path.join(src, "prof_overlap", "*_kernel_trace.csv"))
    if tr:
        rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Grid_Size_X"])) for r in csv.DictReader(open(tr[0]))
                      if short(r["Kernel_Name"]) == "vo::pyr_pass_kernel")
        by_grid = defaultdict(list)
        for st, en, g in rows:
            by_grid[g].append((en - st) / 1e3)
        if by_grid:
            out += ["## `pyr_pass_kernel` level by level (same trace; the grid width identifies the level)", "",
                    "| grid x | launches | avg us | min us | max us |", "|---|---|---|---|---|"]
            for g in sorted(by_grid, reverse=True):
                v = by_grid[g]
                out.append("| %d | %d | %.1f | %.1f | %.1f |" % (g, len(v), sum(v) / len(v), min(v), max(v)))
            out.append("")
    pp = pmc.get("vo::pyr_pass_kernel") if pmc else None
    if pp and pp.get("FETCH_SIZE") and pp.get("WRITE_SIZE"):
        mean = lambda v: sum(v) / len(v)
        # four dispatches per step (levels 0 .. 3): per-step totals = 4 x the mean per dispatch
        f, w = 4 * mean(pp["FETCH_SIZE"]) * 1024.0, 4 * mean(pp["WRITE_SIZE"]) * 1024.0
        out += ["Pyramid stage traffic per step of 514 images (4 x the mean per dispatch): FETCH_SIZE %.3f GB raw (x 2 on gfx950 = %.3f GB), "
                "WRITE_SIZE %.3f GB; algorithmic (SURVEY 8d) 0.394 GB, designed (+ the 4-byte Scharr pixel) 1.668 GB: reads 0.315 GB, "
                "writes 1.353 GB." % (f / 1e9, 2 * f / 1e9, w / 1e9), ""]
    for p in sorted(glob.glob(os.path.join(src, "bench*.json"))):
        if os.path.getsize(p):
            name = os.path.basename(p)
            shutil.copy(p, os.path.join(ROOT, "profiles", "r04_" + name))
            out += ["## " + name, "", "```json", open(p).read().strip().splitlines()[-1], "```", ""]
    for extra, dst in (("ingest.json", "r04_ingest.json"), ("sweep.jsonl", "r04_schedule_sweep.jsonl"), ("timeline.txt", "r04_track_frame_timeline.txt"),
                       ("timeline_seq.txt", "r04_one_sequence_timeline.txt")):
        pth = os.path.join(src, extra)
        if os.path.exists(pth) and os.path.getsize(pth):
            shutil.copy(pth, os.path.join(ROOT, "profiles", dst))
    lat = os.path.join(src, "latency.log")
    if os.path.exists(lat):
        txt = "\n".join(l for l in open(lat).read().splitlines() if "amdgpu.ids" not in l)
        open(os.path.join(ROOT, "profiles", "r04_latency_mode.txt"), "w").write(txt + "\n")
        out += ["## latency mode of the drop-in calls (tools/latency_mode.py: host images, PCIe-inclusive, one process per measurement)", "", "```", txt, "```", ""]
    open(os.path.join(ROOT, "profiles", "r04.md"), "w").write("\n".join(out))
    print("profiles/r04.md: %d lines" % len(out))


if __name__ == "__main__":
    main(sys.argv[1])
