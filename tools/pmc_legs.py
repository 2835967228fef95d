"""PMC passes of the LK launch for every bench leg -> profiles/lk_traffic.json / lk_issue.json.

    python tools/pmc_legs.py gpurun_out/<tag>     # written by `bash tools/gpu_round.sh <tag> pmclegs`

Input: gpurun_out/<tag>/pmc_<workload>_<set>/**/*_counter_collection.csv with set = fetch (FETCH_SIZE), write (WRITE_SIZE),
sq (SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES) -- each counter
group its own rocprofv3 run of `python bench.py --workload W [--frames F] --steps 3 --warmup 1 <lean>` -- and
gpurun_out/<tag>/pmc_<workload>.json (that run's bench line: frames per step).  The headline workload's record stays at the top
level of the two JSON files (what round 1-4 wrote), every workload's record goes to "legs"; bench.py looks a leg up by
(workload, frames per step).  Unit and gfx950 corrections as MI355X_MICROARCH.md prescribes: FETCH_SIZE / WRITE_SIZE in KB,
FETCH_SIZE doubled (calibrated: profiles/r01_fetch_calibration.txt)."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ISSUE_COST = 4.05  # SIMD-cycles per wave64 VALU instruction of the hot loop's opcode mix at the micro-benchmark's own costs (profiles/r02_lk_issue_bound.md): a MODEL
ISSUE_FLOOR = 3.65  # the same mix with every opcode at the floor of its class (2.46 / 4.0 / 8.0; tools/isa_histogram.py --floor: iteration
                    # 3.68, cell entry 3.65, level set-up 3.77 -- the smallest of the three): no issue schedule beats it, a BOUND


def counters(d, kernel="lk_circular_kernel"):
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def pyr_counters(d):
    """sum over the pyr_pass_kernel launches of a step (one per level), averaged over the steps"""
    per = {}
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "pyr_pass_kernel" in r["Kernel_Name"]:
                per.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {k: sum(v) for k, v in per.items()}, {k: len(v) for k, v in per.items()}


def main(src):
    tag = os.path.basename(os.path.normpath(src))
    traffic_legs, issue_legs = [], []
    for bj in sorted(glob.glob(os.path.join(src, "pmc_*.json"))):
        wl = os.path.basename(bj)[4:-5]
        try:
            b = json.loads(open(bj).read().strip().splitlines()[-1])
        except (ValueError, IndexError):
            print("no bench line for", wl)
            continue
        frames = b["config"]["frames_per_step_per_gpu"]
        n_images = b["config"]["pyramids_per_step_per_gpu"]
        cmd = "python bench.py --workload %s --frames %d --steps 3 --warmup 1 --no-cpu-baseline --validate 0 --sustain 0 --no-replay-leg --no-configs" % (wl, frames)
        if wl.startswith("replay"):  # the exact replay: the lock-step loop with the pairs resident; counters averaged over ALL its LK launches
            cmd = "python bench.py --mode sequences --workload kitti%s --seqs %d --steps 20 --warmup 4 --no-cpu-baseline --validate 0" % (wl[6:], frames)
        fetch, _ = counters(os.path.join(src, "pmc_%s_fetch" % wl))
        write, _ = counters(os.path.join(src, "pmc_%s_write" % wl))
        sq, nsq = counters(os.path.join(src, "pmc_%s_sq" % wl))
        if "FETCH_SIZE" in fetch and "WRITE_SIZE" in write:
            f, w = fetch["FETCH_SIZE"] * 1024.0, write["WRITE_SIZE"] * 1024.0
            rec = {"workload": wl, "frames_per_step": frames, "fetch_bytes_per_launch": f, "write_bytes_per_launch": w,
                   "hbm_bytes_per_launch": 2 * f + w, "hbm_bytes_per_launch_uncorrected": f + w,
                   "algorithmic_bytes_per_launch": b["roofline"]["bytes_per_launch"],
                   "source": "r06 (gpurun_out/%s): rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate runs of `%s`; KB -> bytes; "
                             "FETCH_SIZE doubled (gfx950 correction, profiles/r01_fetch_calibration.txt); WRITE_SIZE as reported" % (tag, cmd)}
            pf, npf = pyr_counters(os.path.join(src, "pmc_%s_fetch" % wl))
            pw, npw = pyr_counters(os.path.join(src, "pmc_%s_write" % wl))
            levels = 4 if not wl.endswith("l4") else 5
            if "FETCH_SIZE" in pf and "WRITE_SIZE" in pw and npf["FETCH_SIZE"] % levels == 0:
                steps_f, steps_w = npf["FETCH_SIZE"] / levels, npw["WRITE_SIZE"] / levels
                pfb, pwb = pf["FETCH_SIZE"] * 1024.0 / steps_f, pw["WRITE_SIZE"] * 1024.0 / steps_w
                rec["pyramid_stage"] = {"images_per_step": n_images, "fetch_bytes_per_step": 2 * pfb, "write_bytes_per_step": pwb,
                                        "hbm_bytes_per_step": 2 * pfb + pwb, "kernel": "pyr_pass_kernel (%d launches per step)" % levels}
            traffic_legs.append(rec)
        if "SQ_INSTS_VALU" in sq and "GRBM_GUI_ACTIVE" in sq and "SQ_WAVES" in sq:
            valu, waves, cyc = sq["SQ_INSTS_VALU"], sq["SQ_WAVES"], sq["GRBM_GUI_ACTIVE"] / 8.0
            per_cyc = cyc * 1024.0 / valu
            # per FEATURE = per point of the leg's own bench line, not per launched wave: the lock-step loop sizes its grid for the
            # largest feature set and a third of its waves leave at once (batch legs: waves == points to 1e-4)
            feats = float(b["roofline"].get("points_per_launch") or waves)
            issue_legs.append({"workload": wl, "frames_per_step": frames, "valu_instructions_per_launch": valu, "waves_per_launch": waves,
                               "points_per_launch": feats, "valu_instructions_per_feature": valu / feats,
                               "salu_instructions_per_feature": sq["SQ_INSTS_SALU"] / feats if "SQ_INSTS_SALU" in sq else None,
                               "lds_instructions_per_feature": sq["SQ_INSTS_LDS"] / feats if "SQ_INSTS_LDS" in sq else None,
                               "shader_cycles_per_launch": cyc, "simd_cycles_per_valu_instruction": per_cyc,
                               "issue_cost_bound_cycles_per_valu_instruction": ISSUE_COST, "measured_over_bound": per_cyc / ISSUE_COST,
                               "issue_floor_cycles_per_valu_instruction": ISSUE_FLOOR,
                               "dispatches_averaged": nsq["SQ_INSTS_VALU"],
                               "source": "r06 (gpurun_out/%s): rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU "
                                         "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES (its own run of `%s`), means over the lk_circular_kernel "
                                         "dispatches; the bound = the hot loop's opcode mix x measured issue costs "
                                         "(profiles/r02_lk_issue_bound.md)" % (tag, cmd)})
    for name, legs in (("lk_traffic.json", traffic_legs), ("lk_issue.json", issue_legs)):
        path = os.path.join(ROOT, "profiles", name)
        try:
            old = json.load(open(path))
        except (OSError, ValueError):
            old = {}
        head = [l for l in legs if l["workload"] == "kitti2000"]
        if head:  # the headline's record at the top level, as before
            old = dict(head[0])
        old["legs"] = legs
        json.dump(old, open(path, "w"), indent=1)
        print(name, [(l["workload"], l["frames_per_step"]) for l in legs])
    for l in traffic_legs:
        print("  %-10s x %3d: fetch %.3f GB (x 2) + write %.3f GB = %.3f GB per LK launch = %.2f x the algorithmic bytes" % (
            l["workload"], l["frames_per_step"], l["fetch_bytes_per_launch"] / 1e9, l["write_bytes_per_launch"] / 1e9,
            l["hbm_bytes_per_launch"] / 1e9, l["hbm_bytes_per_launch"] / l["algorithmic_bytes_per_launch"]))
    for l in issue_legs:
        print("  %-10s x %3d: %.0f VALU instructions per feature, %.2f SIMD-cycles each (bound %.2f)" % (
            l["workload"], l["frames_per_step"], l["valu_instructions_per_feature"], l["simd_cycles_per_valu_instruction"], ISSUE_COST))


if __name__ == "__main__":
    main(sys.argv[1])
