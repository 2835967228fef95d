"""Counters and resources of every kernel BEHIND LK (filter, triangulation, the pose chain) per bench leg
 -> a markdown table (profiles/r06_pose_pmc.md).

    python tools/pose_pmc.py gpurun_out/<tag>                  # written by `bash tools/gpu_round.sh <tag> posepmc`
    python tools/pose_pmc.py --trace <kernel_trace.csv> out    # helper: one row of resources per kernel of a trace

Input per workload W: pose_W_kernel_stats.csv (rocprofv3 --kernel-trace --stats), pose_W_resources.csv (VGPR / scratch /
LDS / grid of each kernel, from the trace), posepmc_W_{a,b,fetch,write}/**/_counter_collection.csv -- every counter group
its own rocprofv3 run of `python bench.py --workload W --steps 3 --warmup 1 <lean>`.  Per kernel: mean per dispatch.
Derived columns:
  VALU/wave      SQ_INSTS_VALU / SQ_WAVES
  VMEM/wave      (SQ_INSTS_VMEM_RD + SQ_INSTS_VMEM_WR) / SQ_WAVES -- scratch_load / scratch_store count here (spill traffic)
  busy%          SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES (quad-cycles both): share of a resident wave's time an instruction of
                 it is being executed
  wait%          SQ_WAIT_ANY / SQ_WAVE_CYCLES: parked on s_waitcnt (memory / scratch latency)
  vmem%          SQ_ACTIVE_INST_VMEM / SQ_WAVE_CYCLES
  HBM KB         FETCH_SIZE (doubled: MI355X_MICROARCH.md, gfx950 correction) + WRITE_SIZE, KB per dispatch
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    n = name.split("(")[0].replace("void ", "").strip()
    return n.replace("vo::", "")


SKIP = ("lk_circular_kernel", "pyr_pass_kernel", "fast_", "bucket_kernel", "at::", "elementwise", "fill", "Memset", "copy")


def trace_resources(trace_csv, out_csv):
    res = {}
    for r in csv.DictReader(open(trace_csv)):
        k = short(r["Kernel_Name"])
        if k not in res:
            res[k] = [r.get("VGPR_Count", ""), r.get("Accum_VGPR_Count", ""), r.get("SGPR_Count", ""), r.get("LDS_Block_Size", ""),
                      r.get("Scratch_Size", ""), r.get("Workgroup_Size_X", ""), r.get("Grid_Size_X", "")]
    with open(out_csv, "w", newline="") as f:
        wr = csv.writer(f)
        wr.writerow(["kernel", "vgpr", "agpr", "sgpr", "lds", "scratch", "wg", "grid"])
        for k, v in res.items():
            wr.writerow([k] + [str(x) for x in v])


def read_resources(path):
    """kernel -> row; tolerant of the first version's unquoted names (`epnp_kernel<1, false>` holds a comma): the seven numbers
    are taken from the right"""
    res = {}
    if not os.path.exists(path):
        return res
    rows = list(csv.reader(open(path)))
    for r in rows[1:]:
        if len(r) < 8:
            continue
        name = ",".join(r[:len(r) - 7])
        res[name] = dict(zip(["vgpr", "agpr", "sgpr", "lds", "scratch", "wg", "grid"], r[-7:]))
    return res


def counters(d):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


def main(src):
    out = []
    for st in sorted(glob.glob(os.path.join(src, "pose_*_kernel_stats.csv"))):
        wl = os.path.basename(st)[5:-len("_kernel_stats.csv")]
        stats = {short(r["Name"]): r for r in csv.DictReader(open(st))}
        resf = os.path.join(src, "pose_%s_resources.csv" % wl)
        res = read_resources(resf)
        c = defaultdict(dict)
        for name in ("a", "b", "fetch", "write"):
            for k, v in counters(os.path.join(src, "posepmc_%s_%s" % (wl, name))).items():
                c[k].update(v)
        total = sum(float(r["TotalDurationNs"]) for r in stats.values())
        out.append("## %s\n" % wl)
        out.append("| kernel | calls | avg us | % of GPU time | VGPR | scratch B/lane | LDS | waves | VALU/wave | VMEM/wave | LDS/wave | busy% | wait% | vmem% | HBM KB |")
        out.append("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
        for k, r in sorted(stats.items(), key=lambda kv: -float(kv[1]["TotalDurationNs"])):
            if any(s in k for s in SKIP) and "lk_circular" not in k:
                continue
            cc = c.get(k, {})
            w = cc.get("SQ_WAVES", 0.0)
            wc = cc.get("SQ_WAVE_CYCLES", 0.0)
            rr = res.get(k, {})

            def per_wave(*names):
                return "%.0f" % (sum(cc.get(n, 0.0) for n in names) / w) if w else ""

            def pct(name):
                return "%.0f" % (100.0 * cc[name] / wc) if wc and name in cc else ""
            hbm = ""
            if "FETCH_SIZE" in cc or "WRITE_SIZE" in cc:
                hbm = "%.0f" % (2.0 * cc.get("FETCH_SIZE", 0.0) + cc.get("WRITE_SIZE", 0.0))
            out.append("| `%s` | %s | %.1f | %.1f | %s | %s | %s | %.0f | %s | %s | %s | %s | %s | %s | %s |" % (
                k, r["Calls"], float(r["AverageNs"]) / 1e3, 100.0 * float(r["TotalDurationNs"]) / total, rr.get("vgpr", ""),
                rr.get("scratch", ""), rr.get("lds", ""), w, per_wave("SQ_INSTS_VALU"), per_wave("SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"),
                per_wave("SQ_INSTS_LDS"), pct("SQ_ACTIVE_INST_ANY"), pct("SQ_WAIT_ANY"), pct("SQ_ACTIVE_INST_VMEM"), hbm))
        out.append("")
    print("\n".join(out))


if __name__ == "__main__":
    if sys.argv[1] == "--trace":
        trace_resources(sys.argv[2], sys.argv[3])
    else:
        main(sys.argv[1])
