"""Turn one tools/gpu_round.sh session directory (gpurun_out/<tag>/) into the committed evidence under profiles/.

    python tools/profile_summary.py gpurun_out/<tag> profiles/r06

What it takes from the session (each part of gpu_round.sh that ran):
    tests     pytest.log                      -> <prefix>_pytest_gpu.txt   (the summary tail)
    bench     bench.json                      -> <prefix>_bench.json       (the ONE line `python bench.py` printed)
    prof      kernel_stats_batch.csv          -> <prefix>_kernel_stats_batch.csv  (rocprofv3 --kernel-trace --stats of the headline)
    latency   latency.log                     -> <prefix>_latency_mode.txt
    timeline  timeline.txt                    -> <prefix>_track_frame_timeline.txt
    pmclegs   lk_traffic.json, lk_issue.json  -> profiles/lk_traffic.json, profiles/lk_issue.json  (written by tools/pmc_legs.py;
                                                 bench.py reads them for roofline.traffic / valu_issue_frac)
    posepmc   pose_pmc.md                     -> <prefix>_pose_pmc_tables.md  (tools/pose_pmc.py)
and prints the cross-checks a reader of the bench line makes: the rocprofv3 average of the dominant kernel against the
HIP-event duration in the line, roofline.frac recomputed from bytes / time / peak, traffic against the algorithmic bytes.
(One script: rounds 2-4 had a generation each; the PMC tables of round 1's layout are tools/pmc_legs.py / tools/pose_pmc.py now.)
"""
import csv
import json
import os
import shutil
import sys


def main(src, prefix):
    os.makedirs(os.path.dirname(prefix) or ".", exist_ok=True)
    prof_dir = os.path.dirname(prefix) or "."
    copied = []

    def take(name, dst, tail=None):
        p = os.path.join(src, name)
        if not os.path.exists(p) or not os.path.getsize(p):
            return None
        if tail:
            open(dst, "w").write("".join(open(p).readlines()[-tail:]))
        else:
            shutil.copy(p, dst)
        copied.append(dst)
        return dst

    take("pytest.log", prefix + "_pytest_gpu.txt", tail=25)
    bench = take("bench.json", prefix + "_bench.json")
    stats = take("kernel_stats_batch.csv", prefix + "_kernel_stats_batch.csv")
    take("latency.log", prefix + "_latency_mode.txt")
    take("timeline.txt", prefix + "_track_frame_timeline.txt")
    take("pose_pmc.md", prefix + "_pose_pmc_tables.md")
    for name in ("lk_traffic.json", "lk_issue.json"):
        take(name, os.path.join(prof_dir, name))
    print("copied:", ", ".join(os.path.relpath(c) for c in copied) or "(nothing)")
    b = None
    if bench:
        b = json.loads(open(bench).read().strip().splitlines()[-1])
        r = b["roofline"]
        print("bench line: %.0f %s, %.3f ms per step, %d steps; %s launch %.3f ms; frac %.4f (recomputed %.4f); valu_issue_frac %s; "
              "traffic %s of %.3f GB algorithmic" % (
                  b["value"], b["unit"], b["ms_per_step"], b["steps"], r["kernel"], r["launch_ms"], r["frac"],
                  r["bytes_per_launch"] / (r["launch_ms"] * 1e-3) / 1e9 / r["peak"], r.get("valu_issue_frac"),
                  "%.3f GB" % (r["traffic"] / 1e9) if r.get("traffic") else "n/a", r["bytes_per_launch"] / 1e9))
        for c in b.get("configs", []):
            if c.get("value") is not None and c.get("unit") == "frames/s":
                print("  %-44s %9.0f frames/s  %7.3f ms per step  validated %s%s" % (
                    c["name"], c["value"], c["ms_per_step"], c.get("validated_frames"),
                    "  %.1f GB/s of PCIe" % c["pcie_gb_s"] if c.get("pcie_gb_s") else ""))
    if stats:
        rows = list(csv.DictReader(open(stats)))
        print("kernel trace (rocprofv3 --kernel-trace --stats), top of %d kernels:" % len(rows))
        for r in rows[:8]:
            print("  %-44s calls %5s  avg %10.1f us  %5.1f %%" % (r["Name"].split("(")[0].replace("void ", "")[:44], r["Calls"],
                                                                float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
        if b:
            k = [r for r in rows if b["roofline"]["kernel"] in r["Name"]]
            if k:
                avg = float(k[0]["AverageNs"]) / 1e6
                print("cross-check: %s %.3f ms by rocprofv3 against %.3f ms by HIP events in the bench line (%+.1f %%)" % (
                    b["roofline"]["kernel"], avg, b["roofline"]["launch_ms"], 100.0 * (avg / b["roofline"]["launch_ms"] - 1.0)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
