"""Timeline of the LAST N kernels of a rocprofv3 --kernel-trace CSV (start offset, duration, gap to the previous end):
where one synchronous call's time goes, kernel by kernel and launch gap by launch gap.

    rocprofv3 --kernel-trace --output-format csv -d out -- python tools/latency_mode.py track 6 30
    python tools/kernel_timeline.py out 45
    python tools/kernel_timeline.py out 60 lk_circular:3      # 60 kernels starting at the 3rd-last lk_circular_kernel
"""
import csv
import glob
import os
import sys


def main():
    d, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 45
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        raise SystemExit("no *kernel_trace.csv under " + d)
    rows = []
    with open(files[0]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:], r.get("Queue_Id", "")))
    rows.sort()
    if len(sys.argv) > 3:  # start at the k-th last kernel whose name contains the substring
        sub, k = sys.argv[3].split(":")
        hits = [i for i, r in enumerate(rows) if sub in r[2]]
        start = hits[-int(k)] if len(hits) >= int(k) else 0
        rows = rows[start:start + n]
    else:
        rows = rows[-n:]
    t0 = rows[0][0]
    prev_end = rows[0][0]
    print("%10s %9s %9s  %-6s %s" % ("start_us", "dur_us", "gap_us", "queue", "kernel"))
    for s, e, name, q in rows:
        print("%10.1f %9.1f %9.1f  %-6s %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, q, name))
        prev_end = max(prev_end, e)
    print("span %.1f us, kernel time %.1f us" % ((max(r[1] for r in rows) - t0) / 1e3, sum(r[1] - r[0] for r in rows) / 1e3))


if __name__ == "__main__":
    main()
