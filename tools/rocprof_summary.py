"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) into per-kernel statistics
(markdown + csv) -> profiles/.   usage: python tools/rocprof_summary.py <results.db> <out_prefix>"""
import sqlite3
import sys


def main(db_path, out_prefix):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
    scols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    name_col = "display_name" if "display_name" in scols else ("kernel_name" if "kernel_name" in scols else scols[-1])
    q = ("select s.%s, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), "
         "max(d.end - d.start) from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
         "on d.kernel_id = s.id group by s.%s order by 3 desc" % (name_col, name_col))
    rows = list(cur.execute(q))
    total = sum(r[2] for r in rows) or 1
    extra = {}
    for c in ("vgpr_count", "accum_vgpr_count", "sgpr_count", "lds_block_size", "scratch_size", "workgroup_size_x", "grid_size_x"):
        if c in cols:
            extra[c] = c
    with open(out_prefix + ".md", "w") as f, open(out_prefix + ".csv", "w") as g:
        f.write("| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|\n")
        g.write("kernel,calls,total_ns,avg_ns,min_ns,max_ns,percent\n")
        for name, n, tot, avg, mn, mx in rows:
            short = name.split("(")[0]
            f.write("| %s | %d | %.3f | %.2f | %.2f | %.2f | %.2f |\n" % (short, n, tot / 1e6, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total))
            g.write('"%s",%d,%d,%.1f,%d,%d,%.3f\n' % (short, n, tot, avg, mn, mx, 100.0 * tot / total))
    print(open(out_prefix + ".md").read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
