"""Summarise a rocprofv3 rocpd database (kernel dispatches) into a per-kernel table (like --stats).
usage: python tools/prof_summary.py gpurun_out/prof/bench_results.db [out.md]"""
import re
import sqlite3
import sys


def main(db, out=None):
    con = sqlite3.connect(db)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    sym = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    cols = [r[1] for r in cur.execute(f"pragma table_info({disp})")]
    scols = [r[1] for r in cur.execute(f"pragma table_info({sym})")]
    name_col = "display_name" if "display_name" in scols else ("kernel_name" if "kernel_name" in scols else scols[-1])
    q = f"select s.{name_col}, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start) " \
        f"from {disp} d join {sym} s on d.kernel_id = s.id group by s.{name_col} order by 3 desc"
    rows = list(cur.execute(q))
    total = sum(r[2] for r in rows)
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for n, c, t, mn, mx in rows:
        n = re.sub(r"\(.*", "", n)
        n = n.replace("void ", "")
        lines.append(f"| {n[:90]} | {c} | {t/1e6:.3f} | {t/c/1e3:.2f} | {mn/1e3:.2f} | {mx/1e3:.2f} | {100*t/total:.1f} |")
    lines.append(f"| TOTAL | {sum(r[1] for r in rows)} | {total/1e6:.3f} | | | | 100 |")
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main(*sys.argv[1:3])
