"""Summarise a rocprofv3 rocpd SQLite database (kernel-trace) into per-kernel statistics (like --stats CSV)."""
import sqlite3
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    tables = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in tables if t.startswith("rocpd_kernel_dispatch"))
    ks = next(t for t in tables if t.startswith("rocpd_info_kernel_symbol"))
    cols = [r[1] for r in db.execute(f"pragma table_info({kd})")]
    scols = [r[1] for r in db.execute(f"pragma table_info({ks})")]
    name_col = "kernel_name" if "kernel_name" in scols else ("display_name" if "display_name" in scols else scols[-1])
    q = f"select s.{name_col}, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) " \
        f"from {kd} d join {ks} s on d.kernel_id = s.id group by s.{name_col} order by 3 desc"
    rows = list(db.execute(q))
    total = sum(r[2] for r in rows) or 1
    lines = ["Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs"]
    for n, c, t, a, mn, mx in rows:
        lines.append(f'"{n}",{c},{t},{a:.1f},{100.0 * t / total:.2f},{mn},{mx}')
    text = "\n".join(lines)
    print(text)
    if out:
        open(out, "w").write(text + "\n")
    _ = cols


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
