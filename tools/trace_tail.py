"""Summarise the kernels after the last simulator launch in a rocprofv3 kernel-trace CSV (i.e. the last GAE + update() of tools/update_probe.py)."""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
last = max(i for i, r in enumerate(rows) if "lsim_k_step_b" in r["Kernel_Name"])
upd = rows[last + 1:]
busy = sum(r["e"] - r["s"] for r in upd) / 1e6
print("kernels", len(upd), "span ms", (upd[-1]["e"] - upd[0]["s"]) / 1e6, "busy ms", busy)
d = collections.defaultdict(lambda: [0, 0])
for r in upd:
    n = r["Kernel_Name"]
    n = re.sub(r"\(.*", "", n)[:100] if not n.startswith("void at::native") else n[:150]
    d[n][0] += r["e"] - r["s"]; d[n][1] += 1
for k, (t, c) in sorted(d.items(), key=lambda kv: -kv[1][0])[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print(f"{t / 1e6:8.2f} ms {100 * t / 1e6 / busy:5.1f}% calls {c:5d} avg {t / c / 1e3:7.1f} us  {k}")
