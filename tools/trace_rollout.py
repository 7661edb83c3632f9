"""List the kernels of ONE rollout step (between two consecutive simulator launches) from a rocprofv3 kernel-trace CSV of bench.py."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
idx = [i for i, r in enumerate(rows) if "lsim_k_step_a" in r["Kernel_Name"]]
k = len(idx) // 2 + (int(sys.argv[2]) if len(sys.argv) > 2 else 10)
i0, i1 = idx[k], idx[k + 1]
seg = rows[i0:i1]
print("step span us", (seg[-1]["e"] - seg[0]["s"]) / 1e3, "busy us", sum(r["e"] - r["s"] for r in seg) / 1e3, "kernels", len(seg))
for r in seg:
    n = re.sub(r"\(.*", "", r["Kernel_Name"])
    n = n if not n.startswith("void at::native") else r["Kernel_Name"][18:110]
    print(f"{(r['s'] - seg[0]['s']) / 1e3:8.1f} +{(r['e'] - r['s']) / 1e3:6.1f} us  {n[:100]}")
