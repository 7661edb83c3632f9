"""Device idle time inside the last PPO update of a rocprofv3 kernel-trace CSV: span, busy (union over streams), idle, and the kernel pairs
the largest idle gaps sit between.  usage: python tools/trace_idle.py TRACE_kernel_trace.csv [top]"""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
short = lambda n: re.sub(r"at::native::|\(anonymous namespace\)::|void |c10::|at::|std::array<char\*, \d+ul>", "", n)[:44]
last = max(i for i, r in enumerate(rows) if "lsim_k_step_a" in r["Kernel_Name"])
upd = rows[last + 1:]
gaps, cur_end = [], upd[0]["e"]
for i in range(1, len(upd)):
    if upd[i]["s"] > cur_end:
        gaps.append((upd[i]["s"] - cur_end, i))
    cur_end = max(cur_end, upd[i]["e"])
span = (cur_end - upd[0]["s"]) / 1e6
idle = sum(g for g, _ in gaps) / 1e6
print(f"update: {len(upd)} kernels, span {span:.2f} ms, idle {idle:.2f} ms in {len(gaps)} gaps (> 10 us: {sum(1 for g, _ in gaps if g > 10000)}, their sum {sum(g for g, _ in gaps if g > 10000) / 1e6:.2f} ms)")
c = collections.Counter(); n = collections.Counter()
for g, i in gaps:
    k = (short(upd[i - 1]["Kernel_Name"]), short(upd[i]["Kernel_Name"]))
    c[k] += g; n[k] += 1
for (a, b), g in c.most_common(int(sys.argv[2]) if len(sys.argv) > 2 else 14):
    print(f"{g / 1e3:8.1f} us in {n[(a, b)]:3d} gaps  after [{a}] before [{b}]")
