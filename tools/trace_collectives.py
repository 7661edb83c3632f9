"""Where does a collective sit on the device timeline?  For every RCCL kernel of a rocprofv3 kernel-trace CSV: its duration, how long the
device had been idle when it started (no other kernel running), how long it stayed idle after it ended, and the kernels around it.
usage: python tools/trace_collectives.py TRACE_kernel_trace.csv [max rows]"""
import csv, re, sys, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
short = lambda n: re.sub(r"at::native::|\(anonymous namespace\)::|void |c10::|at::", "", n)[:70]
is_coll = lambda n: "nccl" in n.lower() or "rccl" in n.lower()
out = []
for i, r in enumerate(rows):
    if not is_coll(r["Kernel_Name"]):
        continue
    before_end = max((q["e"] for q in rows[max(0, i - 40):i]), default=r["s"])
    after = [q for q in rows[i + 1:i + 40] if q["s"] >= r["s"]]
    overl = [q for q in after if q["s"] < r["e"]]
    nxt = min((q["s"] for q in after if q["s"] >= r["e"]), default=r["e"])
    out.append(dict(dur=(r["e"] - r["s"]) / 1e3, idle_before=max(0, r["s"] - before_end) / 1e3, idle_after=max(0, nxt - r["e"]) / 1e3 if not overl else 0.0,
                    overlapped=len(overl), prev=short(rows[i - 1]["Kernel_Name"]) if i else "", name=short(r["Kernel_Name"])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
for o in out[-n:]:
    print(f"dur {o['dur']:7.1f} us  idle before {o['idle_before']:6.1f}  idle after {o['idle_after']:6.1f}  kernels under it {o['overlapped']:2d}  after: {o['prev']}")
if out:
    med = lambda k: statistics.median(o[k] for o in out)
    print(f"collectives {len(out)}: median duration {med('dur'):.1f} us, idle before {med('idle_before'):.1f} us, idle after {med('idle_after'):.1f} us; "
          f"exposed per collective (median of sum) {statistics.median(o['idle_before'] + o['idle_after'] + (o['dur'] if not o['overlapped'] else 0) for o in out):.1f} us")
