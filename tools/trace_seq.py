"""Print the last N kernels (name, duration) of a rocprofv3 kernel-trace CSV in launch order: the tail of the last minibatch of
tools/update_probe.py.  usage: python tools/trace_seq.py TRACE.csv [N]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 160
prev = None
for r in rows[-n:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"]
    name = re.sub(r"at::native::|\(anonymous namespace\)::|void |std::array<char\*, \d+ul>|c10::|at::", "", name)
    gap = (s - prev) / 1e3 if prev else 0.0
    print(f"{(e - s) / 1e3:8.1f} us  gap {gap:6.1f}  {name[:150]}")
    prev = e
