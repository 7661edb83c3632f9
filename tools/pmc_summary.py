"""Summarise rocprofv3 --pmc passes (one directory per pass, each holding *_counter_collection.csv) into one per-kernel table.

usage: python tools/pmc_summary.py OUT.csv [--traffic OUT.json --kernel lsim_k_step_a --task aliengo --envs 4096 --note "..."] PASS_DIR [PASS_DIR ...]

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KB.  Per /opt/skills/guides/MI355X_MICROARCH.md (HBM section) FETCH_SIZE on
gfx950 tallies 128-B requests at 64 B, so the traffic figure doubles it; WRITE_SIZE is taken as reported.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def collect(dirs, match="lsim_k_"):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))   # kernel -> counter -> [sum, dispatches]
    regs = {}
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            per_dispatch = defaultdict(float)
            meta = {}
            for row in csv.DictReader(open(path)):
                name = row["Kernel_Name"].split("(")[0]
                if match not in name:
                    continue
                key = (row["Dispatch_Id"], name, row["Counter_Name"])
                per_dispatch[key] += float(row["Counter_Value"])      # counters arrive per XCD / dimension: sum them
                meta[name] = (row["VGPR_Count"], row["Accum_VGPR_Count"], row["SGPR_Count"], row["LDS_Block_Size"], row["Scratch_Size"])
            for (_, name, ctr), v in per_dispatch.items():
                a = acc[name][ctr]
                a[0] += v
                a[1] += 1
            regs.update(meta)
    return acc, regs


def main(argv):
    out_csv = argv[0]
    traffic_json = kernel = note = task = envs = None
    rest = argv[1:]
    while rest and rest[0].startswith("--"):
        if rest[0] == "--traffic": traffic_json = rest[1]
        elif rest[0] == "--kernel": kernel = rest[1]
        elif rest[0] == "--note": note = rest[1]
        elif rest[0] == "--task": task = rest[1]
        elif rest[0] == "--envs": envs = int(rest[1])
        rest = rest[2:]
    acc, regs = collect(rest)
    kernels = sorted(acc)
    counters = sorted({c for k in kernels for c in acc[k]})
    lines = ["counter," + ",".join(f"{k}_avg_per_launch" for k in kernels) + ",launches"]
    for c in counters:
        vals = [(acc[k][c][0] / acc[k][c][1]) if acc[k][c][1] else float("nan") for k in kernels]
        n = max(acc[k][c][1] for k in kernels)
        lines.append(c + "," + ",".join(f"{v:.1f}" for v in vals) + f",{n}")
    for k in kernels:
        lines.append(f"# {k}: VGPR={regs[k][0]} AGPR={regs[k][1]} SGPR={regs[k][2]} LDS={regs[k][3]} B scratch={regs[k][4]} B/lane")
    if note:
        lines.append("# " + note)
    open(out_csv, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
    if traffic_json and kernel:
        k = next(x for x in kernels if kernel in x)
        f = acc[k]["FETCH_SIZE"]; w = acc[k]["WRITE_SIZE"]
        fetch_kb = f[0] / f[1]; write_kb = w[0] / w[1]
        vi = acc[k].get("SQ_INSTS_VALU")
        json.dump({"kernel": kernel, "task": task, "envs_per_gpu": envs, "valu_wave_insts_per_launch": (vi[0] / vi[1]) if vi and vi[1] else None, "fetch_size_kb_raw": fetch_kb, "write_size_kb_raw": write_kb,
                   "traffic_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0,
                   "correction": "FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B); WRITE_SIZE as reported",
                   "note": note}, open(traffic_json, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1:])
