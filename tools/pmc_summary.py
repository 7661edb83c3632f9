"""Summarise rocprofv3 --pmc passes (one directory per pass, each holding *_counter_collection.csv) into one per-kernel table.

usage: python tools/pmc_summary.py OUT.csv [--traffic OUT.json --kernel lsim_k_step_a --task aliengo --envs 4096 --solver tgs --append 1 --note "..."]
                                   [--run MODE:ACTIONS:DIR,DIR,...]...  [PASS_DIR ...]
--solver: which kernel A the passes ran (tgs / pgs; bench.py matches it).  --append 1: keep the records of OUT.json that describe other workloads.

Plain PASS_DIRs are merged into the table OUT.csv.  Each --run names the workload its passes were collected on (bench.py --mode MODE with
ACTIONS = policy | normal | zeros) and becomes one record of OUT.json ("runs"), which bench.py matches against its own workload before it
quotes `roofline.traffic` / `valu_issue_frac`; the first run's passes also fill OUT.csv when no plain PASS_DIR is given.

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KB.  Per /opt/skills/guides/MI355X_MICROARCH.md (HBM section) FETCH_SIZE on
gfx950 tallies 128-B requests at 64 B, so the traffic figure doubles it; WRITE_SIZE is taken as reported.
Effective shader clock of a pass = GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel duration (same csv's timestamps), the guide's
"DVFS give-back" recipe; the VALU issue fraction is priced at that clock, not at the 2.4 GHz maximum.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

N_XCD = 8


def collect(dirs, match="lsim_k_"):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))   # kernel -> counter -> [sum, dispatches]
    dur = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))   # kernel -> counter (pass) -> [sum of durations ns, dispatches]
    regs = {}
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            per_dispatch = defaultdict(float)
            span = {}
            meta = {}
            for row in csv.DictReader(open(path)):
                name = row["Kernel_Name"].split("(")[0]
                if match not in name:
                    continue
                key = (row["Dispatch_Id"], name, row["Counter_Name"])
                per_dispatch[key] += float(row["Counter_Value"])      # counters arrive per XCD / dimension: sum them
                span[key] = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
                meta[name] = (row["VGPR_Count"], row["Accum_VGPR_Count"], row["SGPR_Count"], row["LDS_Block_Size"], row["Scratch_Size"])
            for (did, name, ctr), v in per_dispatch.items():
                a = acc[name][ctr]
                a[0] += v
                a[1] += 1
                t = dur[name][ctr]
                t[0] += span[(did, name, ctr)]
                t[1] += 1
            regs.update(meta)
    return acc, dur, regs


def table(acc, regs, note):
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
    return lines


def record(acc, dur, kernel, task, envs, mode, actions, note, solver="tgs"):
    k = next(x for x in sorted(acc) if kernel in x)
    avg = lambda c: (acc[k][c][0] / acc[k][c][1]) if c in acc[k] and acc[k][c][1] else None   # noqa: E731
    fetch_kb, write_kb = avg("FETCH_SIZE"), avg("WRITE_SIZE")
    gui = avg("GRBM_GUI_ACTIVE")
    clk = None
    if gui and dur[k]["GRBM_GUI_ACTIVE"][1]:
        d_ns = dur[k]["GRBM_GUI_ACTIVE"][0] / dur[k]["GRBM_GUI_ACTIVE"][1]
        clk = gui / N_XCD / (d_ns * 1e-9)
    lanes = None
    if avg("SQ_THREAD_CYCLES_VALU") and avg("SQ_ACTIVE_INST_VALU"):
        lanes = avg("SQ_THREAD_CYCLES_VALU") / avg("SQ_ACTIVE_INST_VALU")       # both in quad-cycles: mean active lanes of a VALU instruction
    return {"kernel": k, "task": task, "envs_per_gpu": envs, "mode": mode, "actions": actions, "solver": solver,
            "valu_wave_insts_per_launch": avg("SQ_INSTS_VALU"), "salu_wave_insts_per_launch": avg("SQ_INSTS_SALU"),
            "lds_wave_insts_per_launch": avg("SQ_INSTS_LDS"), "wave_cycles_quad_per_launch": avg("SQ_WAVE_CYCLES"),
            "wait_any_quad_per_launch": avg("SQ_WAIT_ANY"), "wait_inst_any_quad_per_launch": avg("SQ_WAIT_INST_ANY"),
            "mean_active_lanes_per_valu_inst": lanes,
            "fetch_size_kb_raw": fetch_kb, "write_size_kb_raw": write_kb,
            "traffic_bytes_per_launch": ((2.0 * fetch_kb + write_kb) * 1024.0) if fetch_kb is not None and write_kb is not None else None,
            "grbm_gui_active_per_launch": gui, "effective_clock_hz": clk,
            "correction": "FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B); WRITE_SIZE as reported; clock = GRBM_GUI_ACTIVE / 8 XCDs / duration",
            "note": note}


def main(argv):
    out_csv = argv[0]
    traffic_json = kernel = note = task = envs = None
    solver, append = "tgs", False
    runs = []
    rest = argv[1:]
    while rest and rest[0].startswith("--"):
        if rest[0] == "--traffic": traffic_json = rest[1]
        elif rest[0] == "--kernel": kernel = rest[1]
        elif rest[0] == "--note": note = rest[1]
        elif rest[0] == "--task": task = rest[1]
        elif rest[0] == "--envs": envs = int(rest[1])
        elif rest[0] == "--solver": solver = rest[1]
        elif rest[0] == "--append": append = rest[1] not in ("0", "")
        elif rest[0] == "--run":
            mode, actions, dirs = rest[1].split(":", 2)
            runs.append((mode, actions, dirs.split(",")))
        rest = rest[2:]
    table_dirs = rest if rest else (runs[0][2] if runs else [])
    acc, dur, regs = collect(table_dirs)
    lines = table(acc, regs, note)
    open(out_csv, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
    if traffic_json and kernel:
        recs = []
        for mode, actions, dirs in (runs or [("env", "normal", rest)]):
            a, d, _ = collect(dirs)
            if a:
                recs.append(record(a, d, kernel, task, envs, mode, actions, note, solver))
                recs[-1]["file"] = os.path.basename(out_csv)
        if append and os.path.exists(traffic_json):
            key = lambda r: (r.get("task"), r.get("envs_per_gpu"), r.get("mode"), r.get("actions"), r.get("solver", "pgs"))   # noqa: E731
            new = {key(r) for r in recs}
            recs = [r for r in json.load(open(traffic_json)).get("runs", []) if key(r) not in new] + recs
        json.dump({"runs": recs}, open(traffic_json, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1:])
