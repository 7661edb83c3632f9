#!/usr/bin/env python3
"""Register / scratch / LDS budget of every kernel of the library, from the compiler's own metadata (no GPU needed).

    python tools/kernel_regs.py [--unit lsim_learn.hip] [--match wgrad] [--json out.json]

Compiles one translation unit of isaacgymloco_amd/csrc to gfx950 assembly with the flags build.py uses and reads the
amdhsa.kernels notes: arch VGPRs, AGPRs, scratch bytes per lane, spilled VGPRs / SGPRs, LDS bytes.  (the hottest
weight-gradient kernel spilled 40 B per lane next to an unused half of its register budget; this is the check.)"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def demangle(names):
    try:
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def kernel_table(unit, extra=()):
    from isaacgymloco_amd.csrc import build as B
    flags = next(f for s, _, f in B.UNITS if s == unit)
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "unit.s")
        cmd = [os.environ.get("HIPCC", "hipcc")] + B.FLAGS + flags + list(extra) + ["--cuda-device-only", "-S", os.path.join(B.HERE, unit), "-o", asm]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        text = open(asm).read()
    # the amdhsa.kernels YAML: one "- .agpr_count:" ... block per kernel
    recs = []
    for blk in re.split(r"\n  - \.agpr_count:", text)[1:]:
        blk = ".agpr_count:" + blk
        get = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, None])[1]
        name = get("name")
        if name is None:
            continue
        recs.append({"name": name, "vgpr": int(get("vgpr_count")), "agpr": int(get("agpr_count")), "sgpr": int(get("sgpr_count")),
                     "scratch_bytes_per_lane": int(get("private_segment_fixed_size")), "vgpr_spills": int(get("vgpr_spill_count")),
                     "sgpr_spills": int(get("sgpr_spill_count")), "lds_bytes": int(get("group_segment_fixed_size")),
                     "max_flat_workgroup_size": int(get("max_flat_workgroup_size"))})
    dm = demangle([r["name"] for r in recs])
    for r in recs:
        r["kernel"] = re.sub(r"\(.*", "", dm.get(r["name"], r["name"]).replace("void ", ""))
    return recs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--unit", default="lsim_learn.hip")
    ap.add_argument("--match", default="")
    ap.add_argument("--json", default=None)
    ap.add_argument("--flags", default="", help="extra compiler flags")
    a = ap.parse_args()
    recs = [r for r in kernel_table(a.unit, a.flags.split()) if a.match in r["kernel"]]
    print(f"{'kernel':70s} vgpr agpr sgpr scratch vspill sspill    lds")
    for r in recs:
        print(f"{r['kernel'][:70]:70s} {r['vgpr']:4d} {r['agpr']:4d} {r['sgpr']:4d} {r['scratch_bytes_per_lane']:7d} {r['vgpr_spills']:6d} "
              f"{r['sgpr_spills']:6d} {r['lds_bytes']:6d}")
    if a.json:
        json.dump(recs, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
