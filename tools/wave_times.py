"""Per-wave start / end times of kernel A (diagnostics; not part of the product build).

Builds isaacgymloco_amd/csrc/variants/liblsim_wavetimes.so with -DLS_WAVE_TIMES (lane 0 of every wave records the 100 MHz wall clock at
its first and last instruction, its shader-clock ticks and its hardware id), runs the env-only loop and prints, for a few steps, how
the waves' durations and end times are distributed -- is the launch as long as its mean wave or as its slowest one -- and which
property of a robot (contacts, reset, terrain type) or of its place on the chip (XCD, CU, SIMD) the slow ones share.
Run on the GPU box:  python tools/wave_times.py [task] [N]"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "isaacgymloco_amd", "csrc", "variants", "liblsim_wavetimes.so")


def build():
    from isaacgymloco_amd.csrc import build as B
    return B.build_variant(OUT, ["-DLS_WAVE_TIMES"])


PHASES = ["torques + kinematics", "body inertias + bias forces", "leg composites", "leg block", "leg Schur", "base assemble", "base factor", "free leg",
          "free base", "free finish + narrow phase", "contact compaction + limit rows", "constraint rows", "Delassus + TGS sub-iterations + contact forces",
          "integrator"]


def phases(task, N):
    """--phases: the 14 phases of sub-step 1, product code path (a -DLS_WAVE_TIMES=2 build: the 16 checkpoints sit behind those phases)"""
    out = OUT.replace("wavetimes", "wavephases")
    from isaacgymloco_amd.csrc import build as B
    if B.variant_is_stale(out):
        B.build_variant(out, ["-DLS_WAVE_TIMES=2"])
    os.environ["LSIM_LIB"] = out
    import torch
    from isaacgymloco_amd import lib
    from isaacgymloco_amd.envs import config as C
    from isaacgymloco_amd.envs.legged_robot import LeggedRobot
    cfg = C.TASKS[task][0]()
    cfg.env.num_envs = N
    env = LeggedRobot(cfg, sim_device="cuda:0", seed=1)
    env.reset()
    L = lib.load()
    g = torch.Generator(device="cuda:0").manual_seed(0)
    acc = np.zeros(14)
    n = 0
    tb = (ctypes.c_ulonglong * (4 * N))(); cpb = (ctypes.c_uint * (16 * N))()
    for t in range(260):
        env.step_device(torch.randn(N, 12, device="cuda:0", generator=g))
        if t < 200 or t % 10:
            continue
        assert L.lsim_debug_read_wave_times(tb, N) == 0 and L.lsim_debug_read_wave_checkpoints(cpb, N) == 0
        a = np.frombuffer(tb, dtype=np.uint64).reshape(N, 4).astype(np.float64)
        ghz = np.median(a[:, 2] / ((a[:, 1] - a[:, 0]) / 100.0) / 1e3)
        cp = np.frombuffer(cpb, dtype=np.uint32).reshape(N, 16).astype(np.float64)
        acc += np.diff(cp[:, :15], axis=1).mean(0) / (ghz * 1e3)
        n += 1
    acc /= n
    print(f"task {task} N {N}: phases of sub-step 1, mean us per wave (sum {acc.sum():.2f})")
    for nm, v in zip(PHASES, acc):
        print(f"  {v:6.2f}  {nm}")


def main():
    args = [x for x in sys.argv[1:] if not x.startswith("--")]
    task = args[0] if len(args) > 0 else "aliengo"
    N = int(args[1]) if len(args) > 1 else 4096
    if "--phases" in sys.argv:
        return phases(task, N)
    from isaacgymloco_amd.csrc.build import variant_is_stale
    if variant_is_stale(OUT):
        build()
    os.environ["LSIM_LIB"] = OUT
    import torch
    from isaacgymloco_amd import lib
    from isaacgymloco_amd.envs import config as C
    from isaacgymloco_amd.envs.legged_robot import LeggedRobot
    cfg = C.TASKS[task][0]()
    cfg.env.num_envs = N
    env = LeggedRobot(cfg, sim_device="cuda:0", seed=1)
    env.reset()
    L = lib.load()
    g = torch.Generator(device="cuda:0").manual_seed(0)
    buf = (ctypes.c_ulonglong * (4 * N))()
    print(f"task {task} N {N}: times in us relative to the first wave's start")
    for t in range(260):
        env.step_device(torch.randn(N, 12, device="cuda:0", generator=g))
        if t < 200 or t % 20:
            continue
        assert L.lsim_debug_read_wave_times(buf, N) == 0
        a = np.frombuffer(buf, dtype=np.uint64).reshape(N, 4).astype(np.int64)
        t0 = a[:, 0].min()
        start, end = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0
        dur = end - start
        hw = a[:, 3]
        simd, cu, se, xcc = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 13) & 7, (hw >> 16) & 15
        nc = env.buf["contact_count"][:, 0].cpu().numpy()
        rst = env.reset_buf.cpu().numpy().astype(bool)
        q = lambda x: " ".join(f"{v:6.1f}" for v in np.percentile(x, [0, 10, 50, 90, 99, 100]))
        print(f"step {t}: launch {end.max():6.1f} | start p0/10/50/90/99/100 {q(start)} | end {q(end)} | duration {q(dur)} | clock {np.median(a[:, 2] / dur / 1e3):.2f} GHz")
        slow = dur >= np.percentile(dur, 95)
        print(f"          slowest 5 %: contacts {nc[slow].mean():.1f} (all {nc.mean():.1f}), resetting {rst[slow].mean():.2f} (all {rst.mean():.2f}); "
              f"duration vs contacts r = {np.corrcoef(dur, nc)[0, 1]:.2f}")
        # a SIMD's four waves: how different are they, and how different are the SIMDs
        key = ((xcc * 8 + se) * 16 + cu) * 4 + simd
        order = np.argsort(key, kind="stable")
        ks, ds, es = key[order], dur[order], end[order]
        uniq, idx, cnt = np.unique(ks, return_index=True, return_counts=True)
        per = np.array([es[i:i + c].max() for i, c in zip(idx, cnt)])
        mean_in = np.array([ds[i:i + c].mean() for i, c in zip(idx, cnt)])
        print(f"          {len(uniq)} SIMDs seen, waves per SIMD {cnt.min()}..{cnt.max()}; last end per SIMD {q(per)}; mean duration per SIMD {q(mean_in)}")
        rows = " ".join(f"{k}:{dur[(nc == k) & ~rst].mean():.1f}({int(((nc == k) & ~rst).sum())})" for k in range(0, 13) if ((nc == k) & ~rst).sum() > 4)
        print(f"          mean duration by contacts (not resetting): {rows}; resetting: {dur[rst].mean() if rst.any() else 0:.1f} ({int(rst.sum())})")
        worst = np.argsort(-dur)[:12]
        print("          slowest 12: " + " ".join(f"{dur[i]:.0f}us/c{nc[i]}{'R' if rst[i] else ''}/x{xcc[i]}" for i in worst))
        mates = np.array([ds[i:i + c].sum() for i, c in zip(idx, cnt)])
        print(f"          sum of the four durations per SIMD {q(mates)}; r(last end per SIMD, that sum) = {np.corrcoef(per, mates)[0, 1]:.2f}")
        cpb = (ctypes.c_uint * (16 * N))()
        if L.lsim_debug_read_wave_checkpoints(cpb, N) == 0:
            cp = np.frombuffer(cpb, dtype=np.uint32).reshape(N, 16).astype(np.float64)
            ghz = np.median(a[:, 2] / dur / 1e3)
            names = ["load", "sub-step 0", "sub-step 1", "sub-step 2", "sub-step 3", "final kinematics + load issue", "body states + loads consumed", "state stores",
                     "derived state + callback", "termination + rewards", "term. obs + reset path", "observation build", "observation / last_* stores"]
            seg = np.diff(np.concatenate([np.zeros((N, 1)), cp[:, :13]], axis=1), axis=1) / (ghz * 1e3)
            if rst.any():       # inside the reset path (checkpoints 13-15 sit between 9 and 10)
                r = cp[rst] / (ghz * 1e3)
                print(f"          resetting waves, mean us: termination obs build + rows {np.mean(r[:, 13] - r[:, 9]):.1f}; tail set-up + curriculum + new state "
                      f"{np.mean(r[:, 14] - r[:, 13]):.1f}; terrain under the new pose {np.mean(r[:, 15] - r[:, 14]):.1f}; stores + episode sums {np.mean(r[:, 10] - r[:, 15]):.1f}")
            ok = ~rst
            print("          segments, mean us (not resetting | resetting): " + "; ".join(f"{nm} {seg[ok, i].mean():.1f}|{seg[rst, i].mean() if rst.any() else 0:.1f}" for i, nm in enumerate(names)))
        byx = [f"{end[xcc == x].max():.1f}" for x in np.unique(xcc)]
        print(f"          last end per XCD: {' '.join(byx)}")


if __name__ == "__main__":
    main()
