"""Build liblsim.so (the HIP library behind include/lsim.h) in-tree for gfx950.

    python -m isaacgymloco_amd.csrc.build [--force]

hipcc cross-compiles without a GPU; the .so travels to the GPU box with the repo snapshot."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(HERE, "liblsim.so")
# two translation units, two flag sets: (source, headers it depends on, extra flags)
SIM_HEADERS = ["ls_math.h", "ls_shared.h", "ls_physics.h", "ls_post.h", "ls_kernels.h", "ls_api_impl.h"]
LEARN_HEADERS = ["ls_math.h", "ls_rollout.h", "ls_learn.h", "ls_gemm.h", "ls_policy.h", "ls_amp.h"]
# The simulator kernels' time is their vector instruction count, so ITS flags (SIM_FLAGS) are chosen for that:
# -fno-hip-fp32-correctly-rounded-divide-sqrt: ~150 divisions per sub-step cost ~12 instructions each when IEEE-rounded; quotients that
#   must be exact use ls_div_exact (ls_math.h)
# -fgpu-flush-denormals-to-zero: with fp32 denormals kept, every sqrtf / fast division carries a frexp / ldexp range-scaling sequence
#   (5-8 instructions instead of 1-2; 360 of kernel A's 7 400 static vector instructions); nothing in the path lives near 1e-38
# -fno-slp-vectorize: the SLP vectoriser pairs independent fp32 FMAs into v_pk_fma_f32 and pays two v_mov_b32 per pair to pack the
#   operands: more instructions than the scalar form it replaces (kernel A: 7 626 -> 7 401 static, 0.1245 -> 0.116 ms)
# The learner / rollout / policy kernels (lsim_learn.hip) keep IEEE-rounded division and square root and fp32 denormals, like the torch
# kernels they replace (Adam's exp_avg_sq of tiny gradients, sqrt / division in lsim_k_adam_apply and the normalisations).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-pass-failed", "-fno-slp-vectorize"]
SIM_FLAGS = ["-fno-hip-fp32-correctly-rounded-divide-sqrt", "-fgpu-flush-denormals-to-zero"]
UNITS = [("lsim_hip.hip", SIM_HEADERS, SIM_FLAGS), ("lsim_learn.hip", LEARN_HEADERS, [])]


def _deps(src, headers):
    return [os.path.join(HERE, f) for f in [src] + headers] + [os.path.join(ROOT, "include", f) for f in ("lsim.h", "lsim_layout.h")]


def _obj(src):
    return os.path.join(HERE, os.path.splitext(src)[0] + ".o")


def _newer(deps, target):
    return not os.path.exists(target) or any(os.path.getmtime(d) > os.path.getmtime(target) for d in deps)


def stale():
    return any(_newer(_deps(src, hdrs), LIB) for src, hdrs, _ in UNITS)


def build(force=False, verbose=False):
    """compile the translation units that changed (in parallel), link liblsim.so"""
    if not force and not stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "hipcc")
    extra = os.environ.get("LSIM_EXTRA_FLAGS", "").split()
    procs = []
    for src, hdrs, flags in UNITS:
        obj = _obj(src)
        if force or extra or _newer(_deps(src, hdrs), obj):
            cmd = [hipcc] + FLAGS + flags + extra + ["-c", os.path.join(HERE, src), "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((cmd, subprocess.Popen(cmd, cwd=HERE)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [_obj(src) for src, _, _ in UNITS] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=HERE)
    return LIB


def build_variant(out, extra_flags=(), workdir=None):
    """a diagnostics / from-source build of the whole library into `out`: both translation units compiled from source with `extra_flags`
    added (objects go to `workdir`, default next to `out`), never touching the in-tree objects or liblsim.so"""
    hipcc = os.environ.get("HIPCC", "hipcc")
    out = os.path.abspath(out)
    workdir = os.path.abspath(workdir) if workdir else os.path.dirname(out)
    os.makedirs(workdir, exist_ok=True)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    objs, procs = [], []
    for src, _, flags in UNITS:
        obj = os.path.join(workdir, os.path.basename(out) + "." + os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        cmd = [hipcc] + FLAGS + flags + list(extra_flags) + ["-c", os.path.join(HERE, src), "-o", obj]
        procs.append((cmd, subprocess.Popen(cmd, cwd=HERE)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out], cwd=HERE)
    for o in objs:
        os.remove(o)
    return out


def variant_is_stale(out):
    """a diagnostics build made from older sources than the tree's (round 5 shipped such a file to the GPU box: its tool failed on a missing symbol)"""
    return not os.path.exists(out) or any(os.path.getmtime(f) > os.path.getmtime(out) for f in all_sources() + [os.path.abspath(__file__)])


def all_sources():
    return sorted({os.path.join(HERE, f) for src, hdrs, _ in UNITS for f in [src] + hdrs})


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
