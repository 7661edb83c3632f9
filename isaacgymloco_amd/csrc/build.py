"""Build liblsim.so (the HIP library behind include/lsim.h) in-tree for gfx950.

    python -m isaacgymloco_amd.csrc.build [--force]

hipcc cross-compiles without a GPU; the .so travels to the GPU box with the repo snapshot."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(HERE, "liblsim.so")
SOURCES = ["lsim_hip.hip"]
HEADERS = ["ls_math.h", "ls_shared.h", "ls_physics.h", "ls_post.h", "ls_kernels.h", "ls_api_impl.h", "ls_rollout.h", "ls_learn.h", "ls_policy.h"]
# The simulator kernels' time is their vector instruction count, so the flags are chosen for that:
# -fno-hip-fp32-correctly-rounded-divide-sqrt: ~150 divisions per sub-step cost ~12 instructions each when IEEE-rounded; quotients that
#   must be exact use ls_div_exact (ls_math.h)
# -fgpu-flush-denormals-to-zero: with fp32 denormals kept, every sqrtf / fast division carries a frexp / ldexp range-scaling sequence
#   (5-8 instructions instead of 1-2; 360 of kernel A's 7 400 static vector instructions); nothing in the path lives near 1e-38
# -fno-slp-vectorize: the SLP vectoriser pairs independent fp32 FMAs into v_pk_fma_f32 and pays two v_mov_b32 per pair to pack the
#   operands: more instructions than the scalar form it replaces (kernel A: 7 626 -> 7 401 static, 0.1245 -> 0.116 ms)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-pass-failed",
         "-fno-hip-fp32-correctly-rounded-divide-sqrt", "-fgpu-flush-denormals-to-zero", "-fno-slp-vectorize"]


def stale():
    if not os.path.exists(LIB):
        return True
    deps = [os.path.join(HERE, f) for f in SOURCES + HEADERS] + [os.path.join(ROOT, "include", f) for f in ("lsim.h", "lsim_layout.h")]
    return any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps)


def build(force=False, verbose=False):
    if not force and not stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "hipcc")
    cmd = [hipcc] + FLAGS + [os.path.join(HERE, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=HERE)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
