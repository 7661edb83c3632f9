"""ctypes binding of liblsim.so (include/lsim.h).  There is NO fallback: if the HIP library is missing or was built
against another header the import of the simulator fails loudly."""
import ctypes
import os

from . import abi

# LSIM_LIB selects another build of the same HIP library (A/B experiments on kernel variants); never a fallback
LIB_PATH = os.environ.get("LSIM_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "liblsim.so")
_lib = None


class LsimError(RuntimeError):
    pass


def load():
    global _lib
    if _lib is None:
        _lib = load_path(LIB_PATH)
    return _lib


def load_path(path):
    """a build of the HIP library at `path`, prototypes set (load() = the product build, cached; tests load diagnostics variants beside it)"""
    if not os.path.exists(path):
        raise LsimError(f"{path} not found: build the HIP extension first (python -m isaacgymloco_amd.csrc.build); "
                        "there is no CPU fallback for the simulator")
    L = ctypes.CDLL(path)
    abi.check_abi(L, prefix="lsim")
    L.lsim_abi_version.restype = ctypes.c_int
    if L.lsim_abi_version() != abi.ABI_VERSION:
        raise LsimError("liblsim.so ABI version mismatch; rebuild")
    vp, i32, u32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_int64
    L.lsim_query_arena.argtypes = [ctypes.POINTER(abi.LsimConfig), ctypes.POINTER(ctypes.c_size_t)]
    L.lsim_create.argtypes = [ctypes.POINTER(abi.LsimConfig), ctypes.POINTER(abi.LsimRobotModel), vp, vp, vp, i32, ctypes.POINTER(vp)]
    L.lsim_get_buffer.argtypes = [vp, i32, ctypes.POINTER(vp), ctypes.POINTER(i64), ctypes.POINTER(i32), ctypes.POINTER(i32)]
    L.lsim_reset_all.argtypes = [vp, vp]
    L.lsim_reset_envs.argtypes = [vp, vp, vp]
    L.lsim_step.argtypes = [vp, vp, vp]
    L.lsim_step_ex.argtypes = [vp, vp, u32, vp]
    L.lsim_get_step_counter.argtypes = [vp, ctypes.POINTER(i64)]
    L.lsim_set_step_counter.argtypes = [vp, i64]
    L.lsim_get_reset_calls.argtypes = [vp, ctypes.POINTER(u32)]
    L.lsim_set_reset_calls.argtypes = [vp, u32]
    L.lsim_get_stats_row.argtypes = [vp, ctypes.POINTER(i32)]
    L.lsim_last_error.argtypes = [vp]
    L.lsim_last_error.restype = ctypes.c_char_p
    L.lsim_reward_name.restype = ctypes.c_char_p
    L.lsim_buffer_name.restype = ctypes.c_char_p
    f32 = ctypes.c_float
    L.lsim_rollout_act.argtypes = [ctypes.POINTER(abi.LsimRolloutStorage), vp, vp, vp, vp, vp, vp, vp, u32, u32, vp, vp]
    L.lsim_rollout_post.argtypes = [ctypes.POINTER(abi.LsimRolloutStorage), vp, vp, vp, vp, vp, vp, vp, vp, f32, vp]
    L.lsim_rollout_act_at.argtypes = [ctypes.POINTER(abi.LsimRolloutStorage), ctypes.c_int64, ctypes.c_int64, vp, vp, vp, vp, vp, u32, u32, vp, vp]
    L.lsim_rollout_post_at.argtypes = [ctypes.POINTER(abi.LsimRolloutStorage), ctypes.c_int64, vp, vp, vp, vp, vp, vp, f32, vp]
    L.lsim_policy_act_at.argtypes = [vp, ctypes.POINTER(abi.LsimRolloutStorage), ctypes.c_int64, ctypes.c_int64, vp, vp, vp, u32, u32, vp, vp, vp, vp]
    L.lsim_policy_act_post_at.argtypes = [vp, ctypes.POINTER(abi.LsimRolloutStorage), ctypes.c_int64, ctypes.c_int64, vp, vp, vp, u32, u32, vp, vp, vp,
                                          ctypes.c_int64, vp, vp, vp, vp, f32, vp]
    L.lsim_rollout_gae.argtypes = [ctypes.POINTER(abi.LsimRolloutStorage), vp, f32, f32, vp, vp, vp]
    L.lsim_linear_wgrad_workspace.argtypes = [ctypes.c_long, i32, i32, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(i32)]
    L.lsim_linear_wgrad.argtypes = [vp, i64, vp, i64, i64, i32, i32, vp, vp, vp, ctypes.c_size_t, vp]
    L.lsim_wgrad_split_bf16.argtypes = [i32]
    L.lsim_sinkhorn_workspace.argtypes = [ctypes.c_long, i32, ctypes.POINTER(ctypes.c_size_t)]
    L.lsim_sinkhorn.argtypes = [vp, i64, i64, i32, f32, i32, vp, vp, ctypes.c_size_t, vp]
    L.lsim_policy_forward.argtypes = [ctypes.POINTER(abi.LsimHimPolicy), vp, vp, i64, vp, vp, vp]
    L.lsim_estimator_loss_workspace.argtypes = [i64, i32, i32, ctypes.POINTER(ctypes.c_size_t)]
    L.lsim_estimator_loss.argtypes = [vp, i64, vp, i64, vp, vp, i64, i64, i32, i32, f32, f32, i32, vp, vp, vp, vp, vp, ctypes.c_size_t, vp]
    L.lsim_adam_clip_step_workspace.argtypes = [i32, ctypes.POINTER(ctypes.c_size_t)]
    pp = ctypes.POINTER(ctypes.c_void_p)
    L.lsim_adam_clip_step.argtypes = [i32, ctypes.POINTER(i64), pp, pp, pp, pp, pp, vp, f32, f32, f32, f32, f32, vp, vp, ctypes.c_size_t, vp]
    L.lsim_adam_clip_step_ex.argtypes = [i32, ctypes.POINTER(i64), pp, pp, pp, pp, pp, ctypes.POINTER(ctypes.c_float), i32, vp, f32, f32, f32, f32, f32,
                                         vp, vp, ctypes.c_size_t, vp]
    L.lsim_actor_input.argtypes = [vp, i64, i32, vp, i64, i32, i64, vp, vp]
    L.lsim_ppo_loss_workspace.argtypes = [ctypes.c_long, ctypes.POINTER(ctypes.c_size_t)]
    L.lsim_ppo_loss.argtypes = [vp] * 10 + [i64, i32, f32, f32, f32, i32, vp, vp, vp, vp, vp, ctypes.c_size_t, vp]
    L.lsim_ppo_loss_std_workspace.argtypes = [ctypes.c_long, i32, ctypes.POINTER(ctypes.c_size_t)]
    L.lsim_ppo_loss_std.argtypes = [vp] * 10 + [i64, i32, f32, f32, f32, i32, vp, vp, vp, vp, vp, ctypes.c_size_t, vp]
    L.lsim_adaptive_lr.argtypes = [vp, f32, f32, f32, f32, vp, vp]
    L.lsim_linear_elu_wgrad.argtypes = [vp, i64, vp, i64, vp, i64, i64, i32, i32, vp, vp, vp, vp, ctypes.c_size_t, vp]
    L.lsim_linear_elu_forward.argtypes = [vp, i64, vp, vp, i64, i32, i32, vp, i64, vp]
    pend = ctypes.POINTER(abi.LsimWgradPending)
    L.lsim_linear_wgrad_deferred.argtypes = [vp, i64, vp, i64, i64, i32, i32, vp, vp, vp, ctypes.c_size_t, vp, pend]
    L.lsim_linear_elu_wgrad_deferred.argtypes = [vp, i64, vp, i64, vp, i64, i64, i32, i32, vp, vp, vp, vp, ctypes.c_size_t, vp, pend]
    L.lsim_wgrad_reduce_batch.argtypes = [pend, i32, vp]
    L.lsim_normalize_rows.argtypes = [vp, i32, i32, f32, vp]
    L.lsim_gather_rows.argtypes = [vp, i64, vp, i64, vp, vp]
    L.lsim_gather_rows_ld.argtypes = [vp, i64, vp, i64, vp, i64, vp]
    L.lsim_linear_relu_wgrad.argtypes = [vp, i64, vp, i64, vp, i64, i64, i32, i32, vp, vp, vp, vp, ctypes.c_size_t, vp]
    L.lsim_linear_relu_wgrad_deferred.argtypes = [vp, i64, vp, i64, vp, i64, i64, i32, i32, vp, vp, vp, vp, ctypes.c_size_t, vp, pend]
    L.lsim_linear_masked_forward.argtypes = [vp, i64, vp, vp, i64, i64, i32, i32, vp, i64, vp]
    L.lsim_relu_cols_workspace.argtypes = [i64, i32, ctypes.POINTER(ctypes.c_size_t)]
    L.lsim_relu_head_backward.argtypes = [vp, i64, vp, vp, i64, i32, vp, vp, vp, vp, ctypes.c_size_t, vp]
    L.lsim_masked_colsum.argtypes = [vp, i64, vp, i64, i64, i32, vp, vp, ctypes.c_size_t, vp]
    L.lsim_running_moments_workspace.argtypes = [ctypes.POINTER(ctypes.c_size_t)]
    L.lsim_running_moments_update.argtypes = [vp, i64, i64, i32, vp, vp, vp, vp, ctypes.c_size_t, vp]
    L.lsim_amp_pair_rows.argtypes = [vp, i64, vp, i64, vp, vp, ctypes.c_double, ctypes.c_double, i64, i32, vp, i64, vp]
    L.lsim_amp_step_workspace.argtypes = [i64, ctypes.POINTER(ctypes.c_size_t)]
    L.lsim_amp_step.argtypes = [ctypes.POINTER(abi.LsimAmpDisc), vp, vp, vp, vp, vp, i64, vp, vp, vp, vp, vp, i64, i64, vp, ctypes.c_size_t, vp]
    L.lsim_destroy.argtypes = [vp]
    L.lsim_destroy.restype = None
    return L


# ---- roctx ranges (SURVEY.md 8d "roctx ranges per K1-K4"): LSIM_ROCTX=1 brackets the simulator step, the policy / storage kernels of the
# rollout and the learner update with named ranges that `rocprofv3 --marker-trace` shows on the timeline.  Off by default: a push / pop
# pair is two library calls per range on the host-bound rollout loop.
_roctx = None


def _roctx_lib():
    global _roctx
    if _roctx is None:
        _roctx = False
        if os.environ.get("LSIM_ROCTX") == "1":
            for name in ("librocprofiler-sdk-roctx.so", "libroctx64.so"):
                try:
                    R = ctypes.CDLL(name)
                    R.roctxRangePushA.argtypes = [ctypes.c_char_p]
                    _roctx = R
                    break
                except OSError:
                    continue
    return _roctx


class roctx_range:
    """with roctx_range("lsim_step"): ...   (no-op unless LSIM_ROCTX=1 and a roctx library is present)"""
    __slots__ = ("name", "on")

    def __init__(self, name):
        self.name = name.encode()
        self.on = False

    def __enter__(self):
        R = _roctx_lib()
        if R:
            R.roctxRangePushA(self.name)
            self.on = True
        return self

    def __exit__(self, *exc):
        if self.on:
            _roctx.roctxRangePop()
        return False


def check(rc, handle=None, what="lsim call"):
    if rc != 0:
        msg = ""
        if handle is not None and _lib is not None:
            m = _lib.lsim_last_error(handle)
            msg = m.decode() if m else ""
        raise LsimError(f"{what} failed with code {rc} {msg}")
