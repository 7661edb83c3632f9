"""LeggedRobot: drop-in for legged_gym.envs.base.legged_robot.LeggedRobot on top of the HIP simulator.

Same constructor signature, methods and attribute names as the reference class (LR:54-176, BaseTask BT:40-115, the
VecEnv contract rsl_rl/env/vec_env.py:36-60 and the extra attributes the runners / play.py read):

    env = LeggedRobot(cfg, sim_params, physics_engine, sim_device, headless)
    obs, priv = env.reset()
    obs, priv, rew, dones, extras, term_ids, term_priv_obs[, terminal_amp] = env.step(actions)

Every tensor attribute is a zero-copy torch view of device memory owned by the simulator arena -- the role
gymtorch.wrap_tensor plays in the reference (LR:930-944).  All simulation, reward, reset and observation work is done
by the HIP kernels behind include/lsim.h (isaacgymloco_amd/csrc); this file only binds pointers, launches steps on
the current torch stream and assembles the reference's return tuple.  There is no CPU fallback.
"""
import ctypes
import os

import numpy as np
import torch

from .. import abi, lib
from ..robots import aliengo


from ..robots.model import build_robot_model  # noqa: E402,F401  (lives in robots/model.py: importable without torch)
from . import lsim_config as LC
from .terrain import Terrain

_TORCH_DT = {abi.DT_F32: torch.float32, abi.DT_I64: torch.int64, abi.DT_U8: torch.uint8, abi.DT_I32: torch.int32, abi.DT_I16: torch.int16}


# measurement hook: LSIM_STEP_* bits or-ed into every step (LSIM_STEP_FLAGS=16: kernel A without the contact-count wave priorities; 8: kernels A + B on every step)
_EXTRA_STEP_FLAGS = int(os.environ.get("LSIM_STEP_FLAGS", "0"), 0)

class LeggedRobot:
    def __init__(self, cfg, sim_params=None, physics_engine=None, sim_device="cuda:0", headless=True,
                 *, seed=1, rank=0, using_amp=False, terrain=None, terrain_seed=None):
        self._L = self._load_library()
        self.cfg = cfg
        self.sim_params = sim_params
        self.physics_engine = physics_engine
        self.sim_device = sim_device
        self.device = sim_device
        self.headless = headless
        self.using_amp = bool(using_amp)
        self.viewer = None
        self.gym = None
        self.num_envs = int(cfg.env.num_envs)
        self.num_obs = cfg.env.num_observations
        self.num_privileged_obs = cfg.env.num_privileged_obs
        self.num_actions = cfg.env.num_actions
        self.num_one_step_obs = cfg.env.num_one_step_observations
        self.num_one_step_privileged_obs = cfg.env.num_one_step_privileged_obs
        self.history_length = int(self.num_obs / self.num_one_step_obs)
        self.num_dof = self.num_dofs = 12
        self.num_bodies = 17
        self.dof_names = list(aliengo.DOF_NAMES)      # FL, FR, RL, RR x (hip, thigh, calf): the same for every supported quadruped
        # _parse_cfg (LR:1252-1263)
        self.dt = cfg.control.decimation * cfg.sim.dt
        self.obs_scales = cfg.normalization.obs_scales
        self.max_episode_length_s = cfg.env.episode_length_s
        self.max_episode_length = np.ceil(self.max_episode_length_s / self.dt)
        self.reward_scales = {k: v * self.dt for k, v in cfg.rewards.scales.to_dict().items() if v != 0}
        self.command_ranges = {k: list(v) for k, v in cfg.commands.ranges.to_dict().items()}

        dev = torch.device(sim_device)
        self._dev_index = dev.index if dev.index is not None else (torch.cuda.current_device() if dev.type == "cuda" else 0)
        self.terrain = terrain if terrain is not None else Terrain(cfg.terrain, self.num_envs, seed=seed if terrain_seed is None else terrain_seed)
        self.model = build_robot_model(cfg.asset)
        self.lcfg = LC.make_lsim_config(cfg, num_envs=self.num_envs, terrain=self.terrain, model=self.model, seed=seed, rank=rank, using_amp=using_amp)

        nbytes = ctypes.c_size_t()
        lib.check(self._L.lsim_query_arena(ctypes.byref(self.lcfg), ctypes.byref(nbytes)), what="lsim_query_arena")
        self._arena = torch.zeros(nbytes.value, dtype=torch.uint8, device=dev)
        grid_p = orig_p = None
        if self.lcfg.mesh_type != 0:
            self._grid = np.ascontiguousarray(self.terrain.heightsamples, dtype=np.int16)
            self._orig = np.ascontiguousarray(self.terrain.env_origins, dtype=np.float32)
            grid_p, orig_p = self._grid.ctypes.data, self._orig.ctypes.data
        self._h = ctypes.c_void_p()
        self._sync()
        lib.check(self._L.lsim_create(ctypes.byref(self.lcfg), ctypes.byref(self.model), grid_p, orig_p,
                                      self._arena.data_ptr(), self._dev_index, ctypes.byref(self._h)), what="lsim_create")
        self._sync()
        self._bind_buffers()
        self.extras = {}
        self._disturbance = None
        self.reward_curriculum_coef = []
        if cfg.env.send_timeouts:
            self.extras["time_outs"] = self._extras_time_outs
        self.extras["nonfinite_envs"] = self.nonfinite_envs
        self.common_step_counter = 0
        self.init_done = True

    # ------------------------------------------------------------------ the simulator behind this object
    def _load_library(self):
        """the HIP library (include/lsim.h).  There is no CPU path: without a GPU this raises.  [tests/emu_env.py overrides this hook, _sync and
        _stream to put the CPU lane emulator of the kernel sources behind the same Python surface -- a test harness, not a product path]"""
        if not torch.cuda.is_available():
            raise lib.LsimError("LeggedRobot needs a ROCm GPU: the simulator is a HIP library with no CPU path")
        return lib.load()

    def _sync(self):
        torch.cuda.synchronize(self._arena.device)

    # ------------------------------------------------------------------ buffers
    def _bind_buffers(self):
        base = self._arena.data_ptr()
        self.buf = {}
        for name, bid in abi.BUFFER_IDS.items():
            ptr, shape, nd, dt = ctypes.c_void_p(), (ctypes.c_int64 * 4)(), ctypes.c_int(), ctypes.c_int()
            lib.check(self._L.lsim_get_buffer(self._h, bid, ctypes.byref(ptr), shape, ctypes.byref(nd), ctypes.byref(dt)), self._h, "lsim_get_buffer")
            shp = tuple(shape[i] for i in range(nd.value))
            tdt = _TORCH_DT[dt.value]
            n = int(np.prod(shp)) * torch.empty((), dtype=tdt).element_size()
            off = ptr.value - base
            self.buf[name] = self._arena[off:off + n].view(tdt).view(shp)
        b = self.buf
        N = self.num_envs
        # names of the reference (BT:70-79, LR:930-1032)
        self.obs_buf, self.privileged_obs_buf, self.rew_buf = b["obs"], b["priv_obs"], b["rew"]
        self.reset_buf = b["reset"].view(torch.bool)
        self.time_out_buf = b["time_out"].view(torch.bool)
        self._extras_time_outs = b["extras_time_outs"].view(torch.bool)
        self._episode_length_buf = b["episode_length"]
        self.root_states, self.dof_state = b["root_states"], b["dof_state"].view(N * 12, 2)
        self.dof_pos, self.dof_vel = b["dof_state"][..., 0], b["dof_state"][..., 1]
        self.base_quat = self.root_states[:, 3:7]
        self.rigid_body_states = b["rigid_body_states"].view(N * 17, 13)
        self.contact_forces = b["contact_forces"]
        self.torques, self.actions = b["torques"], b["actions"]
        self.last_actions, self.last_last_actions = b["last_actions"], b["last_last_actions"]
        self.last_dof_pos, self.last_dof_vel, self.last_torques, self.last_root_vel = b["last_dof_pos"], b["last_dof_vel"], b["last_torques"], b["last_root_vel"]
        self.commands = b["commands"]
        self.base_lin_vel, self.base_ang_vel, self.projected_gravity = b["base_lin_vel"], b["base_ang_vel"], b["projected_gravity"]
        self.feet_air_time = b["feet_air_time"]
        self.last_contacts, self.contact_filt = b["last_contacts"].view(torch.bool), b["contact_filt"].view(torch.bool)
        self.measured_heights = b["measured_heights"]
        self.terrain_levels, self.terrain_types, self.env_origins = b["terrain_levels"], b["terrain_types"], b["env_origins"]
        self.Kp_factors, self.Kd_factors = b["kp_factors"].view(N, 1), b["kd_factors"].view(N, 1)
        self.motor_strength, self.motor_strength_factors = b["motor_strength"], b["motor_strength_factors"].view(N, 1)
        self.friction_coeffs, self.restitution_coeffs = b["friction"].view(N, 1), b["restitution"].view(N, 1)
        self.payload, self.com_displacement = b["payload"].view(N, 1), b["com_displacement"]
        self.episode_sums = {name: b["episode_sums"][:, abi.REWARD_IDS[name]] for name in self.reward_scales}
        self.termination_privileged_obs_buf, self.terminal_amp_states_buf, self.amp_obs_buf = b["term_priv_obs"], b["term_amp_obs"], b["amp_obs"]
        # running count of env-steps whose simulated state held a NaN / infinity (LSIM_BUF_NONFINITE: cumulative since creation, never cleared by the
        # library) -- a live 0-d device tensor, so carrying it in `extras` costs no host sync; anything but 0 means the solver blew up
        self.nonfinite_envs = b["nonfinite"][0]
        dev = self._arena.device
        self.feet_indices = torch.tensor(list(self.model.feet_bodies), dtype=torch.long, device=dev)
        self.penalised_contact_indices = torch.tensor([i for i in range(17) if (self.model.penalised_body_mask >> i) & 1], dtype=torch.long, device=dev)
        self.termination_contact_indices = torch.tensor([i for i in range(17) if (self.model.termination_body_mask >> i) & 1], dtype=torch.long, device=dev)
        self.default_dof_pos = torch.tensor([self.lcfg.default_dof_pos[i] for i in range(12)], device=dev).unsqueeze(0)
        self.p_gains = torch.tensor([self.lcfg.p_gains[i] for i in range(12)], device=dev)
        self.d_gains = torch.tensor([self.lcfg.d_gains[i] for i in range(12)], device=dev)
        self.torque_limits = torch.tensor([self.model.dof_effort_limit[i] for i in range(12)], device=dev)
        self.dof_vel_limits = torch.tensor([self.model.dof_vel_limit[i] for i in range(12)], device=dev)
        lim = torch.tensor([[self.model.dof_pos_lower[i], self.model.dof_pos_upper[i]] for i in range(12)], device=dev)
        m, r = (lim[:, 0] + lim[:, 1]) / 2, lim[:, 1] - lim[:, 0]          # soft limits, LR:574-578
        s = self.cfg.rewards.soft_dof_pos_limit
        self.dof_pos_limits = torch.stack((m - 0.5 * r * s, m + 0.5 * r * s), dim=1)

    @property
    def episode_length_buf(self):
        return self._episode_length_buf

    @episode_length_buf.setter
    def episode_length_buf(self, value):      # HIMR:90-91 rebinds the attribute; keep the simulator's buffer
        self._episode_length_buf.copy_(value.to(self._episode_length_buf.dtype))

    @property
    def feet_pos(self):                        # LR:203 (re-gathered every step in the reference)
        return self.buf["rigid_body_states"][:, self.feet_indices, 0:3]

    @property
    def feet_vel(self):                        # LR:204
        return self.buf["rigid_body_states"][:, self.feet_indices, 7:10]

    @property
    def height_samples(self):                  # LR:1084 / LR:1098: the int16 height grid on the device
        return self.buf["height_grid"]

    def _get_feet_heights(self, env_ids=None):
        """Height of every foot above the terrain under it (LR:1400-1441): feet xy + border, / horizontal_scale, truncated, clipped to the grid,
        MEAN of the three samples (x, y), (x+1, y), (x, y+1) -- the other samplers take their minimum -- times vertical_scale; `plane`
        terrain returns the feet z.  No reward or observation of the reference calls it; it is part of the task's surface, so it exists,
        as device-side torch ops on the simulator's buffers."""
        mesh = self.cfg.terrain.mesh_type
        if mesh == "plane":
            return self.feet_pos[:, :, 2].clone()
        if mesh == "none":
            raise NameError("Can't measure height with terrain mesh type 'none'")
        feet = self.feet_pos if env_ids is None else self.feet_pos[env_ids]
        points = ((feet + self.cfg.terrain.border_size) / self.cfg.terrain.horizontal_scale).long()
        grid = self.height_samples
        px = torch.clip(points[:, :, 0].reshape(-1), 0, grid.shape[0] - 2)
        py = torch.clip(points[:, :, 1].reshape(-1), 0, grid.shape[1] - 2)
        heights = (grid[px, py] + grid[px + 1, py] + grid[px, py + 1]) / 3         # int16 sums promote to float on the division, as in the reference
        heights = heights.view(feet.shape[0], -1) * self.cfg.terrain.vertical_scale
        return feet[:, :, 2] - heights

    @property
    def disturbance(self):
        """self.disturbance (LR:1017), (N, 17, 3): the body-local force applied to every body in the next simulate().  Only row 0 (the base)
        is ever written (LR:843) and the reference zeroes the whole buffer at the end of each step (LR:235), so what a caller can observe
        between steps is zeros; the force drawn at LR:842 lives on in `pending_force` until the first sub-step consumes it."""
        if self._disturbance is None:
            self._disturbance = torch.zeros(self.num_envs, self.num_bodies, 3, device=self._arena.device)
        return self._disturbance

    @property
    def pending_force(self):                   # (N,3) base force of the next step, body-local frame (what LR:844 hands to the simulator)
        return self.buf["pending_force"]

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self._arena.device).cuda_stream)

    # ------------------------------------------------------------------ VecEnv surface
    def get_observations(self):
        return self.obs_buf

    def get_privileged_observations(self):
        return self.privileged_obs_buf

    def get_amp_observations(self):
        return self.amp_obs_buf

    def _external_call(self):
        """a public entry point (step / reset / reset_idx) is about to overwrite the live buffers: let a device-side rollout that still holds
        references to them (learn/graph_rollout.py: the deferred post-step store) finish first.  step_device() is the rollout's own entry."""
        hook = getattr(self, "before_external_step", None)
        if hook is not None:
            hook()

    def reset_idx(self, env_ids):
        """LR:290-361 called from outside a step: BaseTask.reset's reset_idx(all envs) (BT:113), or a subset chosen by the caller (a play
        script resetting some robots by hand).  The reset_idx that step() itself makes (LR:229) lives inside kernel B.  Like the
        reference: an empty id list returns at once (LR:298); observations are not recomputed here."""
        env_ids = torch.as_tensor(env_ids, device=self._arena.device, dtype=torch.long).flatten()
        if env_ids.numel() == 0:
            return
        self._external_call()
        mask = torch.zeros(self.num_envs, dtype=torch.uint8, device=self._arena.device)
        mask[env_ids] = 1          # an id out of range raises here, as the reference's indexing would
        if env_ids.numel() == self.num_envs and bool(mask.all()):
            lib.check(self._L.lsim_reset_all(self._h, self._stream()), self._h, "lsim_reset_all")
        else:
            lib.check(self._L.lsim_reset_envs(self._h, mask.data_ptr(), self._stream()), self._h, "lsim_reset_envs")
            self._reset_mask_ref = mask   # keep alive until the kernels ran
        self._refresh_extras(force_valid=True)

    def reset(self):
        """BaseTask.reset (BT:111-115): reset_idx(all) then one zero-action step."""
        self.reset_idx(torch.arange(self.num_envs, device=self._arena.device))
        obs, priv, *_ = self.step(torch.zeros(self.num_envs, self.num_actions, device=self._arena.device))
        return obs, priv

    def step_device(self, actions, flags=0):
        """Enqueue one LeggedRobot.step() on the current stream without any host synchronisation.
        Fixed-capacity outputs: reset_buf is the mask of terminated envs, termination_privileged_obs_buf /
        terminal_amp_states_buf rows are valid where the mask is set.  The returned tensors are the LIVE simulator
        buffers (overwritten by the next step): clone what must survive."""
        if actions.dtype != torch.float32 or not actions.is_contiguous() or actions.device != self._arena.device:
            actions = actions.to(device=self._arena.device, dtype=torch.float32).contiguous()
        with lib.roctx_range("lsim_step"):
            lib.check(self._L.lsim_step_ex(self._h, actions.data_ptr(), ctypes.c_uint32(flags | _EXTRA_STEP_FLAGS), self._stream()), self._h, "lsim_step")
        self.common_step_counter += 1
        self._last_actions_ref = actions   # keep alive until the kernels ran
        return self.obs_buf, self.privileged_obs_buf, self.rew_buf, self.reset_buf

    def step(self, actions):
        """LeggedRobot.step (LR:122-176): same 7-tuple (8 with terminal AMP states when using_amp)."""
        self._external_call()
        self.step_device(actions)
        env_ids = self.reset_buf.nonzero(as_tuple=False).flatten()   # the reference's own host sync (LR:225)
        self._refresh_extras(force_valid=len(env_ids) > 0)
        # the reference rebinds obs_buf / privileged_obs_buf to fresh tensors every step (LR:168-171, LR:403-404) and its
        # runner keeps the returned objects across the next step (HIMP:100-101): hand out copies, not the live buffers
        out = (self.obs_buf.clone(), self.privileged_obs_buf.clone(), self.rew_buf.clone(), self.reset_buf.clone(), self.extras, env_ids,
               self.termination_privileged_obs_buf[env_ids])
        if self.using_amp:
            out = out + (self.terminal_amp_states_buf[env_ids],)
        return out

    # ---- resumable simulator-side training state (the reference's checkpoints forget it: HIMR:233-240 save only the networks)
    def state_dict(self):
        """curricula and counters a resumed run needs: terrain levels/types and origins, the live command ranges (command curriculum,
        LR:868-880), common_step_counter (push / disturbance phase, Philox step word), the number of by-hand reset_idx calls
        (salt of their draws), episode lengths"""
        from .. import abi
        c = ctypes.c_int64()
        self._L.lsim_get_step_counter(self._h, ctypes.byref(c))
        rc = ctypes.c_uint32()
        self._L.lsim_get_reset_calls(self._h, ctypes.byref(rc))
        S = abi.STATS["cmd_ranges"]
        return {"step_counter": int(c.value), "reset_calls": int(rc.value), "terrain_levels": self.terrain_levels.cpu().clone(), "terrain_types": self.terrain_types.cpu().clone(),
                "env_origins": self.env_origins.cpu().clone(), "episode_length_buf": self.episode_length_buf.cpu().clone(),
                "command_ranges": self.stats_row()[S:S + 8].cpu().clone(),
                # the simulator conventions the run was made under: observations / rewards of a policy depend on them
                "conventions": self._conventions()}

    def _conventions(self):
        from .. import abi
        c = self.lcfg
        return {"abi_version": int(abi.ABI_VERSION), "lin_vel_at_com": int(c.lin_vel_at_com), "tgs_limit_passes": int(c.tgs_limit_passes),
                "solver_type": int(c.solver_type), "num_position_iterations": int(c.num_position_iterations)}

    def load_state_dict(self, d):
        from .. import abi
        import warnings
        conv, live = d.get("conventions"), self._conventions()
        if conv is None:
            warnings.warn("simulator state saved before round 6: it does not record the conventions it was trained under (centre-of-mass vs link-origin "
                          f"linear velocities, TGS limit passes); this simulator runs {live} -- LSIM_LIN_VEL=origin LSIM_TGS_LIMIT_PASSES=0 restore rounds 1-4")
        else:
            diff = {k: (conv[k], live[k]) for k in ("lin_vel_at_com", "tgs_limit_passes", "solver_type", "num_position_iterations") if conv.get(k) != live[k]}
            if diff:
                warnings.warn(f"simulator state was saved under other conventions (saved, live): {diff}; base_lin_vel observations / contact behaviour differ")
        self._L.lsim_set_step_counter(self._h, ctypes.c_int64(int(d["step_counter"])))
        self._L.lsim_set_reset_calls(self._h, ctypes.c_uint32(int(d.get("reset_calls", 0))))       # (checkpoints from before round 5 do not hold it)
        self.terrain_levels.copy_(d["terrain_levels"]); self.terrain_types.copy_(d["terrain_types"])
        self.env_origins.copy_(d["env_origins"]); self.episode_length_buf.copy_(d["episode_length_buf"])
        S = abi.STATS["cmd_ranges"]
        self.buf["stats"][:, S:S + 8] = d["command_ranges"].to(self.buf["stats"].device)      # both ping-pong rows
        self._sync()

    def stats_row(self):
        row = ctypes.c_int()
        self._L.lsim_get_stats_row(self._h, ctypes.byref(row))
        return self.buf["stats"][row.value]

    def _refresh_extras(self, force_valid=False):
        """extras["episode"] / extras["time_outs"] (LR:346-359).  Like the reference, the dict is only replaced on steps
        with at least one reset; values are device tensors (no host sync here)."""
        if self.cfg.env.send_timeouts:
            self.extras["time_outs"] = self._extras_time_outs
        self.extras["nonfinite_envs"] = self.nonfinite_envs
        if not force_valid:
            return
        S = abi.STATS
        st = self.stats_row().clone()
        n = torch.clamp(st[S["reset_count"]], min=1.0)
        ep = {}
        for name in self.reward_scales:
            ep["rew_" + name] = st[S["episode_sums"] + abi.REWARD_IDS[name]] / n / self.dt
        if self.cfg.terrain.curriculum:
            ep["terrain_level"] = torch.mean(self.terrain_levels.float())
        if self.cfg.commands.curriculum:
            ep["max_command_x"] = st[S["cmd_ranges"] + 1]
        self.extras["episode"] = ep

    def update_reward_curriculum(self, current_iter):
        """LR:830-836: linear schedule (iter0, iter1, coef0, coef1) per entry -> reward_curriculum_coef.  As in the reference the coefficients are
        only stored: no reward term reads them (cfg.rewards.reward_curriculum is False in every shipped config and HYBR:162 is the only
        caller).  The reference's default schedule is a flat list (LRC:181), which its own loop cannot index; a flat list counts as one entry."""
        sched = getattr(self.cfg.rewards, "reward_curriculum_schedule", None) or []
        if sched and not isinstance(sched[0], (list, tuple)):
            sched = [sched]
        coefs = []
        for it0, it1, c0, c1 in sched:
            p = max(min((current_iter - it0) / (it1 - it0), 1), 0)
            coefs.append((1 - p) * c0 + p * c1)
        self.reward_curriculum_coef = coefs

    def set_camera(self, position, lookat):
        """LR:1052-1057 points the viewer's camera; this build is headless (viewer = None, BT:81-98 never creates one): accepted and ignored,
        so that play-style scripts (play.py:117, play.py:143) run unchanged"""
        return None

    def render(self, sync_frame_time=True):
        """BT:117-148: viewer events and drawing; nothing to do without a viewer"""
        return None

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._sync()
            self._L.lsim_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
