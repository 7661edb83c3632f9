"""Parity tests proper: the HIP library, called through the C-ABI, against (1) the golden vectors captured from the
reference's LeggedRobot.step() and (2) the CPU oracle for the dynamics.  Need a real MI355X (-m gpu)."""
import numpy as np
import pytest

import golden_replay as GR
from helpers import C, T, make_oracle, quiet_cfg, abi

pytestmark = pytest.mark.gpu


FORMS = {"fused": 0, "two_kernels": abi.STEP_TWO_KERNELS}      # kernel A with the fused tail + lsim_k_step_finish (default) / kernels A + B on every step


@pytest.mark.parametrize("form", list(FORMS))
@pytest.mark.parametrize("name", GR.SCENARIOS)
def test_hip_matches_reference_step(name, form):
    from hip_backend import HipBackend
    fx = GR.load(name)
    cfg = GR.scenario_cfg(name)
    be = HipBackend(cfg, int(fx["num_envs"]), GR.FixtureTerrain(fx), seed=int(fx["seed"]))
    n = 0
    for t, ref in GR.replay(fx, be, be.get, be.put, extra_flags=FORMS[form]):
        GR.compare_step(t, ref, be.get, be.stats_row)
        n += 1
    assert n == fx["in_actions"].shape[0]
    if "fin_ids" in fx.files:     # the reference's reset_idx(env_ids) called by hand after the last step (LR:290): lsim_reset_envs
        mask = GR.replay_final_reset(fx, be, be.get, be.put)
        GR.compare_final_reset(fx, mask, be.get, be.stats_row)


@pytest.mark.parametrize("name", GR.BIG_SCENARIOS)
def test_hip_matches_reference_step_at_baseline_size(name):
    """BASELINE size, N = 4096 (cfg 2-4) through the C-ABI: the reference's own outputs on the steps around the command-curriculum step
    999 -> 1000 -- every env for rewards / commands / resets / terrain placement, 211+ selected envs (every 37th + bands around each index
    boundary of LR:72-90, LR:649, LR:1234) for the observation rows, fp64 column sums over all envs for the rest."""
    from hip_backend import HipBackend
    fx = GR.load_big(name)
    cfg = GR.big_scenario_cfg(name)
    be = HipBackend(cfg, int(fx["num_envs"]), GR.FixtureTerrain(fx), seed=int(fx["seed"]))
    n = 0
    for t, ref in GR.replay(fx, be, be.get, be.put):
        GR.compare_step(t, ref, be.get, be.stats_row)
        n += 1
    assert n == 3


SOLVERS = {"tgs": 1, "pgs": 0}      # cfg.sim.physx.solver_type (LRC:245): every reference config sets 1; 0 = the build's velocity-level sweeps


@pytest.mark.parametrize("solver", ["tgs", "pgs"])
@pytest.mark.parametrize("quiet", [True, False])
def test_hip_physics_matches_oracle(quiet, solver):
    """Dynamics: structured fp32 wave solver (HIP) vs dense fp64 oracle on the same seeds/actions, for both solvers (TGS with 4 position
    iterations -- kernel lsim_k_step_a_tgs against the oracle's TGS branch -- and the 8 velocity-level sweeps).  fp32 tolerance:
    states agree to 3e-4 abs (joints 3e-3) over 12 steps (48 sub-steps) of contact-rich motion; contact forces to 0.15 N (< 0.1 %)."""
    from hip_backend import HipBackend
    N = 16
    cfg = quiet_cfg("aliengo") if quiet else C.TASKS["aliengo"][0]()
    cfg.sim.physx.solver_type = SOLVERS[solver]
    cfg.terrain.terrain_proportions = [1.0, 0.0, 0.0, 0.0]
    orc, lc, model, ter = make_oracle(cfg, N, seed=5)
    assert lc.solver_type == SOLVERS[solver] and lc.num_position_iterations == 4
    be = HipBackend(cfg, N, ter, seed=5)
    orc.reset_all(); be.reset_all()
    rs = np.random.RandomState(0)
    for t in range(12):
        a = rs.normal(0, 1, (N, 12)).astype(np.float32)
        orc.step(a); be.step(a)
        np.testing.assert_array_equal(be.get("reset"), orc.buf["reset"], err_msg=f"step {t}")
        # bars ~ 5 x the errors measured on MI355X (round 2: root 4.6e-5, joints 7.3e-4, contact forces 0.032 N, reward 6e-7, observations 3.6e-5)
        np.testing.assert_allclose(be.get("root_states"), orc.buf["root_states"], atol=3e-4, rtol=1e-4, err_msg=f"step {t}")
        np.testing.assert_allclose(be.get("dof_state"), orc.buf["dof_state"], atol=3e-3, rtol=1e-3, err_msg=f"step {t}")
        np.testing.assert_allclose(be.get("contact_forces"), orc.buf["contact_forces"], atol=0.15, rtol=2e-3, err_msg=f"step {t}")
        np.testing.assert_allclose(be.get("rew"), orc.buf["rew"], atol=1e-5, rtol=1e-4, err_msg=f"step {t}")
        np.testing.assert_allclose(be.get("obs"), orc.buf["obs"], atol=3e-4, rtol=1e-4, err_msg=f"step {t}")


@pytest.mark.parametrize("solver", ["tgs", "pgs"])
def test_hip_stairs_wall_contacts_match_oracle(solver):
    """Stairs (slope-corrected mesh with vertical risers, TER:72-75): robots dropped all over the staircases, both solvers.  States are
    re-synchronised every step; float-vs-double decisions at triangle edges may differ for single env-steps, so the bar is
    >= 99 % of env-steps with identical termination flag AND state within the fp32 tolerance (measured: 1276 of 1280)."""
    from hip_backend import HipBackend
    N = 32
    cfg = C.TASKS["aliengo_stairs"][0]()
    cfg.sim.physx.solver_type = SOLVERS[solver]
    cfg.terrain.terrain_proportions = [0, 0, 0, 0, 0.5, 0.5]
    cfg.domain_rand.base_init_pos_range = dict(x=[-3.0, 3.0], y=[-3.0, 3.0], z=[0.0, 0.3])
    orc, lc, model, ter = make_oracle(cfg, N, seed=3)
    be = HipBackend(cfg, N, ter, seed=3)
    mesh = be.get("terrain_mesh")
    np.testing.assert_array_equal((mesh >> 16) & 0xFF, _flags_numpy(ter.heightsamples, cfg))
    np.testing.assert_array_equal((mesh.view(np.uint32) >> 24).astype(np.int64), _dzmax_numpy(ter.heightsamples))
    np.testing.assert_array_equal((mesh & 0xFFFF).astype(np.uint16).view(np.int16), ter.heightsamples)
    orc.reset_all(); be.reset_all()
    rs = np.random.RandomState(0)
    ok = tot = 0
    lateral = 0.0
    for t in range(40):
        a = rs.normal(0, 1, (N, 12)).astype(np.float32)
        for k in ("root_states", "dof_state", "commands", "last_actions", "episode_length", "terrain_levels", "env_origins", "kp_factors",
                  "kd_factors", "friction", "pending_force", "feet_air_time", "last_contacts", "episode_sums", "obs", "last_dof_vel"):
            be.put(k, orc.buf[k])
        orc.step(a); be.step(a)
        same = be.get("reset") == orc.buf["reset"]          # a 1 N termination threshold on a chaotic contact force can flip
        e_root = np.abs(be.get("root_states") - orc.buf["root_states"]).max(1)
        e_dof = np.abs(be.get("dof_state") - orc.buf["dof_state"]).reshape(N, -1).max(1)
        ok += int((same & (e_root < 2e-3) & (e_dof < 2e-2)).sum()); tot += N
        feet = be.get("contact_forces")[:, [4, 8, 12, 16], :]
        lat = np.linalg.norm(feet[..., :2], axis=-1)
        if (lat > 5).any():
            lateral = max(lateral, float((lat / (np.abs(feet[..., 2]) + 1e-6))[lat > 5].max()))
    print(f"stairs: {ok} of {tot} env-steps within tolerance")
    assert ok >= 0.99 * tot, (ok, tot)
    assert lateral > 5.0, "risers must be able to produce mostly-horizontal foot forces (feet_stumble, LR:1589-1599)"


def _dzmax_numpy(hf):
    """bits 24-31 of the mesh words: highest vertex of the 4 x 4 block (i - 1 .. i + 2) x (j - 1 .. j + 2) above vertex (i, j), in units of 4
    height steps rounded up, capped at 255 (third restatement of ls_api_impl.h: ls_terrain_mesh_dzmax)"""
    h = hf.astype(np.int64)
    R, C = h.shape
    pad = np.full((R + 3, C + 3), np.iinfo(np.int64).min)
    pad[1:R + 1, 1:C + 1] = h
    m = np.full((R, C), np.iinfo(np.int64).min)
    for a in range(4):
        for b in range(4):
            m = np.maximum(m, pad[a:a + R, b:b + C])
    return np.minimum((m - h + 3) // 4, 255)


def _flags_numpy(hf, cfg):
    """numpy restatement of the vertex displacement rule (third implementation, for the flag buffer)"""
    hf = hf.astype(np.float64)
    R, Cc = hf.shape
    thr = cfg.terrain.slope_treshold * cfg.terrain.horizontal_scale / cfg.terrain.vertical_scale
    mx = np.zeros((R, Cc)); my = np.zeros((R, Cc)); mc = np.zeros((R, Cc))
    mx[:R - 1, :] += hf[1:, :] - hf[:R - 1, :] > thr
    mx[1:, :] -= hf[:R - 1, :] - hf[1:, :] > thr
    my[:, :Cc - 1] += hf[:, 1:] - hf[:, :Cc - 1] > thr
    my[:, 1:] -= hf[:, :Cc - 1] - hf[:, 1:] > thr
    mc[:R - 1, :Cc - 1] += hf[1:, 1:] - hf[:R - 1, :Cc - 1] > thr
    mc[1:, 1:] -= hf[:R - 1, :Cc - 1] - hf[1:, 1:] > thr
    dx = (mx + mc * (mx == 0)).astype(np.int64); dy = (my + mc * (my == 0)).astype(np.int64)
    moved = np.pad((dx != 0) | (dy != 0), ((1, 2), (1, 2)))
    anyw = np.zeros((R, Cc), bool)
    for a in range(4):
        for b in range(4):
            anyw |= moved[a:a + R, b:b + Cc]
    return ((dx + 1) | ((dy + 1) << 2) | (anyw.astype(np.int64) << 4)).astype(np.uint8)


@pytest.mark.parametrize("solver", ["tgs", "pgs"])
def test_hip_full_size_invariants(solver):
    """BASELINE size (N=4096), both solvers: size-independent properties -- finite state, unit quaternions, torque and joint-velocity
    limits respected, standing robots carry their weight, reset bookkeeping consistent."""
    import torch
    from isaacgymloco_amd.envs.legged_robot import LeggedRobot
    cfg = C.aliengo_cfg()
    cfg.env.num_envs = 4096
    cfg.sim.physx.solver_type = SOLVERS[solver]
    env = LeggedRobot(cfg, sim_device="cuda:0", seed=1)
    env.reset()
    g = torch.Generator(device="cuda").manual_seed(0)
    vmax_t = torch.tensor([20, 20, 15.89] * 4, device="cuda")
    n_over, worst = 0, 0.0
    for t in range(60):
        a = torch.randn(4096, 12, device="cuda", generator=g) * 0.3
        env.step_device(a)
        r = env.dof_vel.abs() / vmax_t
        n_over += int((env.dof_vel.abs() > vmax_t + 1e-3).sum()); worst = max(worst, float(r.max()))
    torch.cuda.synchronize()
    root = env.root_states.cpu().numpy()
    assert np.isfinite(root).all() and np.isfinite(env.obs_buf.cpu().numpy()).all() and np.isfinite(env.rew_buf.cpu().numpy()).all()
    np.testing.assert_allclose(np.linalg.norm(root[:, 3:7], axis=1), 1.0, atol=1e-4)
    tau = env.torques.cpu().numpy()
    assert (np.abs(tau) <= np.array([44, 44, 55] * 4) + 1e-4).all()
    # joint speeds against the URDF limits over ALL 60 steps (round 4 looked at the last step only).  The limit is a constraint row for every
    # joint whose FREE velocity comes within 20 % of it; what is left beyond it are joints without a row that a contact
    # impulse kicked (both solvers alike: CPU oracle, same scenario, 20 of 2.9 M joint-steps for PGS, 40 for TGS with its final limit pass,
    # 208 without it) -- bounded by the 1.5 x safety clamp.  ONE bar for both solvers since round 5.
    total = 60 * 4096 * 12
    print(f"{solver}: joint speeds beyond the limit: {n_over} of {total} joint-steps ({n_over / total:.2e}), max ratio {worst:.3f}")
    assert n_over <= 5e-5 * total
    assert worst <= 1.5 + 1e-4
    ep = env.episode_length_buf.cpu().numpy()
    assert ep.min() >= 0 and ep.max() <= 61
    assert np.abs(env.obs_buf.cpu().numpy()).max() <= 100.0
    # weight support: mean vertical contact force of the robots that are standing ~ m g (24.94 kg + payload 0..3)
    fz = env.contact_forces[:, :, 2].sum(1).cpu().numpy()
    standing = (root[:, 2] - env.env_origins[:, 2].cpu().numpy() > 0.25) & (fz > 50)
    assert standing.sum() > 500
    assert 180.0 < np.median(fz[standing]) < 340.0


@pytest.mark.parametrize("N", [1, 13])
def test_hip_plane_ground_ragged_sizes_match_oracle(N):
    """mesh_type 'plane' (LR:1069-1078) at sizes that leave idle blocks in the XCD-striped grid (grid = 8 * ceil(N / 8))."""
    from hip_backend import HipBackend
    cfg = C.TASKS["aliengo"][0]()
    cfg.terrain.mesh_type = "plane"
    orc, lc, model, ter = make_oracle(cfg, N, seed=11)
    be = HipBackend(cfg, N, ter, seed=11)
    orc.reset_all(); be.reset_all()
    rs = np.random.RandomState(2)
    for t in range(10):
        a = rs.normal(0, 1, (N, 12)).astype(np.float32)
        orc.step(a); be.step(a)
        np.testing.assert_array_equal(be.get("reset"), orc.buf["reset"], err_msg=f"step {t}")
        np.testing.assert_allclose(be.get("root_states"), orc.buf["root_states"], atol=2e-3, rtol=1e-3, err_msg=f"step {t}")
        np.testing.assert_allclose(be.get("obs"), orc.buf["obs"], atol=5e-3, rtol=1e-3, err_msg=f"step {t}")
        np.testing.assert_allclose(be.get("rew"), orc.buf["rew"], atol=1e-3, rtol=1e-3, err_msg=f"step {t}")


def test_hip_is_deterministic_per_seed_and_rank():
    """Counter-based Philox keyed by (seed, rank): same key -> bitwise identical trajectories (resets, pushes, noise included);
    another rank -> a different stream on the same terrain (the multi-GPU sharding contract)."""
    import torch
    from isaacgymloco_amd.envs.legged_robot import LeggedRobot
    def run(rank):
        cfg = C.TASKS["aliengo"][0]()
        cfg.env.num_envs = 96
        cfg.env.episode_length_s = 0.5          # forces time-out resets inside the window
        env = LeggedRobot(cfg, sim_device="cuda:0", seed=5, rank=rank)
        env.reset()
        g = torch.Generator(device="cuda:0").manual_seed(0)
        out = []
        for t in range(40):
            if t == 20:      # make the next step a command-curriculum step (LR:307: common_step_counter % max_episode_length == 0)
                env._L.lsim_set_step_counter(env._h, __import__("ctypes").c_int64(int(env.max_episode_length) * 3 - 1))
            obs, priv, rew, done = env.step_device(torch.randn(96, 12, device="cuda:0", generator=g))
            out.append((obs.clone(), rew.clone(), done.clone(), env.stats_row().clone()))
        torch.cuda.synchronize()
        assert sum(int(d.sum()) for _, _, d, _ in out) > 0
        return out
    a, b, c = run(0), run(0), run(1)
    S = abi.STATS
    multi = 0
    for (o1, r1, d1, s1), (o2, r2, d2, s2) in zip(a, b):
        assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2)
        # the per-step reductions over waves -- extras["episode"] sums, reset count, live command ranges -- are bitwise reproducible too:
        # they are accumulated in fixed point, so the arrival order of the waves cannot show
        assert torch.equal(s1[:S["fix"]], s2[:S["fix"]])
        multi += int(s1[S["reset_count"]] > 1)
    assert multi > 0, "no step with several resets: the order-independence of the sums was not exercised"
    assert any(not torch.equal(o1, o3) for (o1, _, _, _), (o3, _, _, _) in zip(a, c))


def test_hip_api_rejects_bad_arguments():
    """error behaviour of the C-ABI on a live handle: negative LSIM_E_* codes, message via lsim_last_error, handle stays usable"""
    import ctypes
    import torch
    from isaacgymloco_amd.envs.legged_robot import LeggedRobot
    cfg = C.TASKS["aliengo"][0]()
    cfg.env.num_envs = 8
    env = LeggedRobot(cfg, sim_device="cuda:0", seed=1)
    L, h = env._L, env._h
    assert L.lsim_step(h, None, None) == abi.E_INVALID
    ptr, shape, nd, dt = ctypes.c_void_p(), (ctypes.c_int64 * 4)(), ctypes.c_int(), ctypes.c_int()
    assert L.lsim_get_buffer(h, abi.NUM_BUFFERS, ctypes.byref(ptr), shape, ctypes.byref(nd), ctypes.byref(dt)) == abi.E_INVALID
    assert L.lsim_get_buffer(h, -1, ctypes.byref(ptr), shape, ctypes.byref(nd), ctypes.byref(dt)) == abi.E_INVALID
    assert L.lsim_step(None, ctypes.c_void_p(env.actions.data_ptr()), None) == abi.E_INVALID
    env.reset()
    obs, _, _, _ = env.step_device(torch.zeros(8, 12, device="cuda:0"))
    assert torch.isfinite(obs).all()


@pytest.mark.parametrize("robot", ["go1", "go2"])
def test_hip_go1_matches_oracle(robot):
    """second robots (tasks "go1" / "go2", model tables robots/tables/go1.json / go2.json -- BASELINE config 5's Go2, kinematics fitted to the
    reference's Go2 clips): HIP vs oracle with full physics and random actions"""
    from hip_backend import HipBackend
    N = 16
    cfg = C.TASKS[robot][0]()
    cfg.terrain.terrain_proportions = [1.0, 0.0, 0.0, 0.0]
    orc, lc, model, ter = make_oracle(cfg, N, seed=5)
    be = HipBackend(cfg, N, ter, seed=5)
    orc.reset_all(); be.reset_all()
    rs = np.random.RandomState(0)
    for t in range(12):
        a = rs.normal(0, 1, (N, 12)).astype(np.float32)
        orc.step(a); be.step(a)
        np.testing.assert_array_equal(be.get("reset"), orc.buf["reset"], err_msg=f"step {t}")
        np.testing.assert_allclose(be.get("root_states"), orc.buf["root_states"], atol=2e-3, rtol=1e-3, err_msg=f"step {t}")
        np.testing.assert_allclose(be.get("dof_state"), orc.buf["dof_state"], atol=5e-3, rtol=1e-3, err_msg=f"step {t}")
        np.testing.assert_allclose(be.get("obs"), orc.buf["obs"], atol=5e-3, rtol=1e-3, err_msg=f"step {t}")


@pytest.mark.parametrize("task", ["aliengo_amp", "aliengo_stairs"])
def test_hip_soak_states_stay_bounded(task):
    """600 steps of rough N(0,1)..3 N(0,1) actions at N = 1024, incl. the AMP task whose config has no termination block (fallen robots
    keep simulating): nothing non-finite, no robot launched (a joint-velocity clamp once pumped momentum into airborne robots until
    velocities reached 1e4 m/s and the AMP task produced NaN)."""
    import torch
    from isaacgymloco_amd.envs.legged_robot import LeggedRobot
    cfg = C.TASKS[task][0]()
    cfg.env.num_envs = 1024
    env = LeggedRobot(cfg, sim_device="cuda:0", seed=3, using_amp=(task == "aliengo_amp"))
    env.reset()
    g = torch.Generator(device="cuda:0").manual_seed(0)
    zmax = torch.zeros((), device="cuda:0"); vmax = torch.zeros((), device="cuda:0")
    for i in range(600):
        obs, priv, rew, done = env.step_device(torch.randn(1024, 12, device="cuda:0", generator=g) * (1.0 if i % 200 < 150 else 3.0))
        zmax = torch.maximum(zmax, env.root_states[:, 2].abs().max())
        vmax = torch.maximum(vmax, env.root_states[:, 7:13].abs().max())
    for t in (obs, priv, rew, env.root_states, env.dof_state, env.contact_forces):
        assert torch.isfinite(t).all()
    assert float(zmax) < 4.0 and float(vmax) < 60.0, (float(zmax), float(vmax))


def test_hip_handles_are_independent_and_stream_ordered():
    """C-ABI hardening: (1) two handles in one process do not share state -- interleaving their steps gives each the trajectory it has
    alone; (2) work is ordered on whatever HIP stream the caller passes, not on the default stream; (3) a library-owned arena
    (arena_dev = NULL, the non-PyTorch host case) produces the same buffers as a caller-owned one."""
    import ctypes
    import torch
    from isaacgymloco_amd import lib
    from isaacgymloco_amd.envs.legged_robot import LeggedRobot

    def make(seed):
        cfg = C.TASKS["aliengo"][0]()
        cfg.env.num_envs = 40
        e = LeggedRobot(cfg, sim_device="cuda:0", seed=seed)
        e.reset()
        return e
    g = torch.Generator(device="cuda:0").manual_seed(1)
    acts = [torch.randn(40, 12, device="cuda:0", generator=g) for _ in range(12)]
    solo = make(21)
    ref = []
    for a in acts:
        ref.append(solo.step_device(a)[0].clone())
    a_env, b_env = make(21), make(22)                       # same seed as `solo`, plus a second simulator in between
    side = torch.cuda.Stream()
    got = []
    for a in acts:
        b_env.step_device(a * 2.0)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                       # non-default stream for handle A
            got.append(a_env.step_device(a)[0].clone())
        torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    for r, o in zip(ref, got):
        assert torch.equal(r, o)
    # ---- library-owned arena through the raw C-ABI
    L = lib.load()
    h = ctypes.c_void_p()
    grid = np.ascontiguousarray(solo.terrain.heightsamples, dtype=np.int16)
    orig = np.ascontiguousarray(solo.terrain.env_origins, dtype=np.float32)
    lib.check(L.lsim_create(ctypes.byref(solo.lcfg), ctypes.byref(solo.model), grid.ctypes.data, orig.ctypes.data, None, 0, ctypes.byref(h)),
              what="lsim_create(NULL arena)")
    lib.check(L.lsim_reset_all(h, None), h, "lsim_reset_all")
    zero = torch.zeros(40, 12, device="cuda:0")
    lib.check(L.lsim_step(h, ctypes.c_void_p(zero.data_ptr()), None), h, "lsim_step")     # LeggedRobot.reset() = reset_all + one zero-action step
    for a in acts[:3]:
        lib.check(L.lsim_step(h, ctypes.c_void_p(a.data_ptr()), None), h, "lsim_step")
    torch.cuda.synchronize()
    ptr, shape, nd, dt = ctypes.c_void_p(), (ctypes.c_int64 * 4)(), ctypes.c_int(), ctypes.c_int()
    lib.check(L.lsim_get_buffer(h, abi.BUFFER_IDS["obs"], ctypes.byref(ptr), shape, ctypes.byref(nd), ctypes.byref(dt)), h, "lsim_get_buffer")
    host = np.empty((shape[0], shape[1]), np.float32)
    import torch.cuda
    tmp = torch.empty(shape[0], shape[1], device="cuda:0")
    ctypes.CDLL("libamdhip64.so").hipMemcpy(ctypes.c_void_p(tmp.data_ptr()), ptr, ctypes.c_size_t(host.nbytes), 3)   # device to device
    torch.cuda.synchronize()
    assert torch.equal(tmp, ref[2])
    L.lsim_destroy(h)


@pytest.mark.gpu
def test_feet_heights_surface_method_matches_a_numpy_restatement():
    """LeggedRobot._get_feet_heights (LR:1400-1441; the reference never calls it): mean of three grid samples under each foot"""
    import numpy as np
    import torch
    from helpers import C
    from isaacgymloco_amd.envs.legged_robot import LeggedRobot
    cfg = C.TASKS["aliengo_stairs"][0]()
    cfg.env.num_envs = 64
    env = LeggedRobot(cfg, sim_device="cuda:0", seed=1)
    env.reset()
    for _ in range(20):
        env.step_device(torch.randn(64, 12, device="cuda:0"))
    got = env._get_feet_heights().cpu().numpy()
    feet = env.feet_pos.cpu().numpy().astype(np.float32)
    grid = env.height_samples.cpu().numpy()
    hs, vs, border = np.float32(cfg.terrain.horizontal_scale), np.float32(cfg.terrain.vertical_scale), np.float32(cfg.terrain.border_size)
    p = ((feet + border) / hs).astype(np.int64)                      # truncation toward zero, as .long()
    px = np.clip(p[:, :, 0], 0, grid.shape[0] - 2)
    py = np.clip(p[:, :, 1], 0, grid.shape[1] - 2)
    mean3 = (grid[px, py].astype(np.float32) + grid[px + 1, py] + grid[px, py + 1]) / np.float32(3)
    want = feet[:, :, 2] - mean3 * vs
    assert got.shape == (64, 4) and np.allclose(got, want, atol=1e-5)
    sub = env._get_feet_heights(torch.tensor([3, 7], device="cuda:0")).cpu().numpy()
    assert np.allclose(sub, want[[3, 7]], atol=1e-5)
    env.close()


def test_hip_leg_kinematics_reproduce_the_reference_mocap_toe_positions():
    """P1 on the device: the joint angles of the reference's Aliengo mocap frames (isaacgymloco_amd/data/mocap_aliengo.npz: the reference's
    files re-packed unchanged) through kernel A's own kinematics -- one sub-step of 1 us, robots 5 m above the ground, so that the published
    rigid_body_states are the forward kinematics of the injected joint angles to < 1e-6 -- against the toe positions in the base frame that
    the reference's retargeting tool stored beside them (columns 19:31).  658 frames x 4 feet; CPU twin: tests/test_model.py."""
    import os
    from hip_backend import HipBackend
    data = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "isaacgymloco_amd", "data", "mocap_aliengo.npz"), allow_pickle=True)
    frames = np.concatenate([data[f"frames_{c}"] for c in range(int(data["num_clips"]))])
    N = len(frames)
    cfg = quiet_cfg("aliengo")
    cfg.sim.dt = 1e-6
    cfg.control.decimation = 1
    be = HipBackend(cfg, N, T.Terrain(cfg.terrain, N, seed=1), seed=1)
    be.reset_all()
    root = np.zeros((N, 13), np.float32); root[:, 2] = 5.0; root[:, 6] = 1.0
    dof = np.zeros((N, 12, 2), np.float32); dof[:, :, 0] = frames[:, 7:19]
    be.put("root_states", root); be.put("dof_state", dof)
    be.step(np.zeros((N, 12), np.float32), flags=abi.STEP_NO_RESET)
    body = be.get("rigid_body_states")
    feet = [4, 8, 12, 16]
    got = body[:, feet, 0:3] - be.get("root_states")[:, None, 0:3]
    want = frames[:, 19:31].reshape(N, 4, 3)
    assert float(np.abs(got - want).max()) < 5e-5, float(np.abs(got - want).max())      # fp32 chain of three rotations at |p| ~ 0.4 m + 5 m offset


def test_hip_recover_task_matches_oracle():
    """task "aliengo_recover" on the device: robots on their backs and sides (resets draw roll / pitch / yaw from +-3.14), resting on trunk and
    upper legs, `_up` reward set -- HIP kernels vs the oracle, the same check as tests/test_emu_golden.py runs on the lane emulator"""
    from hip_backend import HipBackend
    from test_emu_golden import _recover_task_check
    _recover_task_check(lambda cfg, lc, model, ter, N: HipBackend(cfg, N, ter, seed=3), lambda be, k: be.get(k), lambda be, k, v: be.put(k, v), steps=16)


@pytest.mark.parametrize("task", ["aliengo", "aliengo_stairs", "aliengo_amp"])
def test_hip_fused_tail_equals_the_two_kernel_form(task):
    """kernel A running reset_idx + observations for its own robot (+ lsim_k_step_finish) against kernels A + B on every step
    (LSIM_STEP_TWO_KERNELS): N = 4096, full physics, time-out and fall resets, a command-curriculum step inside the window -- every
    buffer of the arena bit for bit after every step (the stats rows up to the two-kernel form's ticket word)"""
    import torch
    from isaacgymloco_amd.envs.legged_robot import LeggedRobot
    envs = []
    for _ in range(2):
        cfg = C.TASKS[task][0]()
        cfg.env.num_envs = 4096
        cfg.env.episode_length_s = 0.6
        envs.append(LeggedRobot(cfg, sim_device="cuda:0", seed=11, using_amp=(task == "aliengo_amp")))
    ea, eb = envs
    ea.reset(); eb.reset()
    g = torch.Generator(device="cuda:0").manual_seed(1)
    resets = 0
    for t in range(45):
        if t == 20:
            for e in envs:
                e._L.lsim_set_step_counter(e._h, __import__("ctypes").c_int64(int(e.max_episode_length) * 2 - 1))
        act = torch.randn(4096, 12, device="cuda:0", generator=g)
        ea.step_device(act); eb.step_device(act, flags=abi.STEP_TWO_KERNELS)
        torch.cuda.synchronize()
        for name in abi.BUFFER_IDS:
            xa, xb = ea.buf[name], eb.buf[name]
            if name == "stats":
                xa, xb = xa[:, :abi.STATS["fix"]], xb[:, :abi.STATS["fix"]]
            assert torch.equal(xa, xb), f"step {t}: buffer {name} differs between the fused and the two-kernel form"
        resets += int(ea.reset_buf.sum())
    assert resets > 4096
