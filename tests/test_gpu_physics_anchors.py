"""GPU: anchors for the unpinned dynamics that do not need PhysX.  The reference's simulator is closed source, so none
of this is reference parity; these tests bound the discretisation and solver errors of the build's own physics on the HIP library:
  * time-step halving: the integrator is first order, so the error against a 4x finer run falls by ~3x when the step is halved;
  * Gauss-Seidel sweeps: 8 (shipped) against 64 -- the state difference after one second is bounded;
  * the 8-contact cap: how often it binds at BASELINE size on the stairs task (histogram from LSIM_BUF_CONTACT_COUNT);
  * BASELINE-size (N = 4096) invariants for the stairs and AMP configurations (the flat one is in test_gpu_parity.py);
  * mechanical energy of a passive robot in free flight, evaluated from the published body states and the model table alone: first-order drift;
  * the response to joint torques against M^-1 tau with the mass matrix M assembled from kinetic energies of the published body states;
  * impulse-momentum: the reported net contact forces account for the change of the linear momentum through an impact, a skid and stance.
The configuration being replaced is legged_robot_config.py:238-255 (dt 5 ms, TGS, 4 position iterations)."""
import numpy as np
import pytest
import torch

from helpers import C, T, quiet_cfg

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _env(cfg, N, seed=1):
    from isaacgymloco_amd.envs.legged_robot import LeggedRobot
    cfg.env.num_envs = N
    return LeggedRobot(cfg, sim_device=DEV, seed=seed)


def _trajectory(refine, scenario, steps=25):
    """25 control steps (0.5 s) at sim dt = 5 ms / refine with decimation 4 * refine: the control period stays 20 ms"""
    cfg = quiet_cfg()
    cfg.sim.dt = 0.005 / refine
    cfg.control.decimation = 4 * refine
    cfg.termination.fall_down = False
    if scenario == "flight":       # tumbling free flight under gravity, PD loop holding a moving pose: no contact within 0.5 s from 3 m
        cfg.init_state.pos = [0.0, 0.0, 3.0]
        cfg.domain_rand.base_init_vel_range = dict(x=[0.3, 0.3], y=[-0.2, -0.2], z=[0.0, 0.0], roll=[0.8, 0.8], pitch=[-0.5, -0.5], yaw=[0.4, 0.4])
    else:                          # dropped from just above its standing height: settles on four feet
        cfg.init_state.pos = [0.0, 0.0, 0.40]
    env = _env(cfg, 8)
    env.reset()
    g = torch.Generator(device=DEV).manual_seed(3)
    amp = 0.5 if scenario == "flight" else 0.1
    for t in range(steps):
        env.step_device(amp * torch.randn(8, 12, device=DEV, generator=g).clamp(-2, 2))
    torch.cuda.synchronize()
    assert int(env.reset_buf.sum()) == 0
    out = np.concatenate([env.root_states[:, :7].cpu().numpy(), env.dof_pos.cpu().numpy()], axis=1).astype(np.float64)
    cf = env.contact_forces.abs().sum().item()
    env.close()
    return out, cf


@pytest.mark.parametrize("scenario", ["flight", "stance"])
def test_time_step_halving_is_first_order(scenario):
    x1, c1 = _trajectory(1, scenario)
    x2, _ = _trajectory(2, scenario)
    x4, _ = _trajectory(4, scenario)
    assert (c1 == 0.0) == (scenario == "flight")                        # flight: no contact at all; stance: feet loaded
    e1 = np.abs(x1 - x4).max(axis=1)                                    # per env, against the 1.25 ms run
    e2 = np.abs(x2 - x4).max(axis=1)
    # x(dt) = x* + C dt + O(dt^2):  e1 = 0.75 C dt,  e2 = 0.25 C dt  ->  ratio 3 for a first-order scheme (5 for a second-order one)
    ratio = np.median(e1 / np.maximum(e2, 1e-9))
    print(f"{scenario}: median error 5 ms {np.median(e1):.2e}, 2.5 ms {np.median(e2):.2e}, ratio {ratio:.2f}")
    if scenario == "flight":
        assert 2.2 < ratio < 5.5, ratio                                 # measured 4.25: semi-implicit Euler, error dominated by the O(dt) term
        assert np.median(e1) < 8e-2                                     # 0.5 s of tumbling at 5 ms: centimetres / centiradians (measured 4e-2)
    else:                                                               # contacts switch on and off: only monotone improvement is asserted
        assert np.median(e2) < np.median(e1)
        assert np.median(e1) < 2e-2


def test_eight_gauss_seidel_sweeps_against_sixty_four(monkeypatch):
    """the shipped 8 sweeps against 64 on the same seeds: one second of standing / small-action motion on flat ground"""
    from isaacgymloco_amd.envs import lsim_config as LC

    def run(iters):
        monkeypatch.setitem(LC.SOLVER_DEFAULTS, "solver_iterations", iters)
        cfg = quiet_cfg()
        cfg.sim.physx.solver_type = 0         # the velocity-level sweeps (the TGS default has no sweep count to vary)
        cfg.init_state.pos = [0.0, 0.0, 0.40]
        env = _env(cfg, 64)
        assert env.lcfg.solver_iterations == iters
        env.reset()
        g = torch.Generator(device=DEV).manual_seed(5)
        for t in range(50):
            env.step_device(0.25 * torch.randn(64, 12, device=DEV, generator=g))
        torch.cuda.synchronize()
        assert int(env.reset_buf.sum()) == 0
        out = (env.root_states.cpu().numpy().astype(np.float64), env.dof_pos.cpu().numpy().astype(np.float64),
               env.contact_forces[:, :, 2].sum(1).cpu().numpy())
        env.close()
        return out
    r8, q8, f8 = run(8)
    r64, q64, f64 = run(64)
    dz = np.abs(r8[:, 2] - r64[:, 2])
    dxy = np.linalg.norm(r8[:, :2] - r64[:, :2], axis=1)
    dq = np.abs(q8 - q64).max(axis=1)
    print(f"8 vs 64 sweeps after 1 s: base height diff median {np.median(dz):.2e} max {dz.max():.2e} m; xy {np.median(dxy):.2e} / {dxy.max():.2e} m; "
          f"joint {np.median(dq):.2e} / {dq.max():.2e} rad; vertical load {np.median(f8):.1f} vs {np.median(f64):.1f} N")
    assert np.median(dz) < 2e-3 and dz.max() < 2e-2
    assert np.median(dxy) < 1e-2
    assert np.median(dq) < 2e-2
    assert abs(np.median(f8) - np.median(f64)) < 0.03 * np.median(f64)          # both carry the robot's weight


def test_contact_cap_histogram_at_baseline_size_on_stairs():
    """aliengo_stairs, N = 4096, 1000 steps of N(0,1) actions: share of env-steps whose uncapped contact count exceeds LSIM_MAX_CONTACTS"""
    cfg = C.TASKS["aliengo_stairs"][0]()
    env = _env(cfg, 4096)
    env.reset()
    env.episode_length_buf = torch.randint_like(env.episode_length_buf, high=int(env.max_episode_length))
    g = torch.Generator(device=DEV).manual_seed(0)
    hist = torch.zeros(65, dtype=torch.long, device=DEV)
    for t in range(1000):
        env.step_device(torch.randn(4096, 12, device=DEV, generator=g))
        hist += torch.bincount(env.buf["contact_count"][:, 0].long().clamp(0, 64), minlength=65)
    torch.cuda.synchronize()
    h = hist.cpu().numpy()
    over = h[9:].sum() / h.sum()
    print("active collision points per env-step (max over sub-steps), counts 0..12:", h[:13].tolist(), f"-> cap binds in {100 * over:.3f} % of env-steps")
    assert h.sum() == 4096 * 1000
    assert over < 0.02
    assert h[1:9].sum() > 0.5 * h.sum()          # the robots were on the ground most of the time


@pytest.mark.parametrize("task", ["aliengo_stairs", "aliengo_amp"])
def test_full_size_invariants_for_the_other_baseline_configs(task):
    """BASELINE configs 3 and 4 at N = 4096: finite state, unit quaternions, torque / joint-velocity limits, bounded observations, reset
    bookkeeping, weight support of the standing robots"""
    cfg = C.TASKS[task][0]()
    cfg.env.num_envs = 4096
    from isaacgymloco_amd.envs.legged_robot import LeggedRobot
    env = LeggedRobot(cfg, sim_device=DEV, seed=1, using_amp=(task == "aliengo_amp"))
    env.reset()
    g = torch.Generator(device=DEV).manual_seed(0)
    for t in range(60):
        env.step_device(torch.randn(4096, 12, device=DEV, generator=g) * 0.3)
    torch.cuda.synchronize()
    root = env.root_states.cpu().numpy()
    for name in ("obs_buf", "privileged_obs_buf", "rew_buf", "amp_obs_buf"):
        assert np.isfinite(getattr(env, name).cpu().numpy()).all(), name
    assert np.isfinite(root).all()
    np.testing.assert_allclose(np.linalg.norm(root[:, 3:7], axis=1), 1.0, atol=1e-4)
    assert (np.abs(env.torques.cpu().numpy()) <= np.array([44, 44, 55] * 4) + 1e-4).all()
    assert (np.abs(env.dof_vel.cpu().numpy()) <= 1.5 * np.array([20, 20, 15.89] * 4) + 1e-3).all()
    ep = env.episode_length_buf.cpu().numpy()
    assert ep.min() >= 0 and ep.max() <= 61
    assert np.abs(env.obs_buf.cpu().numpy()).max() <= 100.0
    fz = env.contact_forces[:, :, 2].sum(1).cpu().numpy()
    standing = (root[:, 2] - env.env_origins[:, 2].cpu().numpy() > 0.25) & (fz > 50)
    assert standing.sum() > 500
    assert 180.0 < np.median(fz[standing]) < 340.0
    if task == "aliengo_amp":     # AMP features are the raw joint / base state (LR:406-416)
        amp = env.amp_obs_buf.cpu().numpy()
        np.testing.assert_allclose(amp[:, :12], env.dof_pos.cpu().numpy(), atol=1e-6)
        np.testing.assert_allclose(amp[:, 18:30], env.dof_vel.cpu().numpy(), atol=1e-6)


def _total_mass():
    import json, os
    from helpers import ROOT
    return sum(b["mass"] for b in json.load(open(os.path.join(ROOT, "isaacgymloco_amd", "robots", "tables", "aliengo.json")))["bodies"])


def _mechanical_energy(env, g=9.81):
    """kinetic + potential energy of every robot from the simulator's own outputs (rigid_body_states: link-origin position, quaternion xyzw,
    linear velocity of the centre of mass -- or of the link origin, lsim_config.lin_vel_at_com -- angular velocity, world frame) and the model table (mass, centre of mass and inertia about it in the link
    frame) -- nothing of the build's dynamics code is used"""
    import json, os
    from helpers import ROOT
    bodies = json.load(open(os.path.join(ROOT, "isaacgymloco_amd", "robots", "tables", "aliengo.json")))["bodies"]
    s = env.rigid_body_states.view(env.num_envs, 17, 13).double().cpu().numpy()
    E = np.zeros(env.num_envs)
    for b, bd in enumerate(bodies):
        p, q, v, w = s[:, b, 0:3], s[:, b, 3:7], s[:, b, 7:10], s[:, b, 10:13]
        x, y, z, ww = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
        R = np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * ww), 2 * (x * z + y * ww)], -1),
                      np.stack([2 * (x * y + z * ww), 1 - 2 * (x * x + z * z), 2 * (y * z - x * ww)], -1),
                      np.stack([2 * (x * z - y * ww), 2 * (y * z + x * ww), 1 - 2 * (x * x + y * y)], -1)], 1)      # (N, 3, 3)
        c = np.einsum("nij,j->ni", R, np.asarray(bd["com"], dtype=np.float64))
        vc = v if env.lcfg.lin_vel_at_com else v + np.cross(w, c)       # include/lsim.h lin_vel_at_com: the tensor already holds the COM's velocity
        Iw = np.einsum("nij,jk,nlk->nil", R, np.asarray(bd["inertia"], dtype=np.float64), R)
        E += 0.5 * bd["mass"] * (vc * vc).sum(1) + 0.5 * np.einsum("ni,nij,nj->n", w, Iw, w) + bd["mass"] * g * (p[:, 2] + c[:, 2])
    return E


def _energy_run(refine, steps=20):
    cfg = quiet_cfg()
    cfg.sim.dt = 0.005 / refine
    cfg.control.decimation = 4 * refine
    cfg.termination.fall_down = False
    cfg.control.stiffness = {"joint": 0.0}          # no actuation: a passive multibody in free flight
    cfg.control.damping = {"joint": 0.0}
    cfg.init_state.pos = [0.0, 0.0, 4.0]
    cfg.domain_rand.base_init_vel_range = dict(x=[0.5, 0.5], y=[-0.3, -0.3], z=[1.0, 1.0], roll=[1.5, 1.5], pitch=[-1.0, -1.0], yaw=[0.7, 0.7])
    env = _env(cfg, 8)
    env.reset()
    g = torch.Generator(device=DEV).manual_seed(5)
    env.dof_vel[:] = 1.5 * (torch.rand(8, 12, device=DEV, generator=g) - 0.5)        # joints swinging at up to 0.75 rad/s: no stop is reached in 0.4 s
    zero = torch.zeros(8, 12, device=DEV)
    env.step_device(zero)                                                             # body states are those of the end of a step
    E0 = _mechanical_energy(env)
    K0 = E0 - _potential_only(env)
    for _ in range(steps):
        env.step_device(zero)
    torch.cuda.synchronize()
    assert env.contact_forces.abs().sum().item() == 0.0 and int(env.reset_buf.sum()) == 0
    E1 = _mechanical_energy(env)
    env.close()
    return E0, E1, K0


def _potential_only(env):
    import json, os
    from helpers import ROOT
    bodies = json.load(open(os.path.join(ROOT, "isaacgymloco_amd", "robots", "tables", "aliengo.json")))["bodies"]
    s = env.rigid_body_states.view(env.num_envs, 17, 13).double().cpu().numpy()
    U = np.zeros(env.num_envs)
    for b, bd in enumerate(bodies):
        q = s[:, b, 3:7]
        x, y, z, ww = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
        Rz = np.stack([2 * (x * z - y * ww), 2 * (y * z + x * ww), 1 - 2 * (x * x + y * y)], -1)
        U += bd["mass"] * 9.81 * (s[:, b, 2] + Rz @ np.asarray(bd["com"], dtype=np.float64))
    return U


def test_mechanical_energy_of_a_passive_robot_in_free_flight():
    """VERDICT r1 item 6 (energy drift): with the actuators off and no contact, kinetic + potential energy -- evaluated from the published body
    states and the model table alone -- is an invariant of the exact dynamics.  The first-order integrator makes it drift; the drift must be
    small against the kinetic energy in play and shrink with the time step."""
    E0, E1, K0 = _energy_run(1)
    F0, F1, _ = _energy_run(2)
    M, T_ = _total_mass(), 20 * 0.02
    # symplectic Euler in a uniform field loses exactly M g^2 dt / 2 per second on the falling centre of mass (v first, then x with the new v);
    # what is left after that closed-form term is the drift of the articulated / rotational part
    ff1, ff2 = -0.5 * M * 9.81 ** 2 * 0.005 * T_, -0.5 * M * 9.81 ** 2 * 0.0025 * T_
    d1, d2 = (E1 - E0) / K0, (F1 - F0) / K0
    r1, r2 = (E1 - E0 - ff1) / K0, (F1 - F0 - ff2) / K0
    print(f"kinetic energy {np.median(K0):.1f} J; relative drift over 0.4 s: {np.median(d1):+.3e} at 5 ms, {np.median(d2):+.3e} at 2.5 ms; "
          f"without the free-fall term: {np.median(r1):+.3e}, {np.median(r2):+.3e}")
    assert np.median(np.abs(d2)) < 0.65 * np.median(np.abs(d1))          # first order: halves with the step
    assert np.median(np.abs(r1)) < 0.10                                  # articulated + rotational part (measured +6.6 % of 10 J: explicit update of
    assert np.median(np.abs(r2)) < 0.65 * np.median(np.abs(r1))          # the orientation gains energy), also first order (measured +3.3 % at 2.5 ms)


def _rest_cfg(dt):
    """one sub-step of `dt` per control step, no gravity, a tilted base, the default joint pose"""
    cfg = quiet_cfg()
    cfg.sim.dt = dt
    cfg.sim.gravity = [0.0, 0.0, 0.0]
    cfg.control.decimation = 1
    cfg.termination.fall_down = False
    cfg.init_state.pos = [0.0, 0.0, 4.0]
    cfg.domain_rand.base_init_rot_range = dict(roll=[0.3, 0.3], pitch=[-0.2, -0.2], yaw=[0.0, 0.0])
    return cfg


def test_torque_response_against_a_mass_matrix_built_from_the_kinematic_outputs():
    """The joint-space inertia matrix M (18 x 18: base linear, base angular, 12 joints) is assembled WITHOUT any of the build's dynamics
    code: kinetic energy, evaluated from the published body states and the model table, for unit generalised velocities and their pairs
    (M_ij = T(e_i + e_j) - T(e_i) - T(e_j)).  From rest and without gravity the bias forces vanish, so one short step under joint torques
    tau must change the generalised velocity by dt * M^-1 [0; tau] -- which is what the structured solver (composite inertias, leg / base
    Schur complement, Cholesky factors) has to reproduce."""
    nv = 18
    pairs = [(i, j) for i in range(nv) for j in range(i + 1, nv)]
    N = nv + len(pairs)
    cfg = _rest_cfg(1e-6)
    cfg.control.stiffness = {"joint": 0.0}
    cfg.control.damping = {"joint": 0.0}
    env = _env(cfg, N)
    env.reset()
    V = torch.zeros(N, nv, device=DEV)
    for i in range(nv):
        V[i, i] = 1.0
    for k, (i, j) in enumerate(pairs):
        V[nv + k, i] = 1.0
        V[nv + k, j] = 1.0
    env.root_states[:, 7:13] = V[:, :6]              # base linear (world) and angular (world) velocity
    env.dof_vel[:] = V[:, 6:]
    env.step_device(torch.zeros(N, 12, device=DEV))  # 1 us: the body states of the prescribed velocities
    T_ = _mechanical_energy(env, g=0.0)
    env.close()
    M = np.zeros((nv, nv))
    for i in range(nv):
        M[i, i] = 2.0 * T_[i]
    for k, (i, j) in enumerate(pairs):
        M[i, j] = M[j, i] = T_[nv + k] - T_[i] - T_[j]
    assert abs(M[0, 0] - _total_mass()) < 1e-3 and np.all(np.linalg.eigvalsh(M) > 0)     # total mass on the translation block; positive definite

    dt = 1e-4
    cfg = _rest_cfg(dt)
    cfg.control.stiffness = {"joint": 1.0}           # tau = Kp * action_scale * a at the default pose (q = q0, qd = 0): torques set by the actions
    cfg.control.damping = {"joint": 0.0}
    env = _env(cfg, 16)
    env.reset()
    env.root_states[:, 7:13] = 0.0
    env.dof_vel[:] = 0.0
    g = torch.Generator(device=DEV).manual_seed(2)
    tau = 20.0 * (torch.rand(16, 12, device=DEV, generator=g) - 0.5)
    tau[:12] = 10.0 * torch.eye(12, device=DEV)      # one joint at a time, then 4 random combinations
    env.step_device(tau / cfg.control.action_scale)
    assert torch.allclose(env.torques, tau, atol=1e-4)
    dv = torch.cat([env.root_states[:, 7:13], env.dof_vel], dim=1).double().cpu().numpy() / dt
    env.close()
    rhs = np.zeros((16, nv)); rhs[:, 6:] = tau.double().cpu().numpy()
    ref = np.linalg.solve(M, rhs.T).T
    err = np.abs(dv - ref).max(axis=1) / np.abs(ref).max(axis=1)
    print(f"generalised acceleration under joint torques vs M^-1 tau with M from the kinematic outputs: max relative error {err.max():.2e}")
    assert err.max() < 1e-4          # measured 9e-7


def _linear_momentum(env):
    import json, os
    from helpers import ROOT
    bodies = json.load(open(os.path.join(ROOT, "isaacgymloco_amd", "robots", "tables", "aliengo.json")))["bodies"]
    s = env.rigid_body_states.view(env.num_envs, 17, 13).double().cpu().numpy()
    P = np.zeros((env.num_envs, 3))
    for b, bd in enumerate(bodies):
        q, v, w = s[:, b, 3:7], s[:, b, 7:10], s[:, b, 10:13]
        x, y, z, ww = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
        R = np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * ww), 2 * (x * z + y * ww)], -1),
                      np.stack([2 * (x * y + z * ww), 1 - 2 * (x * x + z * z), 2 * (y * z - x * ww)], -1),
                      np.stack([2 * (x * z - y * ww), 2 * (y * z + x * ww), 1 - 2 * (x * x + y * y)], -1)], 1)
        c = np.einsum("nij,j->ni", R, np.asarray(bd["com"], dtype=np.float64))
        P += bd["mass"] * (v if env.lcfg.lin_vel_at_com else v + np.cross(w, c))
    return P


def _momentum_run(dt, seconds=0.8):
    cfg = quiet_cfg()
    cfg.sim.dt = dt
    cfg.control.decimation = 1
    cfg.termination.fall_down = False
    cfg.init_state.pos = [0.0, 0.0, 0.55]
    cfg.domain_rand.base_init_vel_range = dict(x=[0.8, 0.8], y=[-0.4, -0.4], z=[0.0, 0.0], roll=[0.0, 0.0], pitch=[0.0, 0.0], yaw=[0.5, 0.5])
    cfg.control.stiffness = {"joint": 20.0}          # soft gains: no velocity-limit rows (they act inside the robot and cancel anyway)
    cfg.control.damping = {"joint": 0.5}
    env = _env(cfg, 8)
    env.reset()
    zero = torch.zeros(8, 12, device=DEV)
    env.step_device(zero)
    P = _linear_momentum(env)
    W = np.array([0.0, 0.0, -_total_mass() * 9.81])
    resid, force = [], []
    for k in range(int(round(seconds / dt))):        # free fall, impact on four feet, skid to rest
        env.step_device(zero)
        Pn = _linear_momentum(env)
        F = env.contact_forces.double().cpu().numpy().sum(1)          # (N, 3) total of the per-body net contact forces of this sub-step
        resid.append(np.abs((Pn - P) - (F + W) * dt).max(axis=1))
        force.append(np.abs(F).max(axis=1))
        P = Pn
    env.close()
    return np.array(resid), np.array(force)


def test_reported_contact_forces_account_for_the_change_of_momentum():
    """impulse-momentum theorem on the published tensors (P5, LR:943-944): with one sub-step per control step every sub-step's net contact
    forces are visible, and the robot's linear momentum -- from the body states and the model table -- must change per step by
    (sum of the reported contact forces + weight) * dt: landing impact, sliding friction and steady stance alike."""
    r1, f1 = _momentum_run(0.005)
    r2, f2 = _momentum_run(0.0025)
    k = int(np.argmax(r1.max(axis=1)))
    s1, s2 = r1.sum(0).max(), r2.sum(0).max()
    print(f"peak total contact force {f1.max():.0f} N; worst per-step momentum residual {r1.max():.2e} N s at 5 ms (step {k}, impulse "
          f"{f1[k].max() * 0.005:.1f} N s), {r2.max():.2e} at 2.5 ms; residuals summed over 0.8 s: {s1:.2f} N s at 5 ms, {s2:.2f} at 2.5 ms")
    assert f1.max() > 400.0 and (f1.max(axis=1) > 50.0).sum() > 60          # the scenario really lands and stands
    # the solver's velocity update is exactly M^-1 J^T lambda; what is left is the configuration moving under the same generalised velocity
    # within a step (first-order integrator): a few per cent of the impact step's impulse, and it halves with the step
    assert r1.max() < 0.04 * f1[k].max() * 0.005
    assert s2 < 0.65 * s1
