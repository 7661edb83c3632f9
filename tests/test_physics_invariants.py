"""Physical invariants of the build's dynamics (no reference oracle exists for PhysX): checked on the CPU
oracle, on the lane-emulated kernel sources and (-m gpu) on the HIP library itself.  These are NOT reference parity."""
import numpy as np
import pytest

from helpers import C, make_oracle, quiet_cfg, abi, LC, aliengo, T

G = 9.81


class _HipSim:
    """the HIP library behind the same numpy surface as the oracle / emulator objects (sim.buf[name] is a host copy)"""

    class _Buf:
        def __init__(self, be):
            self.be = be

        def __getitem__(self, name):
            return self.be.get(name)

    def __init__(self, cfg, N, ter, seed):
        from hip_backend import HipBackend
        self.be = HipBackend(cfg, N, ter, seed=seed)
        self.buf = _HipSim._Buf(self.be)
        self.cfg = self.be.env.lcfg

    def step(self, a, flags=0):
        self.be.step(a, flags)

    def reset_all(self):
        self.be.reset_all()


KINDS = ["oracle", "emu", pytest.param("hip", marks=pytest.mark.gpu)]      # the HIP leg runs with -m gpu on the MI355X
SOLVERS = {"tgs": 1, "pgs": 0}      # cfg.sim.physx.solver_type (LRC:245): 1 = TGS, 4 position iterations (every reference config); 0 = 8 velocity-level sweeps
both_solvers = pytest.mark.parametrize("solver", ["tgs", "pgs"])


def _quiet(solver, **kw):
    cfg = quiet_cfg(**kw)
    cfg.sim.physx.solver_type = SOLVERS[solver]
    return cfg


def _make(kind, cfg, N, seed=1):
    if kind == "hip":
        from isaacgymloco_amd.envs.legged_robot import build_robot_model
        ter = T.Terrain(cfg.terrain, N, seed=1)
        return _HipSim(cfg, N, ter, seed), build_robot_model(cfg.asset)
    orc, lc, model, ter = make_oracle(cfg, N, seed=seed)
    if kind == "oracle":
        return orc, model
    import emu_binding
    return emu_binding.EmuSim(lc, model, ter.heightsamples, ter.env_origins), model


def _total_momentum(sim, model, e=0):
    """linear momentum and angular momentum about the world origin from rigid_body_states (COM motion + spin)."""
    bs = sim.buf["rigid_body_states"][e].astype(np.float64)
    P, Lm = np.zeros(3), np.zeros(3)
    for i in range(17):
        b = model.bodies[i]
        q = bs[i, 3:7]
        x, y, z, w = q
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        c = bs[i, 0:3] + R @ np.array(b.com)
        om = bs[i, 10:13]
        vc = bs[i, 7:10] if sim.cfg.lin_vel_at_com else bs[i, 7:10] + np.cross(om, c - bs[i, 0:3])      # include/lsim.h: lin_vel_at_com
        I = np.array(b.inertia)
        Il = np.array([[I[0], I[1], I[2]], [I[1], I[3], I[4]], [I[2], I[4], I[5]]])
        P += b.mass * vc
        Lm += np.cross(c, b.mass * vc) + R @ Il @ R.T @ om
    return P, Lm


def _tumble(kind, solver, sim_dt, steps, gz):
    cfg = _quiet(solver)
    cfg.init_state.pos = [0.0, 0.0, 3.0]
    cfg.sim.dt = sim_dt
    cfg.sim.gravity = [0.0, 0.0, gz]
    cfg.domain_rand.base_init_vel_range = dict(x=[0.3, 0.3], y=[-0.2, -0.2], z=[0.0, 0.0], roll=[0.8, 0.8], pitch=[-0.5, -0.5], yaw=[0.4, 0.4])
    cfg.termination.fall_down = False
    sim, model = _make(kind, cfg, 2)
    sim.reset_all()
    a = np.zeros((2, 12), np.float32)
    sim.step(a)
    P0, L0 = _total_momentum(sim, model)
    for _ in range(steps):
        sim.step(a)
    P1, L1 = _total_momentum(sim, model)
    assert sim.buf["reset"].sum() == 0
    return P0, L0, P1, L1, sum(b.mass for b in model.bodies)


@both_solvers
@pytest.mark.parametrize("kind", KINDS)
def test_momentum_conserved_without_gravity(kind, solver):
    """Free flight, g = 0, tumbling robot holding its pose with the PD loop (internal forces only): linear and angular
    momentum are conserved; the residual is the first-order integration error and halves with the time step."""
    P0, L0, P1, L1, mass = _tumble(kind, solver, 0.005, 10, 0.0)
    assert np.linalg.norm(P1 - P0) < 1.5e-3 * np.linalg.norm(P0)
    assert np.linalg.norm(L1 - L0) < 2e-2 * np.linalg.norm(L0)
    Ph0, Lh0, Ph1, Lh1, _ = _tumble(kind, solver, 0.0025, 20, 0.0)
    assert np.linalg.norm(Ph1 - Ph0) < 0.65 * np.linalg.norm(P1 - P0)
    assert np.linalg.norm(Lh1 - Lh0) < 0.65 * np.linalg.norm(L1 - L0)


@both_solvers
@pytest.mark.parametrize("kind", KINDS)
def test_free_fall_under_gravity(kind, solver):
    """dP_z = -m g t (to the integrator's first order); the horizontal leak of the semi-implicit Euler step is O(dt)."""
    P0, L0, P1, L1, mass = _tumble(kind, solver, 0.005, 10, -G)
    t = 10 * 4 * 0.005
    np.testing.assert_allclose(P1[2] - P0[2], -mass * G * t, rtol=2e-3)
    Ph0, Lh0, Ph1, Lh1, _ = _tumble(kind, solver, 0.0025, 20, -G)
    assert np.linalg.norm((Ph1 - Ph0)[:2]) < 0.65 * np.linalg.norm((P1 - P0)[:2])
    assert np.linalg.norm((P1 - P0)[:2]) < 0.04 * np.linalg.norm(P0[:2])


@both_solvers
@pytest.mark.parametrize("kind", KINDS)
def test_static_stance_supports_weight(kind, solver):
    cfg = _quiet(solver)
    cfg.init_state.pos = [0.0, 0.0, 0.40]
    sim, model = _make(kind, cfg, 2)
    sim.reset_all()
    a = np.zeros((2, 12), np.float32)
    for _ in range(100):
        sim.step(a)
    fz = sim.buf["contact_forces"][0, :, 2].sum()
    mass = sum(b.mass for b in model.bodies)
    assert abs(fz - mass * G) < 0.03 * mass * G, (fz, mass * G)
    feet = sim.buf["contact_forces"][0, [4, 8, 12, 16], 2]
    assert (feet > 20).all()                                                        # all four feet carry load
    assert abs(sim.buf["root_states"][0, 9]) < 0.02                                 # at rest
    np.testing.assert_allclose(sim.buf["contact_forces"][0, :, :2].sum(0), 0.0, atol=0.05 * mass * G)
    # left/right symmetry of the settled pose
    q = sim.buf["dof_state"][0, :, 0]
    np.testing.assert_allclose(q[1:3], q[4:6], atol=0.02)


@both_solvers
@pytest.mark.parametrize("kind", KINDS)
def test_limits_respected(kind, solver):
    cfg = _quiet(solver)
    sim, model = _make(kind, cfg, 4)
    sim.reset_all()
    rs = np.random.RandomState(1)
    within = total = 0
    for t in range(40):
        sim.step((rs.normal(0, 6.0, (4, 12))).astype(np.float32))                   # violent actions
        tau = sim.buf["torques"]
        assert (np.abs(tau) <= np.array([44, 44, 55] * 4) + 1e-4).all()
        qd = sim.buf["dof_state"][..., 1]
        vmax = np.array([20, 20, 15.89] * 4)
        assert (np.abs(qd) <= 1.5 * vmax + 1e-3).all()                              # hard bound on solver residue
        within += int((np.abs(qd) <= 1.01 * vmax).sum()); total += qd.size          # the limit itself is a constraint row (8 PGS sweeps)
        q = sim.buf["dof_state"][..., 0]
        lo = np.array([model.dof_pos_lower[j] for j in range(12)]); hi = np.array([model.dof_pos_upper[j] for j in range(12)])
        assert (q > lo - 0.08).all() and (q < hi + 0.08).all()                      # soft: resolved at velocity level (measured: 0.056 without, 0.063 with the final limit pass)
        assert np.isfinite(sim.buf["root_states"]).all()
        np.testing.assert_allclose(np.linalg.norm(sim.buf["root_states"][:, 3:7], axis=1), 1.0, atol=1e-5)
    assert within >= 0.95 * total, (within, total)


@both_solvers
@pytest.mark.parametrize("kind", KINDS)
def test_saturated_motors_do_not_spin_up_a_robot_in_free_flight(kind, solver):
    """Regression for the joint-velocity clamp: with gravity off, far from the ground and every motor saturated against its velocity
    limit or its stops, the robot is a closed system -- its linear momentum must stay (nearly) constant, the joint velocities must hold
    their limits through the constraint rows (not through the 1.5 x safety clamp) and the base must not spin up."""
    cfg = _quiet(solver)
    cfg.init_state.pos = [0.0, 0.0, 10.0]
    cfg.sim.gravity = [0.0, 0.0, 0.0]
    cfg.termination.fall_down = False
    sim, model = _make(kind, cfg, 2)
    sim.reset_all()
    rs = np.random.RandomState(4)
    a0 = np.sign(rs.normal(0, 1, (2, 12))).astype(np.float32) * 30.0              # saturating targets -> joints run into vmax / the stops
    sim.step(a0)
    P0, _ = _total_momentum(sim, model)
    wmax = qdmax = 0.0
    for t in range(150):
        if t % 25 == 0:
            a0 = np.sign(rs.normal(0, 1, (2, 12))).astype(np.float32) * 30.0
        sim.step(a0)
        wmax = max(wmax, float(np.abs(sim.buf["root_states"][:, 10:13]).max()))
        qdmax = max(qdmax, float(np.abs(sim.buf["dof_state"].reshape(2, 12, 2)[:, :, 1]).max()))
    P1, _ = _total_momentum(sim, model)
    assert wmax < 15.0, wmax                   # the clamp version reached 54 rad/s here and kept accelerating
    # the 20 rad/s limit is held by the rows, not by the safety clamp at 30: 8 sweeps converge on it (measured 20.1); TGS relaxes every row
    # once per position iteration, 4 passes in all, and leaves more on these stiffly coupled rows (round 4: 25.4; with the final
    # velocity-level pass over the limit rows, lsim_config.tgs_limit_passes = 1, see the print)
    print(f"{kind} {solver}: fastest joint {qdmax:.2f} rad/s, base spin {wmax:.2f} rad/s, momentum change {np.abs(P1 - P0).max():.3f}")
    assert qdmax < (22.0 if solver == "pgs" else 27.0), qdmax
    assert np.abs(P1 - P0).max() < 1.5, (P0, P1)
    assert np.all(np.abs(sim.buf["root_states"][:, 2] - 10.0) < 1.0)


@pytest.mark.parametrize("solver", ["tgs", "pgs"])
@pytest.mark.parametrize("quiet", [True, False])
def test_emu_physics_matches_oracle(quiet, solver):
    """CPU leg of tests/test_gpu_parity.py::test_hip_physics_matches_oracle: the lane-emulated kernel sources (fp32, structured solver, the
    TGS form with its velocity rows) against the oracle (dense fp64) for both solvers, at the HIP test's tolerances."""
    import emu_binding
    N = 16
    cfg = quiet_cfg("aliengo") if quiet else C.TASKS["aliengo"][0]()
    cfg.sim.physx.solver_type = SOLVERS[solver]
    cfg.terrain.terrain_proportions = [1.0, 0.0, 0.0, 0.0]
    orc, lc, model, ter = make_oracle(cfg, N, seed=5)
    assert lc.solver_type == SOLVERS[solver]
    emu = emu_binding.EmuSim(lc, model, ter.heightsamples, ter.env_origins)
    orc.reset_all(); emu.reset_all()
    rs = np.random.RandomState(0)
    for t in range(12):
        a = rs.normal(0, 1, (N, 12)).astype(np.float32)
        orc.step(a); emu.step(a)
        np.testing.assert_array_equal(emu.buf["reset"], orc.buf["reset"], err_msg=f"step {t}")
        np.testing.assert_allclose(emu.buf["root_states"], orc.buf["root_states"], atol=3e-4, rtol=1e-4, err_msg=f"step {t}")
        np.testing.assert_allclose(emu.buf["dof_state"], orc.buf["dof_state"], atol=3e-3, rtol=1e-3, err_msg=f"step {t}")
        np.testing.assert_allclose(emu.buf["contact_forces"], orc.buf["contact_forces"], atol=0.15, rtol=2e-3, err_msg=f"step {t}")
        np.testing.assert_allclose(emu.buf["rew"], orc.buf["rew"], atol=1e-5, rtol=1e-4, err_msg=f"step {t}")
        np.testing.assert_allclose(emu.buf["obs"], orc.buf["obs"], atol=3e-4, rtol=1e-4, err_msg=f"step {t}")


@pytest.mark.parametrize("solver", ["tgs", "pgs"])
def test_emu_stairs_wall_contacts_match_oracle(solver):
    """Stairs (slope-corrected mesh with vertical risers, TER:72-75), CPU leg of tests/test_gpu_parity.py's stairs test: the
    lane-emulated kernel sources (fp32) against the oracle (fp64, independent triangle query), states re-synchronised every step."""
    import emu_binding
    N = 16
    cfg = C.TASKS["aliengo_stairs"][0]()
    cfg.sim.physx.solver_type = SOLVERS[solver]
    cfg.terrain.terrain_proportions = [0, 0, 0, 0, 0.5, 0.5]
    cfg.domain_rand.base_init_pos_range = dict(x=[-3.0, 3.0], y=[-3.0, 3.0], z=[0.0, 0.3])
    orc, lc, model, ter = make_oracle(cfg, N, seed=3)
    emu = emu_binding.EmuSim(lc, model, ter.heightsamples, ter.env_origins)
    from test_gpu_parity import _dzmax_numpy           # the packed words' "nothing above this height" byte against a numpy restatement
    np.testing.assert_array_equal((np.array(emu.buf["terrain_mesh"]).view(np.uint32) >> 24).astype(np.int64), _dzmax_numpy(ter.heightsamples))
    orc.reset_all(); emu.reset_all()
    rs = np.random.RandomState(0)
    ok = tot = walls = 0
    for t in range(30):
        a = rs.normal(0, 1, (N, 12)).astype(np.float32)
        for k in ("root_states", "dof_state", "commands", "last_actions", "episode_length", "terrain_levels", "env_origins", "kp_factors",
                  "kd_factors", "friction", "pending_force", "feet_air_time", "last_contacts", "episode_sums", "obs", "last_dof_vel"):
            emu.buf[k][...] = orc.buf[k]
        orc.step(a); emu.step(a)
        same = emu.buf["reset"] == orc.buf["reset"]
        e_root = np.abs(emu.buf["root_states"] - orc.buf["root_states"]).max(1)
        e_dof = np.abs(emu.buf["dof_state"] - orc.buf["dof_state"]).reshape(N, -1).max(1)
        ok += int((same & (e_root < 2e-3) & (e_dof < 2e-2)).sum()); tot += N
        feet = orc.buf["contact_forces"][:, [4, 8, 12, 16], :]
        walls += int((np.linalg.norm(feet[..., :2], axis=-1) > 2.0 * np.abs(feet[..., 2]) + 1.0).sum())
    print(f"stairs (emulator): {ok} of {tot} env-steps within tolerance")
    # measured: 479 of 480 with 3 collision points per calf (rounds 2-5); 475 of 480 with the 6 per calf of round 6 (robots/common.py: twice the calf points
    # brush treads and risers in this dropped-robot scenario; the five misses are contact-sequence forks of fp32 vs fp64 with EQUAL contact counts on both sides)
    assert ok >= 0.98 * tot, (ok, tot)
    assert walls > 0, "the scenario must exercise riser (mostly horizontal) foot contacts"


@pytest.mark.parametrize("N", [1, 13])
def test_emu_plane_ground_ragged_sizes_match_oracle(N):
    """mesh_type 'plane' (LR:1069-1078, no height grid, no terrain curriculum) at sizes that do not fill a block group."""
    import emu_binding
    cfg = C.TASKS["aliengo"][0]()
    cfg.terrain.mesh_type = "plane"
    cfg.terrain.measure_heights = True
    orc, lc, model, ter = make_oracle(cfg, N, seed=11)
    assert lc.mesh_type == 0 and lc.terrain_curriculum == 0
    emu = emu_binding.EmuSim(lc, model, ter.heightsamples, ter.env_origins)
    orc.reset_all(); emu.reset_all()
    rs = np.random.RandomState(2)
    for t in range(10):
        a = rs.normal(0, 1, (N, 12)).astype(np.float32)
        orc.step(a); emu.step(a)
        np.testing.assert_array_equal(emu.buf["reset"], orc.buf["reset"], err_msg=f"step {t}")
        np.testing.assert_allclose(emu.buf["root_states"], orc.buf["root_states"], atol=2e-3, rtol=1e-3, err_msg=f"step {t}")
        np.testing.assert_allclose(emu.buf["obs"], orc.buf["obs"], atol=5e-3, rtol=1e-3, err_msg=f"step {t}")
        np.testing.assert_allclose(emu.buf["rew"], orc.buf["rew"], atol=1e-3, rtol=1e-3, err_msg=f"step {t}")
    assert np.all(emu.buf["measured_heights"] == 0.0)     # LR:1478-1479: zeros on a plane


@pytest.mark.parametrize("robot,total_mass,steps", [("go1", 11.31, 60), ("go2", 15.017, 160)])   # Go2's softer gains (Kp 20, Kd 0.5) settle in ~2.5 s
def test_go1_table_emu_matches_oracle_and_stands(robot, total_mass, steps):
    """second robots (robots/tables/go1.json / go2.json, tasks "go1" / "go2"): the lane-emulated kernels agree with the oracle, and a
    zero-action robot dropped from its init height settles on its feet (total foot force ~ m g, no termination)"""
    import emu_binding
    cfg = quiet_cfg(robot)
    orc, lc, model, ter = make_oracle(cfg, 4, seed=2)
    mass = sum(model.bodies[i].mass for i in range(17))
    assert abs(mass - total_mass) < 5e-3
    emu = emu_binding.EmuSim(lc, model, ter.heightsamples, ter.env_origins)
    orc.reset_all(); emu.reset_all()
    a = np.zeros((4, 12), np.float32)
    for t in range(steps):
        orc.step(a); emu.step(a)
        np.testing.assert_array_equal(emu.buf["reset"], orc.buf["reset"], err_msg=f"step {t}")
        np.testing.assert_allclose(emu.buf["root_states"], orc.buf["root_states"], atol=3e-3, rtol=1e-3, err_msg=f"step {t}")
    assert orc.buf["reset"].sum() == 0
    fz = orc.buf["contact_forces"][:, [4, 8, 12, 16], 2].sum(1)
    np.testing.assert_allclose(fz, mass * G, rtol=0.05)
    assert np.all(orc.buf["root_states"][:, 2] > 0.2) and np.all(orc.buf["root_states"][:, 2] < 0.45)


@both_solvers
@pytest.mark.parametrize("kind", KINDS)
def test_foot_contact_forces_stay_in_the_friction_pyramid(kind, solver):
    """Flat ground, a robot thrashing under random actions: every foot force pushes (f_z >= 0) and its tangential components stay inside
    the solver's friction pyramid |f_x|, |f_y| <= mu f_z with mu = average(terrain friction, robot friction); feet do not
    sink into the ground by more than the contact offset."""
    cfg = _quiet(solver)
    cfg.terrain.mesh_type = "plane"
    cfg.init_state.pos = [0.0, 0.0, 0.40]
    sim, model = _make(kind, cfg, 4)
    sim.reset_all()
    mu = 0.5 * (float(cfg.terrain.static_friction) + float(sim.buf["friction"][0]))
    rng = np.random.default_rng(3)
    foot_r = [model.points[i].radius for i in range(model.num_collision_points) if model.points[i].body in (4, 8, 12, 16)][0]
    seen = 0
    for _ in range(60):
        sim.step(rng.normal(0.0, 1.0, (4, 12)).astype(np.float32))
        f = sim.buf["contact_forces"][:, [4, 8, 12, 16], :].astype(np.float64)
        load = f[..., 2] > 1.0
        seen += int(load.sum())
        assert (f[..., 2] > -1e-3).all()
        assert (np.abs(f[..., 0])[load] <= mu * f[..., 2][load] * (1 + 1e-3) + 1e-2).all()
        assert (np.abs(f[..., 1])[load] <= mu * f[..., 2][load] * (1 + 1e-3) + 1e-2).all()
        z = sim.buf["rigid_body_states"][:, [4, 8, 12, 16], 2]
        assert (z > foot_r - 0.02).all(), z.min()
    assert seen > 100           # the feet were on the ground most of the time


def _quat_R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


@pytest.mark.parametrize("convention", ["com", "origin"])
@pytest.mark.parametrize("kind", KINDS)
def test_published_linear_velocity_is_the_centre_of_mass_velocity(kind, convention, monkeypatch):
    """P5 (LR:929-941, LR:198-199): PhysX's state tensors carry the linear velocity of each body's CENTRE OF MASS.  Robots tumbling in
    free flight at |w| ~ 5 rad/s with the base COM displaced by the full +-5 cm payload range (LR:1025-1028): the published linear velocity of
    the root and of all 17 bodies must equal d/dt of (link position + R c) -- a kinematic identity of the published POSITIONS, evaluated as
    a central difference over two steps (corrected for the half-sub-step lead of a semi-implicit Euler position update) -- and must differ from the link origin's velocity by w x R c, ~0.4 m/s here.  With lin_vel_at_com = 0 (LSIM_LIN_VEL=origin: the
    convention of rounds 1-4) the same tensors hold the link-origin velocity instead.  base_lin_vel (LR:198) is the published root value."""
    monkeypatch.setenv("LSIM_LIN_VEL", convention)
    cfg = _quiet("tgs")
    cfg.init_state.pos = [0.0, 0.0, 6.0]
    cfg.termination.fall_down = False
    cfg.domain_rand.randomize_com_displacement = True
    cfg.domain_rand.com_displacement_range = [0.05, 0.05]            # every env at the corner of the range
    cfg.domain_rand.base_init_vel_range = dict(x=[0.4, 0.4], y=[-0.3, -0.3], z=[0.5, 0.5], roll=[2.0, 2.0], pitch=[-3.0, -3.0], yaw=[4.0, 4.0])
    N = 4
    sim, model = _make(kind, cfg, N)
    assert bool(sim.cfg.lin_vel_at_com) == (convention == "com")
    sim.reset_all()
    a = np.zeros((N, 12), np.float32)
    hist = []
    for _ in range(7):
        sim.step(a)
        hist.append((sim.buf["rigid_body_states"].astype(np.float64).reshape(N, 17, 13).copy(), sim.buf["root_states"].astype(np.float64).copy(),
                     sim.buf["base_lin_vel"].astype(np.float64).copy()))
    assert sim.buf["reset"].sum() == 0 and np.abs(sim.buf["contact_forces"]).sum() == 0
    comd = sim.buf["com_displacement"].astype(np.float64)
    np.testing.assert_allclose(comd, 0.05, atol=1e-6)
    step_dt, sub_dt = 4 * 0.005, 0.005

    def point(bs, e, i, at_com):
        c = np.array(model.bodies[i].com, dtype=np.float64) + (comd[e] if i == 0 else 0.0)
        return bs[e, i, 0:3] + (_quat_R(bs[e, i, 3:7]) @ c if at_com else 0.0)
    worst_own, worst_base, worst_other, sep = 0.0, 0.0, np.inf, 0.0
    for t in range(1, 6):
        for e in range(N):
            for i in range(17):
                v_pub = hist[t][0][e, i, 7:10]
                # semi-implicit Euler: a position increment uses the velocity AFTER the sub-step's kick, so the central difference of the
                # positions runs half a sub-step ahead of the published velocity: a dt / 2, a = the point's own acceleration (gravity +
                # centripetal, up to 20 m/s^2 at the feet), estimated from the published velocities themselves
                lag = 0.5 * sub_dt * (hist[t + 1][0][e, i, 7:10] - hist[t - 1][0][e, i, 7:10]) / (2 * step_dt)
                d_com = (point(hist[t + 1][0], e, i, True) - point(hist[t - 1][0], e, i, True)) / (2 * step_dt) - lag
                d_org = (point(hist[t + 1][0], e, i, False) - point(hist[t - 1][0], e, i, False)) / (2 * step_dt) - lag
                own, other = (d_com, d_org) if convention == "com" else (d_org, d_com)
                worst_own = max(worst_own, float(np.abs(v_pub - own).max()) / (0.015 + 0.012 * float(np.linalg.norm(v_pub))))
                if i == 0:
                    worst_base = max(worst_base, float(np.abs(v_pub - own).max()))
                    worst_other = min(worst_other, float(np.linalg.norm(v_pub - other)))
                    sep = max(sep, float(np.linalg.norm(d_com - d_org)))
            np.testing.assert_array_equal(hist[t][1][e, 7:10], hist[t][0][e, 0, 7:10])              # the root tensor's velocity is row 0's
            want = _quat_R(hist[t][1][e, 3:7]).T @ hist[t][1][e, 7:10]                              # LR:198: quat_rotate_inverse(base_quat, root_states[:, 7:10])
            np.testing.assert_allclose(hist[t][2][e], want, atol=2e-6)
    print(f"{kind} {convention}: base velocity vs d/dt of its own point: worst {worst_base:.4f} m/s; all bodies, in units of their bar: {worst_own:.2f}; "
          f"base vs the OTHER point: at least {worst_other:.3f} m/s")
    assert sep > 0.3                    # the two conventions are far apart in this motion ...
    assert worst_base < 0.02            # ... the published base velocity follows its own point (first-order integrator: ~0.01 m/s at this spin) ...
    assert worst_own < 1.0              # ... so does every body's (bar 0.015 m/s + 1.2 % of its speed: the feet move at 3 m/s on a 0.4 m arm)
    assert worst_other > 0.2            # ... and none follows the other point
