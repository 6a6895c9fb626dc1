"""Deterministic synthetic planning problems.

Random obstacle worlds and initial states for the Kinova Gen3 7-DOF arm, in the input format of the
reference planner: q0, qd0, qdd0, q_des (7 doubles each) and obstacles as column-major Z=[c g1 g2 g3]
(12 doubles, KSI/uarmtd_planner.m:158-185; boxes are [c, diag(s/2)] as in
SIM/worlds/obstacles/box_obstacle_zonotope.m:21-26).
"""
import numpy as np

# RT/KinovaWithoutGripperInfo.h:76-79
STATE_LB = np.array([-1000.0, -2.41, -1000.0, -2.66, -1000.0, -2.23, -1000.0])
STATE_UB = -STATE_LB
SPEED = np.array([1.3963, 1.3963, 1.3963, 1.3963, 1.2218, 1.2218, 1.2218])


def reference_sample_problem():
    """The sample input the reference keeps as a comment in its main program (RT/armour_main.cu:18-33): the one
    concrete planning problem it ships (10 box obstacles; a known input without a recorded answer)."""
    five = np.array([
        [-0.28239, -0.33281, 0.88069, 0.069825, 0, 0, 0, 0.09508, 0, 0, 0, 0.016624],
        [-0.19033, 0.035391, 1.3032, 0.11024, 0, 0, 0, 0.025188, 0, 0, 0, 0.014342],
        [0.67593, -0.085841, 0.43572, 0.17408, 0, 0, 0, 0.07951, 0, 0, 0, 0.18012],
        [0.75382, 0.51895, 0.4731, 0.030969, 0, 0, 0, 0.22312, 0, 0, 0, 0.22981],
        [0.75382, 0.51895, 0.4731, 0.030969, 0, 0, 0, 0.22312, 0, 0, 0, 0.22981]])
    return dict(q0=np.array([0.6543, -0.0876, -0.4837, -1.2278, -1.5735, -1.0720, 0]), qd0=np.zeros(7), qdd0=np.zeros(7),
                q_des=np.array([0.6831, 0.009488, -0.2471, -0.9777, -1.414, -0.9958, 0]), obstacles=np.vstack([five, five]))


def random_problem(seed, num_obstacles):
    """One world: dict(q0, qd0, qdd0, q_des [7], obstacles [O,12])."""
    rng = np.random.default_rng(seed)
    cont = np.abs(STATE_LB) >= 1000.0
    q0 = np.where(cont, rng.uniform(-np.pi, np.pi, 7), rng.uniform(STATE_LB + 0.3, STATE_UB - 0.3))
    qd0 = rng.uniform(-0.5, 0.5, 7) * SPEED
    qdd0 = rng.uniform(-1.0, 1.0, 7)
    q_des = q0 + rng.uniform(-np.pi / 8, np.pi / 8, 7)
    c = rng.uniform([-0.8, -0.8, 0.05], [0.8, 0.8, 1.2], (num_obstacles, 3))
    s = rng.uniform(0.01, 0.5, (num_obstacles, 3))
    obs = np.zeros((num_obstacles, 12))
    obs[:, 0:3] = c
    obs[:, 3] = s[:, 0] / 2
    obs[:, 7] = s[:, 1] / 2
    obs[:, 11] = s[:, 2] / 2
    return dict(q0=q0, qd0=qd0, qdd0=qdd0, q_des=q_des, obstacles=obs)


def random_fetch_problem(seed, num_obstacles):
    """One world for the Fetch preset (CMP/FetchInfo.h:86-89: joint and speed limits): a random state inside the joint limits and
    `num_obstacles` random boxes -- BASELINE configs[4] uses 100."""
    rng = np.random.default_rng(seed)
    lb = np.array([-1.6056, -1.221, -np.pi, -2.251, -np.pi, -2.16, -np.pi]) + 0.3
    ub = np.array([1.6056, 1.518, np.pi, 2.251, np.pi, 2.16, np.pi]) - 0.3
    speed = np.array([1.256, 1.454, 1.571, 1.521, 1.571, 2.268, 2.268])
    q0 = rng.uniform(lb, ub)
    return dict(q0=q0, qd0=rng.uniform(-0.5, 0.5, 7) * speed, qdd0=rng.uniform(-1, 1, 7), q_des=q0 + rng.uniform(-0.3, 0.3, 7),
                obstacles=random_problem(seed, num_obstacles)["obstacles"])


def random_fetch8_problem(seed, num_obstacles):
    """One world for the 8-factor "Fetch 8-DOF" preset (include/armour_robot_fetch.h): the torso yaw inside its +-1 rad, the arm as
    random_fetch_problem, `num_obstacles` random boxes."""
    rng = np.random.default_rng(50_000 + seed)
    p = random_fetch_problem(seed, num_obstacles)
    q_t = rng.uniform(-0.7, 0.7)
    return dict(q0=np.concatenate([[q_t], p["q0"]]), qd0=np.concatenate([[rng.uniform(-0.5, 0.5) * 0.5], p["qd0"]]),
                qdd0=np.concatenate([[rng.uniform(-1, 1)], p["qdd0"]]), q_des=np.concatenate([[q_t + rng.uniform(-0.3, 0.3)], p["q_des"]]),
                obstacles=p["obstacles"])


def random_batch(first_seed, batch, num_obstacles):
    """`batch` worlds with seeds first_seed..first_seed+batch-1, stacked: q0 [B,7] ... obstacles [B,O,12]."""
    ps = [random_problem(first_seed + b, num_obstacles) for b in range(batch)]
    return {k: np.stack([p[k] for p in ps]) for k in ps[0]}


def random_k(seed, count, n=7):
    """Evaluation points k ~ U[-1,1]^n."""
    return np.random.default_rng(10_000 + seed).uniform(-1.0, 1.0, (count, n))


def load_scene_csv(path):
    """Saved random scene of the reference (kinova_src/saved_worlds/random/*.csv, load_saved_world.m:4-13):
    row 1 start, row 2 goal, row 3 NaN, rows 4.. = [cx cy cz sx sy sz NaN] boxes."""
    rows = np.genfromtxt(path, delimiter=",")
    q_start, q_goal = rows[0, :7], rows[1, :7]
    boxes = rows[3:, :6]
    obs = np.zeros((boxes.shape[0], 12))
    obs[:, 0:3] = boxes[:, 0:3]
    obs[:, 3] = boxes[:, 3] / 2
    obs[:, 7] = boxes[:, 4] / 2
    obs[:, 11] = boxes[:, 5] / 2
    return q_start, q_goal, obs


def synthetic_offline_jrs(qd0, T=100, t_plan=0.5, t_total=1.0):
    """Stand-in for the offline joint-reachable-set tables of the ARMTD comparison planner.

    The reference loads, per joint, the file JRS_<v>.mat of the initial-velocity bin nearest to qd0 (401 bins on
    [-pi, pi], KSI/uarmtd_planner.m:249-255) -- zonotopes CORA computed offline (CMP/offline_jrs/
    create_orig_offline_jrs.m) -- and writes six rows per joint into armtd.in (:277-312).  Those .mat files are
    not part of the reference checkout, so tests and probes use this closed-form enclosure with the same meaning and
    shapes: for time interval j, cos/sin of theta(t) = v t + ka t^2/2 (then the constant-deceleration stop), v in the
    bin, ka = k_range*k, as  centre + generator*k + [-radius, radius]  (first-order Taylor, Lagrange remainder).
    Returns (jrs [n,6,T], k_range [n]); real tables go through armour_set_problems_armtd unchanged.
    """
    qd0 = np.asarray(qd0, dtype=np.float64).ravel()
    n = qd0.size
    c_kvi = np.linspace(-np.pi, np.pi, 401)
    delta_kvi = (c_kvi[1] - c_kvi[0]) / 2
    dt = t_total / T
    jrs = np.zeros((n, 6, T))
    k_range = np.zeros(n)
    ts = t_total - t_plan
    for i in range(n):
        v = c_kvi[np.argmin(np.abs(qd0[i] - c_kvi))]
        ka = max(np.pi / 24, abs(v) / 3)          # delta_kai of create_orig_offline_jrs.m:56
        k_range[i] = ka
        for j in range(T):
            tm, half = (j + 0.5) * dt, 0.5 * dt
            if tm < t_plan:                          # theta = v t + ka t^2/2
                th, dth_dv, dth_da, rate = v * tm, tm, 0.5 * tm * tm, abs(v) + ka * tm
                da_dt = tm
            else:                                    # stop from (theta_p, w_p) with constant deceleration w_p/ts
                s = tm - t_plan
                f = s - 0.5 * s * s / ts             # theta = theta_p + w_p f(s)
                th, dth_dv = v * t_plan + v * f, t_plan + f
                dth_da = 0.5 * t_plan * t_plan + t_plan * f
                rate = (abs(v) + ka * t_plan) * (1 - s / ts + half / ts)
                da_dt = t_plan * (1 - s / ts + half / ts)
            a = dth_da * ka                          # k-dependent part of theta - theta_c
            e = dth_dv * delta_kvi + rate * half + da_dt * half * ka   # everything else
            rem = 0.5 * (abs(a) + e) ** 2
            jrs[i, 0, j], jrs[i, 1, j], jrs[i, 2, j] = np.cos(th), -np.sin(th) * a, abs(np.sin(th)) * e + rem
            jrs[i, 3, j], jrs[i, 4, j], jrs[i, 5, j] = np.sin(th), np.cos(th) * a, abs(np.cos(th)) * e + rem
    return jrs, k_range
