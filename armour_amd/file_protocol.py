"""The reference's process boundary: text files under `buffer/` (transport A).

Input  `armour.in`  (KSI/uarmtd_planner.m:158-185, parsed by RT/armour_main.cu:53-76):
    4 lines of n numbers "%.10f " (q0, qd0, qdd0, q_des), one line with nObs, then nObs lines of 12 numbers
    = column-major Z = [c g1 g2 g3] of each obstacle zonotope.
Outputs (RT/armour_main.cu:312-372, read back by KSI/uarmtd_planner.m:192-230):
    armour.out                          n lines k_opt (setprecision(10)) or a single -1, then the time in ms
    armour_joint_position_center.out    T*J rows x 3   sliced link centres
    armour_joint_position_radius.out    T*J*3 rows x 6 link independent generators
    armour_control_input_radius.out     T rows x n     torque radius
    armour_constraints.out              m rows (setprecision(6))
C++ streams with setprecision(p) print like "%.{p}g"; that is what the writers below use.
"""
import os

import numpy as np

IN_NAME = "armour.in"
OUT_NAMES = ("armour.out", "armour_joint_position_center.out", "armour_joint_position_radius.out",
             "armour_control_input_radius.out", "armour_constraints.out")


def write_armour_in(path, q0, qd0, qdd0, q_des, obstacles):
    """What uarmtd_planner.replan writes before spawning the planner process."""
    obs = np.asarray(obstacles, dtype=np.float64).reshape(-1, 12)
    with open(path, "w") as f:
        for v in (q0, qd0, qdd0, q_des):
            f.write("".join("%.10f " % x for x in np.asarray(v, dtype=np.float64)) + "\n")
        f.write("%d\n" % obs.shape[0])
        for row in obs:
            f.write("".join("%.10f " % x for x in row) + "\n")


ARMTD_IN_NAME = "armtd.in"


def write_armtd_in(path, q0, qd0, q_des, jrs, k_range, obstacles):
    """Input of the ARMTD comparison planner as uarmtd_planner.replan writes it (KSI/uarmtd_planner.m:257-327, parsed
    by CMP/armtd_main.cu:57-110): q0, qd0, q_des; per joint six lines of T numbers (centre, k-generator, radius of cos,
    then of sin) and a line with its k_range; nObs; nObs lines of 12 numbers."""
    obs = np.asarray(obstacles, dtype=np.float64).reshape(-1, 12)
    jrs = np.asarray(jrs, dtype=np.float64)
    with open(path, "w") as f:
        for v in (q0, qd0, q_des):
            f.write("".join("%.10f " % x for x in np.asarray(v, dtype=np.float64)) + "\n")
        for i in range(jrs.shape[0]):
            for row in jrs[i]:
                f.write("".join("%.10f " % x for x in row) + "\n")
            f.write("%.10f\n " % k_range[i])
        f.write("%d\n" % obs.shape[0])
        for row in obs:
            f.write("".join("%.10f " % x for x in row) + "\n")


def parse_armour_in(path, n=7, max_obstacles=None):
    """Whitespace-separated token stream exactly as the `>>` loop of RT/armour_main.cu:53-76 reads it.
    Raises ValueError where the reference writes -1 and throws (missing file, bad obstacle count)."""
    if not os.path.exists(path):
        raise ValueError("Error reading input files")
    tok = open(path).read().split()
    need = 4 * n + 1
    if len(tok) < need:
        raise ValueError("input file too short")
    vals = [float(t) for t in tok[:4 * n]]
    q0, qd0, qdd0, q_des = (np.array(vals[i * n:(i + 1) * n]) for i in range(4))
    nobs = int(float(tok[4 * n]))
    if nobs < 0 or (max_obstacles is not None and nobs > max_obstacles):
        raise ValueError("Number of obstacles larger than MAX_OBSTACLE_NUM")
    if len(tok) < need + 12 * nobs:
        raise ValueError("input file too short for %d obstacles" % nobs)
    obs = np.array([float(t) for t in tok[need:need + 12 * nobs]]).reshape(nobs, 12)
    return dict(q0=q0, qd0=qd0, qdd0=qdd0, q_des=q_des, obstacles=obs)


def _g(x, p):
    return "%.*g" % (p, x)


def write_outputs(dirname, k_opt, time_ms, link_centers, link_gens, torque_radius, constraints):
    """k_opt: [n] or None (infeasible -> -1); link_centers [T,J,3]; link_gens [T,J,3,6]; torque_radius [n,T];
    constraints [m]."""
    with open(os.path.join(dirname, OUT_NAMES[0]), "w") as f:
        if k_opt is None:
            f.write("-1\n")
        else:
            for v in k_opt:
                f.write(_g(v, 10) + "\n")
        f.write(_g(time_ms, 10))
    T, J = link_centers.shape[:2]
    with open(os.path.join(dirname, OUT_NAMES[1]), "w") as f:
        for t in range(T):
            for j in range(J):
                f.write("".join(_g(v, 10) + " " for v in link_centers[t, j]) + "\n")
    with open(os.path.join(dirname, OUT_NAMES[2]), "w") as f:
        for t in range(T):
            for j in range(J):
                for r in range(3):
                    f.write("".join(_g(v, 10) + " " for v in link_gens[t, j, r]) + "\n")
    with open(os.path.join(dirname, OUT_NAMES[3]), "w") as f:
        for t in range(T):
            f.write("".join(_g(v, 10) + " " for v in torque_radius[:, t]) + "\n")
    with open(os.path.join(dirname, OUT_NAMES[4]), "w") as f:
        for v in constraints:
            f.write(_g(v, 6) + "\n")


def read_armour_out(path, n=7):
    """uarmtd_planner.m:192-208: a single leading -1 (or fewer than n+1 numbers) means 'no plan'.
    Returns (k_opt or None, time_ms)."""
    vals = np.loadtxt(path, ndmin=1)
    if vals.size == n + 1:
        return vals[:n], float(vals[n])
    return None, float(vals[-1])


def run_planning_iteration(nlp, dirname, k=None):
    """One pass over the file protocol with the device library: read armour.in, build the reach sets, solve the NLP
    (armour_solve; or evaluate at a caller-supplied `k`) and write the five output files -- what the `armour_main`
    binary (armour_amd/csrc/cli_common.h) does natively."""
    import time
    t0 = time.perf_counter()
    p = parse_armour_in(os.path.join(dirname, IN_NAME), n=nlp.n)
    nlp.set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
    if k is None:
        sol = nlp.solve()[0]          # OptimizeTNLP + finalize_solution on the device callbacks
        x, feasible = sol["k_opt"], sol["feasible"]
        g = nlp.eval_g(x)
    else:
        x = np.asarray(k, dtype=np.float64)
        g = nlp.eval_g(x)
        feasible = bool(nlp.finalize_solution(g)[0])
    cen = nlp.link_centers(x)[0]
    ms = (time.perf_counter() - t0) * 1e3
    write_outputs(dirname, x if feasible else None, ms, cen, nlp.link_generators()[0], nlp.torque_radius()[0], g[0])
    return feasible


class ResidentPlanner:
    """`armour_main --serve <dir>` as a child process: the planner stays resident (GPU context, code objects, handles) and
    the per-iteration executables `armour_main <dir>` / `armtd_main <dir>` only forward to it over <dir>/armour.sock
    (armour_amd/csrc/cli_common.h).  Use as a context manager around a simulation that spawns the executables."""

    def __init__(self, dirname, T_armour=128, T_armtd=100, exe=None, start_timeout_s=300.0, log_path=None):
        import subprocess
        import time
        here = os.path.dirname(os.path.abspath(__file__))
        self.exe = exe or os.path.join(here, "bin", "armour_main")
        self.dir = str(dirname)
        self.sock = os.path.join(self.dir, "armour.sock")
        if os.path.exists(self.sock):
            os.remove(self.sock)
        # The planner prints two or three "HIP & C++: ..." lines per served iteration, as the reference's program does.
        # They go to a FILE, never to a pipe nobody reads: a pipe fills after a few hundred iterations (64 KB), the
        # planner then blocks in fflush(stdout) and every forwarding armour_main waits on the socket forever.
        self.log_path = log_path or os.path.join(self.dir, "armour_resident.log")
        self._log = open(self.log_path, "w")
        self.proc = subprocess.Popen([self.exe, "--serve", self.dir, str(T_armour), str(T_armtd)], stdout=self._log, stderr=subprocess.STDOUT)
        t0 = time.time()
        while not os.path.exists(self.sock):
            if self.proc.poll() is not None:
                raise RuntimeError("resident planner exited: " + self.log())
            if time.time() - t0 > start_timeout_s:
                self.proc.kill()       # the front end; its worker child gets SIGTERM from the kernel (PR_SET_PDEATHSIG)
                self.proc.wait()
                raise RuntimeError("resident planner did not come up: " + self.log())
            time.sleep(0.02)

    def log(self):
        """what the planner has printed so far"""
        try:
            self._log.flush()
            with open(self.log_path) as f:
                return f.read()
        except OSError:
            return ""

    def stop(self):
        import subprocess
        if self.proc.poll() is None:
            subprocess.run([self.exe, "--quit", self.dir], timeout=60)
            try:
                self.proc.wait(timeout=60)
            except subprocess.TimeoutExpired:
                self.proc.kill()   # this exact child only; the worker behind it is told by the kernel (PR_SET_PDEATHSIG)
                self.proc.wait()
        if not self._log.closed:
            self._log.close()
        return self.proc.returncode

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.stop()
