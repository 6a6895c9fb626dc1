"""The reference's own planning worlds as inputs of the hot path.

* the 100 saved random scenes of its system test (kinova_src/saved_worlds/random/scene_0NN_0MM.csv, read as
  KSI/kinova_scenarios/load_saved_world.m:4-13 reads them, run by kinova_src/scripts/kinova_run_100_worlds.m:115-186),
  committed as data under tests/golden/scenes/;
* its seven hard scenarios (KSI/kinova_scenarios/get_kinova_scenario_info.m:1-262), as the data table
  tests/golden/scenes/hard_scenarios.json (written by tests/golden/make_hard_scenarios.py).

What crosses into the hot path is the FIRST planning iteration of each world: the arm at rest at `start`
(q0 = start, qd0 = qdd0 = 0: uarmtd_planner.replan reads the agent's reference state, KSI/uarmtd_planner.m:88-91) and the
waypoint of the straight-line high-level planner, q_des = q0 + lookahead * dir / |dir| with the direction wrapped on the
four continuous joints (SIM/planners/high_level_planners/robot_arm_straight_line_HLP.m:43-56).  Lookahead: the planner's
default 1 rad for the saved scenes (kinova_run_100_worlds.m:137-143 does not pass its `lookahead_distance`, so
robot_arm_generic_planner.m:21 holds) and 0.1 for the hard scenarios (kinova_run_hard_scenarios.m:58,147).  The simulator,
the agent and the later iterations (which depend on the executed trajectory) are out of scope.

A handle builds B problems with ONE obstacle count; worlds with fewer boxes are padded with `FAR_BOX`, a 1 cm box 50 m
below the base: its rows are never active (g < -40 m for every k), so optimum and verdict are those of the unpadded world
(tests/test_reference_scenes.py checks that on the device).
"""
import glob
import json
import os

import numpy as np

from .worlds import STATE_LB, load_scene_csv

SCENE_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "scenes")
FAR_BOX = np.array([0.0, 0.0, -50.0, 0.005, 0, 0, 0, 0.005, 0, 0, 0, 0.005])
CONTINUOUS = np.abs(STATE_LB) >= 1000.0          # joints 1,3,5,7 of the Kinova (RT/KinovaWithoutGripperInfo.h:76-77)


def angdiff(a, b):
    """MATLAB angdiff(a, b): b - a wrapped to [-pi, pi]."""
    d = np.asarray(b, dtype=np.float64) - np.asarray(a, dtype=np.float64)
    return (d + np.pi) % (2 * np.pi) - np.pi


def straight_line_waypoint(q_cur, q_goal, lookahead):
    """robot_arm_straight_line_HLP.get_waypoint (:43-56): no clipping at the goal, direction wrapped on continuous joints."""
    q_cur, q_goal = np.asarray(q_cur, dtype=np.float64), np.asarray(q_goal, dtype=np.float64)
    d = q_goal - q_cur
    d[CONTINUOUS] = angdiff(q_cur[CONTINUOUS], q_goal[CONTINUOUS])
    return q_cur + lookahead * d / np.linalg.norm(d)


def boxes_to_obstacles(boxes):
    """[cx cy cz sx sy sz] rows -> column-major Z = [c, diag(s/2)] (12 numbers, KSI/uarmtd_planner.m:178)."""
    boxes = np.asarray(boxes, dtype=np.float64).reshape(-1, 6)
    obs = np.zeros((boxes.shape[0], 12))
    obs[:, 0:3] = boxes[:, 0:3]
    obs[:, 3], obs[:, 7], obs[:, 11] = boxes[:, 3] / 2, boxes[:, 4] / 2, boxes[:, 5] / 2
    return obs


def first_iteration(start, goal, obstacles, lookahead):
    start = np.asarray(start, dtype=np.float64)
    return dict(q0=start.copy(), qd0=np.zeros(7), qdd0=np.zeros(7), q_des=straight_line_waypoint(start, goal, lookahead),
                obstacles=np.asarray(obstacles, dtype=np.float64), goal=np.asarray(goal, dtype=np.float64))


def saved_scene_files():
    files = sorted(glob.glob(os.path.join(SCENE_DIR, "scene_*.csv")))
    if not files:
        raise FileNotFoundError(f"the reference's saved scenes are fixtures of this repository's tests: none under {SCENE_DIR}")
    return files


def saved_scenes(lookahead=1.0):
    """[(name, problem)] of the 100 saved random scenes, first planning iteration."""
    out = []
    for path in saved_scene_files():
        start, goal, obs = load_scene_csv(path)
        out.append((os.path.basename(path)[:-4], first_iteration(start, goal, obs, lookahead)))
    return out


def hard_scenarios(lookahead=0.1):
    """[(name, problem)] of the seven hard scenarios, first planning iteration."""
    with open(os.path.join(SCENE_DIR, "hard_scenarios.json")) as f:
        table = json.load(f)["scenarios"]
    return [("hard_%d_%s" % (s["scenario"], s["name"].replace(" ", "_")),
             first_iteration(s["start"], s["goal"], boxes_to_obstacles(s["boxes"]), lookahead)) for s in table]


def reference_worlds():
    """The 107 worlds the reference itself holds: saved scenes, then hard scenarios."""
    return saved_scenes() + hard_scenarios()


def pad_obstacles(obstacles, count):
    """`count` obstacle rows: the world's own, then FAR_BOX."""
    obstacles = np.asarray(obstacles, dtype=np.float64).reshape(-1, 12)
    if obstacles.shape[0] > count:
        raise ValueError("world has %d obstacles, more than %d" % (obstacles.shape[0], count))
    return np.vstack([obstacles, np.tile(FAR_BOX, (count - obstacles.shape[0], 1))])


def as_batch(worlds, num_obstacles=None):
    """Stack [(name, problem)] into the arrays armour_set_problems takes: q0 [B,7] ... obstacles [B,O,12], with O = the largest
    world's count unless given; `n_obstacles` [B] keeps every world's own count."""
    ps = [p for _, p in worlds]
    own = np.array([p["obstacles"].shape[0] for p in ps])
    O = int(own.max()) if num_obstacles is None else int(num_obstacles)
    out = {k: np.stack([p[k] for p in ps]) for k in ("q0", "qd0", "qdd0", "q_des")}
    out["obstacles"] = np.stack([pad_obstacles(p["obstacles"], O) for p in ps])
    out["n_obstacles"] = own
    out["names"] = [n for n, _ in worlds]
    return out
