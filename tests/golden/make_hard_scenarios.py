"""The seven hard scenarios of the reference as a data table (tests/golden/scenes/hard_scenarios.json).

The reference holds them as MATLAB statements (KSI/kinova_scenarios/get_kinova_scenario_info.m:1-262: start, goal and
box obstacles per case; make_shelf_obstacle.m:1-70 for the shelves of case 4; the axis swap into the Kinova's frame,
get_kinova_scenario_info.m:254-261).  This script restates that arithmetic once and writes the resulting numbers --
start [7], goal [7], boxes [[cx cy cz sx sy sz]] -- so that tests and bench.py read plain data.  Run from anywhere:
python tests/golden/make_hard_scenarios.py
"""
import json
import os

import numpy as np

PI = np.pi


def box(center, side_lengths):
    return (np.asarray(center, dtype=float).ravel(), np.asarray(side_lengths, dtype=float).ravel())


def to_kinova(boxes):
    """fetch_obstacles_to_kinova_obstacles (get_kinova_scenario_info.m:254-261)."""
    out = []
    for c, s in boxes:
        out.append((np.array([c[2] - 0.8, c[1], c[0] + 0.25]), np.array([s[2], s[1], s[0]])))
    return out


def shelf(center, height, width, depth, n_shelves, min_h, max_h, direction):
    """make_shelf_obstacle.m:24-69: two sides and n_shelves boards, 1 cm thick."""
    center = np.asarray(center, dtype=float)
    th = 0.01
    if direction == 1:
        c1, c2 = center + [0, -width / 2, 0], center + [0, width / 2, 0]
        side, board = [depth, th, height], [depth, width, th]
    else:
        c1, c2 = center + [-width / 2, 0, 0], center + [width / 2, 0, 0]
        side, board = [th, depth, height], [width, depth, th]
    out = [box(c1, side), box(c2, side)]
    for h in np.linspace(min_h, max_h, n_shelves):
        out.append(box([center[0], center[1], h], board))
    return out


def scenario(case):
    if case == 1:    # table
        return "table", [0, 0.5, 0, -0.5, 0, 0, 0], [0, -0.5, 0, 0.5, 0, 0, 0], to_kinova([box([1.1, 0, 0.8], [1, 4, 0.01])])
    if case == 2:    # wall / doorway
        return "wall", [PI / 2, 0.5, 0, 0, 0, 0, 0], [-PI / 2, 0.5, 0, 0.5, 0, 0, 0], to_kinova([box([1.1, 0, 0.8], [1, 0.01, 4])])
    if case == 3:    # posts
        return ("posts", [PI / 2, PI / 4, 0, 0, 0, 0, 0], [0.15, -0.75, 0.2, 0.4, 0.3, 0.2, 0],
                to_kinova([box([0.8, -0.25, 2], [0.05, 0.05, 4]), box([0.4, 0.25, 2], [0.05, 0.05, 4])]))
    if case == 4:    # shelves
        s1 = shelf([1.1, 0, 0.7], 1.4, 1.2, 0.8, 3, 0.3, 1.3, 1)
        s2 = shelf([0, 1.1, 0.7], 1.4, 1.2, 0.8, 3, 0.3, 1.3, 2)
        return "shelves", [0, -0.5, 0, 0.5, 0, 0, 0], [-PI / 2, PI / 2, -PI / 2, 0.5, 0, 0, 0], to_kinova(s1 + s2)
    if case == 5:    # inside box
        L = np.array([0.4, 0.4, 0.66])
        C = np.array([0.45, 0, L[2] / 2])
        bs = [box([C[0], C[1] + L[1] / 2, C[2]], [L[0], 0.01, L[2]]),
              box([C[0] - L[0] / 2, C[1], C[2]], [0.01, L[1], L[2]]),
              box([C[0], C[1] - L[1] / 2, C[2]], [L[0], 0.01, L[2]]),
              box([C[0] + L[0] / 2, C[1], C[2]], [0.01, L[1], L[2]])]
        return "inside box", [0, 0, 0, -PI / 2, 0, 0, 0], [0.15, 0.1, 0.2, 0.4, 0.3, 0.2, 0], to_kinova(bs)
    if case == 6:    # sink to cupboard
        cc = np.array([0.6, 0, 0.6])
        cl, cw, sw, sd = 0.5, 2.0, 0.5, 0.3
        cb = np.array([0.6, -0.55, 1.4])
        bl, bw, bd = cl, 0.5, 0.5
        bs = [box(cc + [0, sw / 2 + cw / 2, 0], [cl, cw, 0.01]),
              box(cc + [0, -sw / 2 - cw / 2, 0], [cl, cw, 0.01]),
              box(cc + [0, sw / 2, -sd / 2], [sw, 0.01, sd]),
              box(cc + [0, -sw / 2, -sd / 2], [sw, 0.01, sd]),
              box(cc + [sw / 2, 0, -sd / 2], [0.01, sw, sd]),
              box(cc + [-sw / 2, 0, -sd / 2], [0.01, sw, sd]),
              box(cc + [0, 0, -sd], [sw, sw, 0.01]),
              box(cb + [0, bw / 2, 0], [bl, 0.01, bd]),
              box(cb + [0, -bw / 2, 0], [bl, 0.01, bd]),
              box(cb + [0, 0, bd / 2], [bl, bw, 0.01]),
              box(cb + [0, 0, -bd / 2], [bl, bw, 0.01]),
              box(cb + [bl / 2, 0, 0], [0.01, bw, bd])]
        return ("sink to cupboard", [0, PI / 6, 0, -PI / 3 - 0.15, 0, -PI / 3, 0],
                [PI / 6, 5 * PI / 12, -PI / 2, -PI / 8, PI / 2, -PI / 2, 0], to_kinova(bs))
    if case == 7:    # reach through window
        wc = np.array([0.6, 0, 0.8])
        ws, oh, ow = 0.625, 1.5, 1.5
        bs = [box(wc + [0, 0, -ws / 2 - oh / 2], [0.01, 4, oh]),
              box(wc + [0, 0, ws / 2 + oh / 2], [0.01, 4, oh]),
              box(wc + [0, -ws / 2 - ow / 2, 0], [0.01, ow, 4]),
              box(wc + [0, ws / 2 + ow / 2, 0], [0.01, ow, 4])]
        return "window", [0, PI / 2, 0, -PI / 4, 0, 0, 0], [0, 0, 0, 0, PI / 3, PI / 3, 0], to_kinova(bs)
    raise ValueError(case)


def main():
    table = []
    for case in range(1, 8):
        name, start, goal, boxes = scenario(case)
        table.append(dict(scenario=case, name=name, start=[float(v) for v in start], goal=[float(v) for v in goal],
                          boxes=[[float(v) for v in np.concatenate([c, s])] for c, s in boxes]))
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scenes", "hard_scenarios.json")
    with open(out, "w") as f:
        json.dump(dict(source="KSI/kinova_scenarios/get_kinova_scenario_info.m:1-262 (+ make_shelf_obstacle.m), in the Kinova frame",
                       columns="boxes: cx cy cz sx sy sz", scenarios=table), f, indent=1)
    print(out, [len(t["boxes"]) for t in table])


if __name__ == "__main__":
    main()
