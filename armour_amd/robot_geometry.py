"""Joint frames of an ArmourRobot as plain matrices (host side; used by the CORA-semantics mode and by tests):
rotation of each joint frame in its parent (rpy, RT/PZsparse.cu:160-176), translation, joint axis, link bounding box."""
import numpy as np


def rpy_matrix(roll, pitch, yaw):
    cr, sr, cp, sp, cy, sy = np.cos(roll), np.sin(roll), np.cos(pitch), np.sin(pitch), np.cos(yaw), np.sin(yaw)
    return np.array([[cp * cy, -cp * sy, sp],
                     [cr * sy + cy * sp * sr, cr * cy - sp * sr * sy, -cp * sr],
                     [sr * sy - cr * cy * sp, cy * sr + cr * sp * sy, cp * cr]])


def joint_frames(robot):
    """(T0 [n,3,3], P [3,n], axes [n,3], link centres [n,3], link half sizes [n,3]) of the actuated chain"""
    n = robot.num_factors
    rots = np.array(robot.rots)[:3 * n].reshape(n, 3)
    T0 = np.stack([rpy_matrix(*r) for r in rots])
    P = np.array(robot.trans)[:3 * n].reshape(n, 3).T
    axes = np.zeros((n, 3))
    for i in range(n):
        a = robot.axes[i]
        axes[i, abs(a) - 1] = 1.0 if a > 0 else -1.0
    centers = np.array(robot.link_zonotope_center)[:3 * n].reshape(n, 3)
    half = np.array(robot.link_zonotope_generators)[:3 * n].reshape(n, 3)
    return T0, P, axes, centers, half


def rot_axis(axis, q):
    """rotation by q about a unit axis (Rodrigues)"""
    e = np.asarray(axis, dtype=float)
    U = np.array([[0, -e[2], e[1]], [e[2], 0, -e[0]], [-e[1], e[0], 0]])
    return np.eye(3) + np.sin(q) * U + (1 - np.cos(q)) * (U @ U)


def link_frames(robot, q):
    """world rotation and origin of every actuated joint frame at joint angles q (the scalar form of pzfk / RT fk)"""
    T0, P, axes, _, _ = joint_frames(robot)
    Rw, pw, out = np.eye(3), np.zeros(3), []
    for i in range(len(axes)):
        pw = pw + Rw @ P[:, i]
        Rw = Rw @ T0[i] @ rot_axis(axes[i], q[i])
        out.append((Rw.copy(), pw.copy()))
    return out
