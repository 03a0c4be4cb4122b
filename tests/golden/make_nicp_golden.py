"""Generates tests/golden/nicp_2d_known_answer.json.

The reference holds NO golden vector for the aligner, the finders or the factor, so
this known-answer set is derived independently of the oracle's C code: it evaluates the 3-D
point-to-plane + normal-difference error of the reference's Octave prototype
(srrg2_laser_slam_2d/octave/solver/nicp_post.m:4-26: e = [n_f'(R p + t - p_f); R n - n_f],
J = [n_f' R, -n_f' R [p]x ; 0, -R [n]x], right-multiplied increment, :92-97) with float64 numpy on
three planar correspondences, then restricts it to the SE(2) unknowns (t_x, t_y, rot_z) and the
error rows (point-plane, normal x, normal y).  Jacobian columns are checked against central
differences of the error under X <- X * v2t(dx) before anything is written.

    python tests/golden/make_nicp_golden.py
"""
import json
import os

import numpy as np


def rot_z(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])


def cross_matrix(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0.0]])


def error_and_jacobian_3d(moving6, fixed6, R, t):
    p, n = moving6[:3], moving6[3:]
    pf, nf = fixed6[:3], fixed6[3:]
    e = np.concatenate([[nf @ (R @ p + t - pf)], R @ n - nf])
    J = np.zeros((4, 6))
    J[0, :3] = nf @ R
    J[0, 3:] = -nf @ R @ cross_matrix(p)
    J[1:, 3:] = -R @ cross_matrix(n)
    return e, J


def main():
    pose = np.array([0.3, -0.2, 0.15])
    fixed = np.array([[1.0, 2.0, 0.6, 0.8], [-1.5, 0.5, -1.0, 0.0], [2.5, -1.0, 0.0, -1.0]])
    moving = np.array([[0.9, 2.2, 0.5547002, 0.8320503], [-1.2, 0.1, -0.9805807, 0.1961161], [2.2, -1.3, 0.1240347, -0.9922779]])
    R, t = rot_z(pose[2]), np.array([pose[0], pose[1], 0.0])
    keep_rows, keep_cols = [0, 1, 2], [0, 1, 5]
    H = np.zeros((3, 3)); b = np.zeros(3); chi = 0.0
    per_pair = []
    for f, m in zip(fixed, moving):
        f6 = np.array([f[0], f[1], 0, f[2], f[3], 0]); m6 = np.array([m[0], m[1], 0, m[2], m[3], 0])
        e, J = error_and_jacobian_3d(m6, f6, R, t)
        e2, J2 = e[keep_rows], J[np.ix_(keep_rows, keep_cols)]
        # finite-difference check of the right-perturbation Jacobian
        for k in range(3):
            d = np.zeros(3); d[k] = 1e-6
            def err(dx):
                Rp = R @ rot_z(dx[2]); tp = t + R @ np.array([dx[0], dx[1], 0.0])
                return error_and_jacobian_3d(m6, f6, Rp, tp)[0][keep_rows]
            num = (err(d) - err(-d)) / 2e-6
            assert np.allclose(num, J2[:, k], atol=1e-8), (num, J2[:, k])
        H += J2.T @ J2; b += J2.T @ e2; chi += e2 @ e2
        per_pair.append({"e": e2.tolist(), "J": J2.tolist(), "chi": float(e2 @ e2)})
    dx = -np.linalg.solve(H, b)
    c, s = np.cos(pose[2]), np.sin(pose[2])
    new_pose = [pose[0] + c * dx[0] - s * dx[1], pose[1] + s * dx[0] + c * dx[1], pose[2] + dx[2]]
    # Cauchy-weighted variant: w = 1/(1+chi/tau), inlier <=> chi < tau
    tau = 0.05
    Hc = np.zeros((3, 3)); bc = np.zeros(3); n_in = 0; chi_in = 0.0; chi_out = 0.0
    for pp in per_pair:
        e2, J2, ch = np.array(pp["e"]), np.array(pp["J"]), pp["chi"]
        w = 1.0 / (1.0 + ch / tau)
        Hc += w * J2.T @ J2; bc += w * J2.T @ e2
        if ch < tau:
            n_in += 1; chi_in += ch
        else:
            chi_out += tau * np.log(1.0 + ch / tau)
    out = {"source": "independent float64 evaluation of octave/solver/nicp_post.m:4-26,69-97 restricted to SE(2)",
           "pose": pose.tolist(), "fixed": fixed.tolist(), "moving": moving.tolist(),
           "pairs": per_pair, "H": H.tolist(), "b": b.tolist(), "chi": chi, "dx": dx.tolist(), "pose_after_step": new_pose,
           "cauchy": {"tau": tau, "H": Hc.tolist(), "b": bc.tolist(), "n_inliers": n_in, "chi_inliers": chi_in, "chi_outliers": chi_out}}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "nicp_2d_known_answer.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
