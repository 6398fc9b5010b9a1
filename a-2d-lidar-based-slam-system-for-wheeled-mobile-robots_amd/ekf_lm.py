"""EKF-SLAM filter over the robot pose and 2-D point landmarks (host side of the W12 node,
SURVEY.md 8f-4).

Mirrors ``EKF`` of W12m/ekf_lm.py:12-141 - state ``[x, y, yaw, l0x, l0y, l1x, ...]^T``
(column vector), prediction from the ICP odometry ``u = [dx, dy, dyaw]^T`` in the robot frame,
one range/bearing update per observed landmark with Mahalanobis-distance association
(``M_DIST_TH``; a new landmark is appended when nothing is closer).  The matrices are at most
a few dozen rows: sequential dense algebra with no data parallelism (SURVEY.md section 2 row
8), so it is NumPy on the host.  Formulas follow the reference line by line where the order of
floating-point operations matters; the sparse selector products (``Fx``, ``F``) are replaced by
the block operations they amount to.
"""
from __future__ import annotations

import math

import numpy as np

Cx = np.diag([0.35, 0.35, np.deg2rad(15.0)]) ** 2   # ekf_lm.py:5 (used as process AND measurement noise)
M_DIST_TH = 0.6                                      # :8
STATE_SIZE = 3
LM_SIZE = 2


class EKF:
    def pi_2_pi(self, angle):
        return (angle + math.pi) % (2 * math.pi) - math.pi                       # :141

    # ---------------------------------------------------------------- models
    def motion_model(self, x, u):
        """:52-63; also writes the new pose into x[:3] like the reference."""
        px, py, yaw = float(x[0, 0]), float(x[1, 0]), float(x[2, 0])
        ux, uy, uw = float(u[0, 0]), float(u[1, 0]), float(u[2, 0])
        s = np.array([[px + math.cos(yaw) * ux - math.sin(yaw) * uy],
                      [py + math.sin(yaw) * ux + math.cos(yaw) * uy],
                      [yaw + uw]])
        x[:3] = s
        return x[:3]

    def jacob_f(self, x, u):
        """:65-75 (the derivative part; ``estimate`` adds the identity)."""
        yaw, ux, uy = float(x[2, 0]), float(u[0, 0]), float(u[1, 0])
        jF = np.zeros((3, 3))
        jF[0, 2] = -math.sin(yaw) * ux - math.cos(yaw) * uy
        jF[1, 2] = math.cos(yaw) * ux - math.sin(yaw) * uy
        return jF

    def calc_landmark_position(self, x, z):
        """:77-83: range z[0], bearing z[1] seen from pose x -> 2x1 world position."""
        a = x[2, 0] + z[1]
        return np.array([[x[0, 0] + z[0] * math.cos(a)], [x[1, 0] + z[0] * math.sin(a)]])

    def get_landmark_position_from_state(self, x, ind):
        ind = int(ind)
        return x[3 + 2 * ind: 5 + 2 * ind, :]                                    # :85-88

    def jacob_h(self, q, delta, x, i):
        """:124-138: 2 x len(x) measurement Jacobian for landmark number i (1-based)."""
        sq = math.sqrt(q)
        dx, dy = delta[0, 0], delta[1, 0]
        G = np.array([[-sq * dx, -sq * dy, 0, sq * dx, sq * dy],
                      [dy, -dx, -q, -dy, dx]]) / q
        H = np.zeros((2, len(x)))
        H[:, :3] = G[:, :3]
        H[:, 1 + 2 * i: 3 + 2 * i] = G[:, 3:]
        return H

    def laser_correction(self, lm, xEst, PEst, z, LMid):
        """:110-122: innovation y (2x1), its covariance S and the Jacobian H."""
        delta = lm - xEst[:2]
        q = float(np.dot(delta.T, delta)[0][0])
        z_angle = math.atan2(delta[1][0], delta[0][0]) - xEst[2][0]
        zi = np.array([[math.sqrt(q), self.pi_2_pi(z_angle)]])
        y = (np.asarray(z, dtype=float).reshape(1, 2) - zi).T
        y[1] = self.pi_2_pi(y[1])
        H = self.jacob_h(q, delta, xEst, int(LMid) + 1)
        S = np.dot(np.dot(H, PEst), H.T) + Cx[:2, :2]
        return y, S, H

    def search_correspond_landmark_id(self, xAug, PAug, zi):
        """:90-108: index of the known landmark with the smallest Mahalanobis distance, or the
        landmark count when none is below ``M_DIST_TH`` (first minimum wins, as list.index)."""
        nLM = (len(xAug) - 3) // 2
        best, best_d = nLM, M_DIST_TH
        dists = []
        for i in range(nLM):
            lm = self.get_landmark_position_from_state(xAug, i)
            y, S, _ = self.laser_correction(lm, xAug, PAug, zi, i)
            dists.append(float(np.dot(np.dot(y.T, np.linalg.inv(S)), y)[0, 0]))
        dists.append(M_DIST_TH)
        for i, d in enumerate(dists):                     # min_dist.index(min(min_dist))
            if d < best_d or (d == best_d and i < best):
                best, best_d = i, d
        return best

    # ---------------------------------------------------------------- filter
    def estimate(self, xEst, PEst, z, u):
        """:15-50.  xEst (3+2n)x1, PEst square, z [m,3] rows (range, bearing, id), u 3x1 ->
        the new (xEst, PEst); both grow by one landmark whenever an observation matches none.
        xEst[:3] and PEst[:3,:3] are updated in place by the prediction, like the reference."""
        z = np.asarray(z, dtype=float).reshape(-1, 3)
        n1 = len(xEst)                                   # size BEFORE this call: the reference never refreshes it
        # predict
        G = np.eye(3) + self.jacob_f(xEst[:3], u)
        xEst[:3] = self.motion_model(xEst[:3], u)
        PEst[:3, :3] = np.dot(np.dot(G.transpose(), PEst[:3, :3]), G) + Cx
        # update
        for i in range(z.shape[0]):
            num_lm = (n1 - 3) // 2
            min_id = self.search_correspond_landmark_id(xEst, PEst, z[i, :2])
            if num_lm == min_id:                         # new landmark (:36-38)
                xEst = np.vstack((xEst, self.calc_landmark_position(xEst, z[i, :])))
                PEst = np.vstack((np.hstack((PEst, np.zeros((n1, 2)))),
                                  np.hstack((np.zeros((2, n1)), np.eye(2)))))
            lm = self.get_landmark_position_from_state(xEst, min_id)
            if len(lm) == 0:
                return xEst, PEst
            y, S, H = self.laser_correction(lm, xEst, PEst, z[i, 0:2], min_id)
            K = np.dot(np.dot(PEst, H.transpose()), np.linalg.inv(S))
            xEst = xEst + np.dot(K, y)
            PEst = np.dot((np.eye(len(xEst)) - np.dot(K, H)), PEst)
        xEst[2] = self.pi_2_pi(xEst[2])
        return xEst, PEst
