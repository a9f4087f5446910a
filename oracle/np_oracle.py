"""Independent numpy/scipy restatement of the hot path (cross-check for the C oracle).

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see oracle/o3d_oracle.h).
Written separately from o3d_oracle.c, against SURVEY.md Appendix A, using
scipy.spatial.cKDTree(leafsize=15) for the exact NN search and numpy.linalg
for the 6x6 solve and the 3x3 eigen-decomposition, so that agreement between
the two is evidence that each restates Open3D v0.15.1's algorithm rather than
sharing one bug.
"""
from __future__ import annotations

import numpy as np
from scipy.spatial import cKDTree


def rot_zyx(a, b, g):
    ca, sa, cb, sb, cg, sg = np.cos(a), np.sin(a), np.cos(b), np.sin(b), np.cos(g), np.sin(g)
    Rx = np.array([[1, 0, 0], [0, ca, -sa], [0, sa, ca]])
    Ry = np.array([[cb, 0, sb], [0, 1, 0], [-sb, 0, cb]])
    Rz = np.array([[cg, -sg, 0], [sg, cg, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def vector6_to_matrix4(x):
    T = np.eye(4)
    T[:3, :3] = rot_zyx(x[0], x[1], x[2])
    T[:3, 3] = x[3:6]
    return T


def evaluate(tree: cKDTree, P, r):
    d, j = tree.query(P, k=1, distance_upper_bound=r)  # inf / n when none
    ok = np.isfinite(d) & (d * d < r * r)
    nc = int(ok.sum())
    if nc == 0:
        return np.where(ok, j, -1), 0.0, 0.0, 0
    return np.where(ok, j, -1), nc / len(P), float(np.sqrt(np.sum(d[ok] ** 2) / nc)), nc


def jtj_jtr(P, Q, Nq, corr):
    m = corr >= 0
    p, q, n = P[m], Q[corr[m]], Nq[corr[m]]
    r = np.einsum("ij,ij->i", p - q, n)
    J = np.concatenate([np.cross(p, n), n], axis=1)
    return J.T @ J, J.T @ r, float(r @ r)


def icp_point_to_plane(src, tgt, nrm, max_corr, init=None, max_iter=30, rel_fitness=1e-6, rel_rmse=1e-6):
    tree = cKDTree(tgt, leafsize=15)
    T = np.eye(4) if init is None else np.array(init, dtype=np.float64)
    P = src @ T[:3, :3].T + T[:3, 3]
    corr, fit, rmse, nc = evaluate(tree, P, max_corr)
    it = 0
    for _ in range(max_iter):
        if nc == 0:
            U = np.eye(4)
        else:
            A, b, _ = jtj_jtr(P, tgt, nrm, corr)
            U = vector6_to_matrix4(np.linalg.solve(A, -b))
        T = U @ T
        P = P @ U[:3, :3].T + U[:3, 3]
        pf, pr = fit, rmse
        corr, fit, rmse, nc = evaluate(tree, P, max_corr)
        it += 1
        if abs(pf - fit) < rel_fitness and abs(pr - rmse) < rel_rmse:
            break
    return dict(transformation=T, fitness=fit, inlier_rmse=rmse, iterations=it, n_corr=nc)


def umeyama_update(P, Q, corr):
    """Eigen::umeyama(source, target, with_scaling=False) on the matched pairs ([O3D] TransformationEstimationPointToPoint)."""
    m = corr >= 0
    if not m.any():
        return np.eye(4)
    s, t = P[m], Q[corr[m]]
    ms, mt = s.mean(0), t.mean(0)
    sigma = (t - mt).T @ (s - ms) / len(s)
    U, _, Vt = np.linalg.svd(sigma)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[2, 2] = -1.0
    R = U @ S @ Vt
    out = np.eye(4)
    out[:3, :3] = R
    out[:3, 3] = mt - R @ ms
    return out


def icp_point_to_point(src, tgt, max_corr, init=None, max_iter=30, rel_fitness=1e-6, rel_rmse=1e-6):
    tree = cKDTree(tgt, leafsize=15)
    T = np.eye(4) if init is None else np.array(init, dtype=np.float64)
    P = src @ T[:3, :3].T + T[:3, 3]
    corr, fit, rmse, nc = evaluate(tree, P, max_corr)
    it = 0
    for _ in range(max_iter):
        U = umeyama_update(P, tgt, corr)
        T = U @ T
        P = P @ U[:3, :3].T + U[:3, 3]
        pf, pr = fit, rmse
        corr, fit, rmse, nc = evaluate(tree, P, max_corr)
        it += 1
        if abs(pf - fit) < rel_fitness and abs(pr - rmse) < rel_rmse:
            break
    return dict(transformation=T, fitness=fit, inlier_rmse=rmse, iterations=it, n_corr=nc)


def information_matrix(src, tgt, max_corr, T=None):
    """[O3D] GetInformationMatrixFromPointClouds: sum of G^T G over matched target points, G = [-[q]x | I]."""
    T = np.eye(4) if T is None else np.asarray(T, dtype=np.float64)
    P = src @ T[:3, :3].T + T[:3, 3]
    corr, _, _, _ = evaluate(cKDTree(tgt, leafsize=15), P, max_corr)
    q = tgt[corr[corr >= 0]]
    x, y, z = q[:, 0], q[:, 1], q[:, 2]
    o, l = np.zeros_like(x), np.ones_like(x)
    G = np.stack([np.stack([o, z, -y, l, o, o], 1), np.stack([-z, o, x, o, l, o], 1), np.stack([y, -x, o, o, o, l], 1)], 1)  # m x 3 x 6
    return np.einsum("mra,mrb->ab", G, G)


def carve_flags(scan, sensor, map_pts, map_nrm, subset, voxel=0.1, max_length=20.0, truncation=0.1, min_dot=0.5):
    """getIdxsOfCarvedPoints (helpers.cpp:235-271), one ray-marching step for all rays at a time."""
    sensor = np.asarray(sensor, dtype=np.float64)
    inv = 1.0 / voxel
    subset = np.asarray(subset, dtype=np.int64)
    keys = np.floor(map_pts[subset] * inv).astype(np.int64)
    table = {}
    for k, i in zip(map(tuple, keys), subset):
        table.setdefault(k, []).append(int(i))
    d = scan - sensor
    length = np.linalg.norm(d, axis=1)
    ok = length > 0
    direction = np.zeros_like(d)
    direction[ok] = d[ok] / length[ok, None]
    lim = np.maximum(voxel, np.minimum(length - truncation, max_length))
    flags = np.zeros(len(map_pts), dtype=bool)
    unit = None
    if map_nrm is not None:
        nl = np.linalg.norm(map_nrm, axis=1)
        unit = np.where(nl[:, None] > 0, map_nrm / np.where(nl > 0, nl, 1.0)[:, None], 0.0)
    dist = 0.0
    active = ok.copy()
    while True:
        active &= dist < lim
        if not active.any():
            break
        pos = dist * direction[active] + sensor
        kk = np.floor(pos * inv).astype(np.int64)
        for k, dv in zip(map(tuple, kk), direction[active]):
            ids = table.get(k)
            if ids is None:
                continue
            for i in ids:
                if unit is None or abs(float(dv @ unit[i])) > min_dot:
                    flags[i] = True
        dist += voxel
    return flags


def overlap_indices(src, tgt, T=None, voxel=0.5, min_points=1):
    """computeIndicesOfOverlappingPoints (helpers.cpp:307-332) with np.unique on the voxel keys."""
    T = np.eye(4) if T is None else np.asarray(T, dtype=np.float64)
    P = src @ T[:3, :3].T + T[:3, 3]
    inv = 1.0 / voxel
    ks, kt = np.floor(P * inv).astype(np.int64), np.floor(tgt * inv).astype(np.int64)
    allk, inv_idx = np.unique(np.vstack([ks, kt]), axis=0, return_inverse=True)
    inv_idx = inv_idx.reshape(-1)
    cs = np.bincount(inv_idx[: len(ks)], minlength=len(allk))
    ct = np.bincount(inv_idx[len(ks):], minlength=len(allk))
    ok = (cs >= min_points) & (ct >= min_points)
    return np.flatnonzero(ok[inv_idx[: len(ks)]]), np.flatnonzero(ok[inv_idx[len(ks):]])


def dense_fuse(pts, nrm, voxel):
    """VoxelizedPointCloud (Voxel.cpp:66-114): per-voxel sums / counts; voxels in ascending key order (z, y, x)."""
    keys = np.floor(pts * (1.0 / voxel)).astype(np.int64)
    uniq, inv, cnt = np.unique(keys, axis=0, return_inverse=True, return_counts=True)
    inv = inv.reshape(-1)
    sp = np.zeros((len(uniq), 3))
    np.add.at(sp, inv, pts)
    out_n = None
    if nrm is not None:
        sn = np.zeros((len(uniq), 3))
        np.add.at(sn, inv, nrm)
        out_n = sn / cnt[:, None]
    return sp / cnt[:, None], out_n, cnt


def undistort(pts, lin_vel, ang_vel_rpy, scan_duration, clockwise=False):
    """MotionCompensation.cpp:64-139, vectorised."""
    lin_vel, ang = np.asarray(lin_vel, dtype=np.float64), np.asarray(ang_vel_rpy, dtype=np.float64)
    a = np.arctan2(pts[:, 1], pts[:, 0])
    w = np.where(a < 0, a + 2 * np.pi, a)
    phase = np.where(w == 0.0, 0.0, (1.0 - w / (2 * np.pi)) if clockwise else w / (2 * np.pi))
    s = phase * scan_duration
    out = np.empty_like(pts)
    for i in range(len(pts)):
        U = vector6_to_matrix4(np.concatenate([s[i] * ang, s[i] * lin_vel]))
        out[i] = U[:3, :3] @ pts[i] + U[:3, 3]
    return out


def dense_carve(scan, sensor, voxel_points, voxel, radius=0.1, max_length=20.0, truncation=0.1):
    """Voxel.cpp:162-191 + helpers.cpp:347-377 + VoxelHashMap.cpp:13-44 with python sets (small inputs only)."""
    if not (radius > 0.0 and voxel > 0.0):
        raise ValueError("dense_carve: radius and voxel must be > 0 (the reference's ray step 2 * radius would be zero)")
    sensor = np.asarray(sensor, dtype=np.float64)
    inv = 1.0 / voxel
    occ = {tuple(k): i for i, k in enumerate(np.floor(np.asarray(voxel_points) * inv).astype(np.int64))}
    removed = np.zeros(len(voxel_points), dtype=bool)
    seen = set()
    step = 2.0 * radius

    def key_div(p):
        return tuple(int(math.floor(c / voxel)) for c in p)

    import math

    for p in scan:
        k = tuple(np.floor(p * inv).astype(np.int64))
        if k in seen:  # removeDuplicatePointsWithinSameVoxels keeps the first point of a voxel
            continue
        seen.add(k)
        d = p - sensor
        length = float(np.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]))
        if not length > 0.0:
            continue
        direction = d / length
        lim = max(step, min(length - truncation, max_length))
        dist = 0.0
        while dist < lim:
            pos = dist * direction + sensor
            ck = key_div(pos)
            keys = []
            if True:
                added = False
                dx = -radius
                while dx <= radius:
                    dy = -radius
                    while dy <= radius:
                        dz = -radius
                        while dz <= radius:
                            tp = pos + np.array([dx, dy, dz])
                            kk = key_div(tp)
                            c = np.array(kk, dtype=np.float64) * voxel + voxel * 0.5
                            e = tp - c
                            if math.sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]) <= radius:
                                keys.append(kk)
                                added = added or kk == ck
                            dz += voxel
                        dy += voxel
                    dx += voxel
                if not added:
                    keys.append(ck)
            for kk in keys:
                i = occ.get(kk)
                if i is not None:
                    removed[i] = True
            dist += step
    return removed


def estimate_normals(pts, radius, max_nn):
    tree = cKDTree(pts, leafsize=15)
    d, j = tree.query(pts, k=max_nn, distance_upper_bound=radius)
    if max_nn == 1:
        d, j = d[:, None], j[:, None]
    out = np.empty_like(pts)
    for i in range(len(pts)):
        ok = np.isfinite(d[i]) & (d[i] ** 2 < radius * radius)
        nb = pts[j[i][ok]]
        if len(nb) >= 3:
            mu = nb.mean(0)
            cov = (nb.T @ nb) / len(nb) - np.outer(mu, mu)
        else:
            cov = np.eye(3)
        w, v = np.linalg.eigh(cov)
        n = v[:, 0]
        if len(nb) < 3:
            n = np.array([0.0, 0.0, 1.0])  # identity covariance -> diagonal branch -> (0,0,1)
        n = n / np.linalg.norm(n)
        if n @ (-pts[i]) < 0:
            n = -n
        out[i] = n
    return out


def voxel_down_sample(pts, voxel):
    o = pts.min(0) - voxel / 2
    keys = np.floor((pts - o) / voxel).astype(np.int64)
    uk, inv = np.unique(keys, axis=0, return_inverse=True)
    inv = inv.ravel()
    out = np.zeros((len(uk), 3))
    np.add.at(out, inv, pts)
    cnt = np.bincount(inv, minlength=len(uk))
    return out / cnt[:, None], uk


def voxelize_world(pts, nrm, voxel):
    keys = np.floor(pts * (1.0 / voxel)).astype(np.int64)
    uk, inv = np.unique(keys, axis=0, return_inverse=True)
    inv = inv.ravel()
    out = np.zeros((len(uk), 3))
    on = np.zeros((len(uk), 3))
    np.add.at(out, inv, pts)
    np.add.at(on, inv, nrm)
    cnt = np.bincount(inv, minlength=len(uk))
    on = on / cnt[:, None]
    nn = np.linalg.norm(on, axis=1, keepdims=True)
    on = np.where(nn > 0, on / np.where(nn > 0, nn, 1), on)
    return out / cnt[:, None], on, uk


# ---- A.8 generalized ICP (independent restatement: numpy eigh for M^-1/2, explicit 3-row Jacobians) --------------------
def rotation_e1_to_x(x):
    e1 = np.array([1.0, 0.0, 0.0])
    v = np.cross(e1, x)
    c = float(e1 @ x)
    if c < -0.99:
        return np.eye(3)
    sv = np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
    return np.eye(3) + sv + (sv @ sv) / (1.0 + c)


def covariances_from_normals(nrm, eps=1e-3):
    C = np.diag([eps, 1.0, 1.0])
    out = np.empty((len(nrm), 3, 3))
    for i, n in enumerate(nrm):
        R = rotation_e1_to_x(n)
        out[i] = R @ C @ R.T
    return out


def gicp_jtj_jtr(P, Cs, Q, Ct, corr):
    A = np.zeros((6, 6))
    b = np.zeros(6)
    for i in np.nonzero(corr >= 0)[0]:
        j = corr[i]
        d = P[i] - Q[j]
        w, V = np.linalg.eigh(Ct[j] + Cs[i])
        W = (V / np.sqrt(w)) @ V.T
        p = P[i]
        S = -np.array([[0, -p[2], p[1]], [p[2], 0, -p[0]], [-p[1], p[0], 0]])
        J = W @ np.hstack([S, np.eye(3)])
        r = W @ d
        A += J.T @ J
        b += J.T @ r
    return A, b


def icp_generalized(src, src_nrm, tgt, tgt_nrm, max_corr, init=None, max_iter=30, rel_fitness=1e-6, rel_rmse=1e-6, eps=1e-3):
    tree = cKDTree(tgt, leafsize=15)
    T = np.eye(4) if init is None else np.array(init, dtype=np.float64)
    Cs0 = covariances_from_normals(src_nrm, eps)
    Ct = covariances_from_normals(tgt_nrm, eps)
    R0 = T[:3, :3]
    P = src @ R0.T + T[:3, 3]
    Cs = np.einsum("ab,nbc,dc->nad", R0, Cs0, R0)
    corr, fit, rmse, nc = evaluate(tree, P, max_corr)
    it = 0
    for _ in range(max_iter):
        if nc == 0:
            U = np.eye(4)
        else:
            A, b = gicp_jtj_jtr(P, Cs, tgt, Ct, corr)
            U = vector6_to_matrix4(np.linalg.solve(A, -b))
        T = U @ T
        P = P @ U[:3, :3].T + U[:3, 3]
        Cs = np.einsum("ab,nbc,dc->nad", U[:3, :3], Cs, U[:3, :3])
        pf, pr = fit, rmse
        corr, fit, rmse, nc = evaluate(tree, P, max_corr)
        it += 1
        if abs(pf - fit) < rel_fitness and abs(pr - rmse) < rel_rmse:
            break
    return dict(transformation=T, fitness=fit, inlier_rmse=rmse, iterations=it, n_corr=nc)


def merge_last_colors(pts, col, voxel):
    """Colours of the map merge for points that are all inside the cropping volume (helpers.cpp:40-42,61-63): per world-anchored
    voxel floor(p / v) the colour of the LAST point in cloud order; returns (voxel keys (m,3) sorted, colours (m,3))."""
    keys = np.floor(np.asarray(pts) * (1.0 / voxel)).astype(np.int64)
    uk, inv = np.unique(keys, axis=0, return_inverse=True)
    inv = inv.reshape(-1)
    last = np.full(len(uk), -1, dtype=np.int64)
    np.maximum.at(last, inv, np.arange(len(keys)))
    return uk, np.asarray(col)[last]
