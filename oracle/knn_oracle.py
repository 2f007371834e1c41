"""TEST INFRASTRUCTURE ONLY — CPU oracle of `simple_knn._C.distCUDA2` (reference call site
scene/gaussian_model.py:444; the extension itself is absent from /root/reference: PARITY UNPINNED, semantics
[UPSTREAM-RECALL]: mean of the squared distances to the 3 nearest OTHER points, duplicates count at distance 0).
Two independent implementations: an fp32 brute force in numpy (same arithmetic as the kernel) and scipy's cKDTree
in fp64."""
import numpy as np


def mean_dist2_bruteforce(points: np.ndarray, chunk: int = 2048) -> np.ndarray:
    p = np.ascontiguousarray(points, dtype=np.float32)
    n = p.shape[0]
    out = np.zeros(n, dtype=np.float32)
    have = min(3, n - 1)
    if have <= 0:
        return out
    for s in range(0, n, chunk):
        q = p[s:s + chunk]
        dx = q[:, None, 0] - p[None, :, 0]
        dy = q[:, None, 1] - p[None, :, 1]
        dz = q[:, None, 2] - p[None, :, 2]
        d = dx * dx + dy * dy + dz * dz                      # fp32, same order as the kernel
        d[np.arange(q.shape[0]), np.arange(s, s + q.shape[0])] = np.inf     # exclude self (by index, not by value)
        best = np.partition(d, have - 1, axis=1)[:, :have]
        best.sort(axis=1)
        acc = best[:, 0].copy()
        for k in range(1, have):
            acc = acc + best[:, k]
        out[s:s + chunk] = acc / np.float32(have)
    return out


def mean_dist2_kdtree(points: np.ndarray) -> np.ndarray:
    from scipy.spatial import cKDTree
    p = np.asarray(points, dtype=np.float64)
    n = p.shape[0]
    have = min(3, n - 1)
    if have <= 0:
        return np.zeros(n)
    d, idx = cKDTree(p).query(p, k=have + 1)
    # the query returns the point itself (distance 0) among the k+1 nearest; with exact duplicates "itself" may be
    # another copy, which is equally at distance 0 — dropping one zero-distance entry per row is correct either way.
    return (d[:, 1:] ** 2).mean(axis=1)
