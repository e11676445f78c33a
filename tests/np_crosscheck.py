"""Independent NumPy/SciPy formulation of the hot path, used to cross-check the C++
oracle (and nothing else).  It shares no code with oracle/pcl_restatement.cpp:
voxel keys via vectorised float32 arithmetic + lexsort, clusters via cKDTree
candidate pairs re-tested with the float32 FLANN predicate + connected_components.
"""
import numpy as np
from scipy.sparse import coo_matrix
from scipy.sparse.csgraph import connected_components
from scipy.spatial import cKDTree

f32 = np.float32


def voxel_downsample_np(pts, leaf):
    pts = np.asarray(pts, dtype=f32)
    inv = f32(1.0) / f32(leaf)
    mn = pts.min(axis=0)
    mx = pts.max(axis=0)
    min_b = np.floor(mn * inv).astype(np.int64)
    max_b = np.floor(mx * inv).astype(np.int64)
    div = max_b - min_b + 1
    ijk = (np.floor(pts * inv) - min_b.astype(f32)).astype(np.int64)
    key = ijk[:, 0] + ijk[:, 1] * div[0] + ijk[:, 2] * div[0] * div[1]
    order = np.lexsort((np.arange(len(pts)), key))      # by key, ties by index
    ks = key[order]
    starts = np.flatnonzero(np.r_[True, ks[1:] != ks[:-1]])
    ends = np.r_[starts[1:], len(ks)]
    out = np.empty((len(starts), 3), dtype=f32)
    for v, (b, e) in enumerate(zip(starts, ends)):       # sequential float32 sums
        s = np.zeros(3, dtype=f32)
        for j in order[b:e]:
            s = (s + pts[j]).astype(f32)
        out[v] = s / f32(e - b)
    return out, key


def components_np(pts, tol):
    """labels[i] = smallest point index of i's component under d2 < r2 (float32 L2_Simple)."""
    pts = np.asarray(pts, dtype=f32)
    m = len(pts)
    r2 = f32(np.float64(f32(tol)) * np.float64(f32(tol)))
    tree = cKDTree(pts.astype(np.float64))
    pairs = tree.query_pairs(float(tol) * 1.001 + 1e-6, output_type="ndarray")
    if len(pairs):
        a, b = pts[pairs[:, 0]], pts[pairs[:, 1]]
        dx = (a[:, 0] - b[:, 0]).astype(f32)
        dy = (a[:, 1] - b[:, 1]).astype(f32)
        dz = (a[:, 2] - b[:, 2]).astype(f32)
        d2 = ((dx * dx).astype(f32) + (dy * dy).astype(f32)).astype(f32)
        d2 = (d2 + (dz * dz).astype(f32)).astype(f32)
        keep = d2 < r2
        pairs = pairs[keep]
    g = coo_matrix((np.ones(len(pairs), dtype=np.int8), (pairs[:, 0], pairs[:, 1])), shape=(m, m))
    _, lab = connected_components(g, directed=False)
    first = np.full(lab.max() + 1 if m else 0, m, dtype=np.int64)
    np.minimum.at(first, lab, np.arange(m))
    return first[lab]


def clusters_from_labels(labels, min_cluster, max_cluster, order_fn):
    """CSR in PCL's order given canonical labels; order_fn = the size-sequence permutation."""
    labels = np.asarray(labels)
    roots, inv, counts = np.unique(labels, return_inverse=True, return_counts=True)  # ascending root = discovery
    keep = (counts >= min_cluster) & (counts <= max_cluster)
    kept_ids = np.flatnonzero(keep)
    sizes = counts[kept_ids]
    perm = order_fn(sizes) if len(sizes) else np.zeros(0, dtype=np.int64)
    offsets = [0]
    indices = []
    for c in perm:
        members = np.flatnonzero(inv == kept_ids[c])
        indices.append(members)
        offsets.append(offsets[-1] + len(members))
    idx = np.concatenate(indices).astype(np.int32) if indices else np.zeros(0, dtype=np.int32)
    return np.asarray(offsets, dtype=np.int32), idx
