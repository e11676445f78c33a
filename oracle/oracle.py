"""ctypes front-end of the CPU oracle (oracle/pcl_restatement.cpp: voxel grid, radius graph, clusters;
oracle/occupancy_restatement.cpp: bounds filter + occupancy binning; oracle/sor_restatement.cpp: statistical outlier removal;
oracle/plane_restatement.cpp: RANSAC plane removal).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpcop_oracle.so")

LAYOUT_AOS16, LAYOUT_SOA, LAYOUT_AOS12 = 0, 1, 2

_lib = None


def build(force=False):
    """Compile the oracle with g++ (seconds)."""
    srcs = [os.path.join(_HERE, f) for f in ("pcl_restatement.cpp", "occupancy_restatement.cpp", "sor_restatement.cpp", "plane_restatement.cpp")]
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libpcop_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        fp, ip, up = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_uint32)
        L.pcop_oracle_r2.restype = C.c_float
        L.pcop_oracle_r2.argtypes = [C.c_float]
        L.pcop_oracle_voxel_keys.restype = C.c_int
        L.pcop_oracle_voxel_keys.argtypes = [fp, C.c_int, C.c_size_t, C.c_float, C.c_int, up, ip]
        L.pcop_oracle_voxel_downsample.restype = C.c_int
        L.pcop_oracle_voxel_downsample.argtypes = [fp, C.c_int, C.c_size_t, C.c_float, C.c_int, fp,
                                                   C.POINTER(C.c_size_t), C.POINTER(C.c_int)]
        L.pcop_oracle_voxel_downsample_ex.restype = C.c_int
        L.pcop_oracle_voxel_downsample_ex.argtypes = [fp, C.c_int, C.c_size_t, C.c_float, C.c_int, C.c_int, fp,
                                                      C.POINTER(C.c_size_t), C.POINTER(C.c_int)]
        L.pcop_oracle_knife_edge_report.restype = C.c_int
        L.pcop_oracle_knife_edge_report.argtypes = [fp, C.c_int, C.c_size_t, C.c_float, C.c_int, C.POINTER(C.c_int64)]
        L.pcop_oracle_euclidean_clusters_ex.restype = C.c_int
        L.pcop_oracle_euclidean_clusters_ex.argtypes = [fp, C.c_int, C.c_size_t, C.c_float, C.c_int32, C.c_int32,
                                                        C.c_int, ip, ip, ip]
        L.pcop_oracle_cluster_order.restype = None
        L.pcop_oracle_cluster_order.argtypes = [ip, C.c_int32, ip]
        L.pcop_oracle_cluster_summaries.restype = C.c_int
        L.pcop_oracle_cluster_summaries.argtypes = [fp, C.c_int, C.c_size_t, ip, ip, C.c_int32, fp]
        L.pcop_oracle_knife_edge_pairs.restype = C.c_int64
        L.pcop_oracle_knife_edge_pairs.argtypes = [fp, C.c_int, C.c_size_t, C.c_float, C.c_int]
        L.pcop_oracle_process_xyz.restype = C.c_int
        L.pcop_oracle_process_xyz.argtypes = [fp, C.c_int, C.c_size_t, C.c_int, C.c_float, C.c_float, C.c_int32,
                                              C.c_int32, fp, C.POINTER(C.c_size_t), C.POINTER(C.c_int), ip, ip, ip]
        L.pcop_oracle_occ_grid_dims.restype = None
        L.pcop_oracle_occ_grid_dims.argtypes = [C.c_float] * 5 + [ip, ip]
        L.pcop_oracle_occ_cell.restype = C.c_int32
        L.pcop_oracle_occ_cell.argtypes = [C.c_float, C.c_float, C.POINTER(OccParams)]
        L.pcop_oracle_occupancy.restype = C.c_int
        L.pcop_oracle_occupancy.argtypes = [fp, C.c_size_t, C.POINTER(OccParams), fp, C.POINTER(C.c_size_t),
                                            C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_byte), ip]
        L.pcop_oracle_sor.restype = C.c_int
        L.pcop_oracle_sor.argtypes = [fp, C.c_size_t, C.c_int, C.c_float, fp, C.POINTER(C.c_double), C.POINTER(C.c_uint8),
                                      C.POINTER(C.c_size_t)]
        L.pcop_oracle_segment_planes.restype = C.c_int
        L.pcop_oracle_segment_planes.argtypes = [fp, C.c_size_t, C.c_float, fp, C.c_double, fp, C.POINTER(C.c_size_t), fp,
                                                 C.POINTER(C.c_size_t), fp, C.c_int, ip, ip]
        L.pcop_oracle_plane_rnd.restype = None
        L.pcop_oracle_plane_rnd.argtypes = [ip, C.c_int]
        _lib = L
    return _lib


class OccParams(C.Structure):
    """pcop_oracle_occ_params: the occupancy step's limits, block size, deviation and grid size (od.cpp:87-102, 958-959)."""
    _fields_ = [("x_min", C.c_float), ("x_max", C.c_float), ("y_min", C.c_float), ("y_max", C.c_float),
                ("z_min", C.c_float), ("z_max", C.c_float), ("block_size", C.c_float), ("dev_percent", C.c_float),
                ("grid_width", C.c_int32), ("grid_height", C.c_int32)]


def occ_params(x_min, x_max, y_min, y_max, z_min, z_max, block_size, dev_percent):
    """Parameters with the grid size derived as the reference's main() does (od.cpp:958-959)."""
    w, h = C.c_int32(0), C.c_int32(0)
    lib().pcop_oracle_occ_grid_dims(x_min, x_max, y_min, y_max, block_size, C.byref(w), C.byref(h))
    return OccParams(x_min, x_max, y_min, y_max, z_min, z_max, block_size, dev_percent, w.value, h.value)


def occ_cell(px, py, p):
    return lib().pcop_oracle_occ_cell(C.c_float(px), C.c_float(py), C.byref(p))


def occupancy(xyz, p):
    """build_initial_occupancy_grid_dataset -> dict(xyz, counts, row_avg, grid, index_errors)."""
    pts = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
    n = len(pts)
    cells, rows = p.grid_width * p.grid_height, p.grid_height
    out = np.zeros(max(3 * n, 3), dtype=np.float32)
    counts = np.zeros(max(cells, 1), dtype=np.int64)
    row_avg = np.zeros(max(rows, 1), dtype=np.int64)
    grid = np.zeros(max(cells, 1), dtype=np.int8)
    n_out, bad = C.c_size_t(0), C.c_int32(0)
    rc = lib().pcop_oracle_occupancy(_f(pts), n, C.byref(p), _f(out), C.byref(n_out),
                                     counts.ctypes.data_as(C.POINTER(C.c_longlong)),
                                     row_avg.ctypes.data_as(C.POINTER(C.c_longlong)),
                                     grid.ctypes.data_as(C.POINTER(C.c_byte)), C.byref(bad))
    if rc:
        raise ValueError("pcop_oracle_occupancy: %d" % rc)
    k = n_out.value
    return dict(xyz=out[:3 * k].reshape(k, 3).copy(), n_out=k, counts=counts[:cells], row_avg=row_avg[:rows],
                grid=grid[:cells], index_errors=bad.value)


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _pack(xyz, layout):
    """xyz: (n,3) float32 -> flat float32 array in `layout`."""
    xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
    n = xyz.shape[0]
    if layout == LAYOUT_AOS12:
        return xyz.reshape(-1).copy(), n
    if layout == LAYOUT_SOA:
        return np.ascontiguousarray(xyz.T).reshape(-1).copy(), n
    out = np.ones((n, 4), dtype=np.float32)
    out[:, :3] = xyz
    return out.reshape(-1), n


def _unpack(flat, layout, m, stride):
    if layout == LAYOUT_AOS12:
        return flat[: 3 * m].reshape(m, 3).copy()
    if layout == LAYOUT_SOA:
        return np.stack([flat[0:m], flat[stride:stride + m], flat[2 * stride:2 * stride + m]], axis=1).copy()
    return flat[: 4 * m].reshape(m, 4)[:, :3].copy()


def r2(tol):
    return np.float32(lib().pcop_oracle_r2(C.c_float(float(np.float32(tol)))))


def voxel_keys(xyz, leaf, is_dense=True, layout=LAYOUT_AOS12):
    flat, n = _pack(xyz, layout)
    keys = np.zeros(max(n, 1), dtype=np.uint32)
    grid = np.zeros(9, dtype=np.int32)
    pt = lib().pcop_oracle_voxel_keys(_f(flat), layout, n, np.float32(leaf), int(is_dense),
                                      keys.ctypes.data_as(C.POINTER(C.c_uint32)), _i(grid))
    return keys[:n], grid, bool(pt)


VARIANT_RECIPROCAL = 1   # centroid = sum * (1.0f / n): PCL 1.7 + Eigen 3.2
VARIANT_DESCENDING = 2   # equal voxel keys keep descending point index: another order an unstable std::sort may leave
VARIANT_STDSORT = 4      # equal voxel keys in the order std::sort leaves them when it compares the key alone (PCL's own call, this image's libstdc++)


def voxel_downsample(xyz, leaf, is_dense=True, layout=LAYOUT_AOS12, variant=0):
    """-> (centroids (m,3) float32, passthrough bool).  variant: the other legal instances of PCL's open choices."""
    flat, n = _pack(xyz, layout)
    out = np.zeros(max(4 * n, 4), dtype=np.float32)
    m = C.c_size_t(0)
    pt = C.c_int(0)
    rc = lib().pcop_oracle_voxel_downsample_ex(_f(flat), layout, n, np.float32(leaf), int(is_dense), int(variant), _f(out),
                                               C.byref(m), C.byref(pt))
    if rc:
        raise ValueError("oracle voxel_downsample rc=%d" % rc)
    return _unpack(out, layout, m.value, n), bool(pt.value)


def euclidean_clusters(xyz, tol, min_cluster, max_cluster, use_tree=True, layout=LAYOUT_AOS12):
    """-> (offsets int32[k+1], indices int32[offsets[k]])."""
    flat, m = _pack(xyz, layout)
    offsets = np.zeros(m + 1, dtype=np.int32)
    indices = np.zeros(max(m, 1), dtype=np.int32)
    k = C.c_int32(0)
    rc = lib().pcop_oracle_euclidean_clusters_ex(_f(flat), layout, m, np.float32(tol), int(min_cluster),
                                                 int(max_cluster), int(use_tree), _i(offsets), _i(indices),
                                                 C.byref(k))
    if rc:
        raise ValueError("oracle euclidean_clusters rc=%d" % rc)
    return offsets[: k.value + 1].copy(), indices[: offsets[k.value]].copy()


def cluster_order(sizes):
    sizes = np.ascontiguousarray(sizes, dtype=np.int32)
    order = np.zeros(len(sizes), dtype=np.int32)
    lib().pcop_oracle_cluster_order(_i(sizes), len(sizes), _i(order))
    return order


def cluster_summaries(xyz, offsets, indices, layout=LAYOUT_AOS12):
    flat, m = _pack(xyz, layout)
    offsets = np.ascontiguousarray(offsets, dtype=np.int32)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    k = len(offsets) - 1
    out = np.zeros((max(k, 1), 4), dtype=np.float32)
    lib().pcop_oracle_cluster_summaries(_f(flat), layout, m, _i(offsets), _i(indices), k, _f(out))
    return out[:k]


def knife_edge_pairs(xyz, tol, ulps=4):
    flat, m = _pack(xyz, LAYOUT_AOS12)
    return int(lib().pcop_oracle_knife_edge_pairs(_f(flat), LAYOUT_AOS12, m, np.float32(tol), int(ulps)))


def knife_edge_report(xyz, tol, ulps=4):
    """The margin diagnostic of SURVEY.md 8c on a cloud of any size -> (pairs with d2 within `ulps` ulp of r2,
    how many of them decide the components)."""
    flat, m = _pack(xyz, LAYOUT_AOS12)
    out = (C.c_int64 * 2)()
    rc = lib().pcop_oracle_knife_edge_report(_f(flat), LAYOUT_AOS12, m, np.float32(tol), int(ulps), out)
    if rc:
        raise ValueError("oracle knife_edge_report rc=%d" % rc)
    return int(out[0]), int(out[1])


def process_xyz(xyz, leaf, tol, min_cluster, max_cluster, is_dense=True, layout=LAYOUT_AOS12):
    """Whole chain -> dict(voxel_xyz, passthrough, offsets, indices)."""
    flat, n = _pack(xyz, layout)
    vox = np.zeros(max(3 * n, 3), dtype=np.float32)
    offsets = np.zeros(n + 1, dtype=np.int32)
    indices = np.zeros(max(n, 1), dtype=np.int32)
    m = C.c_size_t(0)
    pt = C.c_int(0)
    k = C.c_int32(0)
    rc = lib().pcop_oracle_process_xyz(_f(flat), layout, n, int(is_dense), np.float32(leaf), np.float32(tol),
                                       int(min_cluster), int(max_cluster), _f(vox), C.byref(m), C.byref(pt),
                                       _i(offsets), _i(indices), C.byref(k))
    if rc:
        raise ValueError("oracle process_xyz rc=%d" % rc)
    return dict(voxel_xyz=vox[: 3 * m.value].reshape(-1, 3).copy(), passthrough=bool(pt.value),
                offsets=offsets[: k.value + 1].copy(), indices=indices[: offsets[k.value]].copy())


def segment_planes(xyz, dist_thresh, axis=(0.0, 0.0, 1.0), eps_angle=20.0, max_planes=64):
    """segment_plane_and_extract_indices (od.cpp:342-428) -> dict(xyz what is left, last_plane its last plane's points,
    coeffs (planes, 4) refined coefficients, iterations RANSAC iterations per plane, found_all False when a round found no plane)."""
    pts = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
    n = len(pts)
    out = np.zeros(max(3 * n, 3), dtype=np.float32)
    last = np.zeros(max(3 * n, 3), dtype=np.float32)
    coeffs = np.zeros((max_planes, 4), dtype=np.float32)
    its = np.zeros(max_planes, dtype=np.int32)
    ax = np.asarray(axis, dtype=np.float32)
    n_out, n_last, n_planes = C.c_size_t(0), C.c_size_t(0), C.c_int(0)
    rc = lib().pcop_oracle_segment_planes(_f(pts), n, np.float32(dist_thresh), _f(ax), float(eps_angle), _f(out), C.byref(n_out),
                                          _f(last), C.byref(n_last), _f(coeffs), max_planes, C.byref(n_planes), _i(its))
    k = min(n_planes.value, max_planes)
    return dict(xyz=out[:3 * n_out.value].reshape(-1, 3).copy(), last_plane=last[:3 * n_last.value].reshape(-1, 3).copy(),
                coeffs=coeffs[:k].copy(), iterations=its[:k].copy(), n_planes=n_planes.value, found_all=rc == 0)


def plane_rnd(count):
    out = np.zeros(count, dtype=np.int32)
    lib().pcop_oracle_plane_rnd(_i(out), count)
    return out


def sor(xyz, mean_k, stddev_mul):
    """StatisticalOutlierRemoval -> dict(distances float32[n], threshold, keep bool[n], xyz kept points)."""
    pts = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
    n = len(pts)
    dist = np.zeros(max(n, 1), dtype=np.float32)
    keep = np.zeros(max(n, 1), dtype=np.uint8)
    thr, nk = C.c_double(0.0), C.c_size_t(0)
    rc = lib().pcop_oracle_sor(_f(pts), n, int(mean_k), np.float32(stddev_mul), _f(dist), C.byref(thr),
                               keep.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(nk))
    if rc:
        raise ValueError("pcop_oracle_sor: %d" % rc)
    keep = keep[:n].astype(bool)
    return dict(distances=dist[:n], threshold=thr.value, keep=keep, xyz=pts[keep])
