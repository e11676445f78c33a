"""ctypes binding of the C ABI in include/pcop.h (libpcop.so, HIP for gfx950).

This is plumbing for tests and bench.py; the product is the shared library.
There is no fallback: if libpcop.so is missing or no GPU is present, calls fail loudly.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PCOP_LIB", os.path.join(_HERE, "csrc", "libpcop.so"))   # PCOP_LIB: experiment builds

LAYOUT_AOS16, LAYOUT_SOA, LAYOUT_AOS12 = 0, 1, 2
MEM_HOST, MEM_DEVICE = 0, 1
MAX_STAGES = 128

EXPORTS = [
    "pcop_create", "pcop_destroy", "pcop_set_params", "pcop_get_params", "pcop_last_error", "pcop_status_string",
    "pcop_set_profiling", "pcop_get_stage_times", "pcop_stream", "pcop_voxel_downsample", "pcop_euclidean_clusters",
    "pcop_process_pointcloud2", "pcop_process_xyz", "pcop_submit_xyz", "pcop_wait", "pcop_cluster_summaries",
    "pcop_dev_alloc", "pcop_dev_free", "pcop_dev_upload", "pcop_dev_download", "pcop_device_count",
    "pcop_set_cloud_layout", "pcop_cluster_cloud", "pcop_process_node", "pcop_process_node_clouds", "pcop_process_node_pointcloud2", "pcop_segment_planes", "pcop_segment_planes_clouds", "pcop_debug_buffer", "pcop_debug_counter", "pcop_create_batch", "pcop_max_frames", "pcop_submit_batch", "pcop_wait_batch",
    "pcop_cluster_summaries_frame", "pcop_occupancy_grid_dims", "pcop_occupancy_filter", "pcop_process_filtered",
    "pcop_statistical_outlier_removal", "pcop_bind_thread_near_device", "pcop_debug_parse_cpulist",
]


class OccParams(C.Structure):
    """pcop_occ_params (include/pcop.h): limits, block size, deviation and grid size of the occupancy step."""
    _fields_ = [("x_min", C.c_float), ("x_max", C.c_float), ("y_min", C.c_float), ("y_max", C.c_float),
                ("z_min", C.c_float), ("z_max", C.c_float), ("block_size", C.c_float), ("dev_percent", C.c_float),
                ("grid_width", C.c_int32), ("grid_height", C.c_int32)]


class PcopError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("pcop status %d: %s" % (status, msg))
        self.status = status


class Params(C.Structure):
    _fields_ = [("leaf", C.c_float), ("tol", C.c_float), ("min_cluster", C.c_int32), ("max_cluster", C.c_int32),
                ("publish_clouds", C.c_int32), ("reserved", C.c_int32 * 3)]


class Result(C.Structure):
    _fields_ = [("passthrough", C.c_int32), ("num_voxels", C.c_int32), ("num_clusters", C.c_int32),
                ("num_indices", C.c_int32), ("offsets", C.POINTER(C.c_int32)), ("indices", C.POINTER(C.c_int32)),
                ("voxel_xyz", C.POINTER(C.c_float))]


class NodeStats(C.Structure):
    _fields_ = [("n_in", C.c_int32), ("n_filtered", C.c_int32), ("index_errors", C.c_int32), ("n_voxels", C.c_int32),
                ("n_inliers", C.c_int32), ("n_planes", C.c_int32), ("n_clustered", C.c_int32), ("planes_found_all", C.c_int32),
                ("threshold", C.c_double)]


class NodeClouds(C.Structure):
    """pcop_node_clouds: the clouds of the topics between the steps of the node's chain (caller's buffers + counts)."""
    _fields_ = [("voxel_grid", C.c_void_p), ("n_voxel_grid", C.c_int32), ("statistical_outliers", C.c_void_p), ("n_statistical_outliers", C.c_int32),
                ("planar_cloud", C.c_void_p), ("n_planar_cloud", C.c_int32), ("indices_cloud", C.c_void_p), ("n_indices_cloud", C.c_int32),
                ("cloud_f", C.c_void_p), ("n_cloud_f", C.c_int32)]


class PlaneParams(C.Structure):
    """pcop_plane_params: distance threshold, axis and angle of the RANSAC plane removal (od.cpp:110-111, 768)."""
    _fields_ = [("dist_thresh", C.c_float), ("axis", C.c_float * 3), ("eps_angle", C.c_double)]


def plane_params(dist_thresh, axis=(0.0, 0.0, 1.0), eps_angle=20.0):
    return PlaneParams(np.float32(dist_thresh), (C.c_float * 3)(*[float(a) for a in axis]), float(eps_angle))


class StageTimes(C.Structure):
    _fields_ = [("count", C.c_int32), ("name", C.c_char_p * MAX_STAGES), ("ms", C.c_float * MAX_STAGES)]


class RadixPlan(C.Structure):
    _fields_ = [("key_bits", C.c_uint32), ("npass", C.c_uint32), ("bits", C.c_uint32 * 4), ("shift", C.c_uint32 * 4)]


class DevState(C.Structure):
    """Mirror of pcop::DevState (csrc/pcop_device.hpp) for the stage-wise tests."""
    _fields_ = [("err", C.c_uint32), ("n_in", C.c_uint32), ("is_dense", C.c_uint32), ("n_valid", C.c_uint32),
                ("passthrough", C.c_uint32), ("m", C.c_uint32), ("kind", C.c_uint32), ("trivial", C.c_uint32),
                ("emit_cloud", C.c_uint32), ("out_layout", C.c_uint32), ("out_soa_stride", C.c_uint32),
                ("tile", C.c_uint32), ("inv_leaf", C.c_float), ("vmin", C.c_float * 3), ("vmax", C.c_float * 3),
                ("min_b", C.c_int32 * 3), ("div_b", C.c_int32 * 3), ("mul", C.c_uint32 * 3), ("plan_voxel", RadixPlan),
                ("src_x", C.c_void_p), ("src_y", C.c_void_p), ("src_z", C.c_void_p), ("inv_cell", C.c_double),
                ("c_n", C.c_uint32), ("c_from_input", C.c_uint32), ("r2", C.c_float), ("min_cluster", C.c_uint32),
                ("max_cluster", C.c_uint32), ("hash_mask", C.c_uint32), ("n_cells", C.c_uint32),
                ("n_kept", C.c_uint32), ("n_kept_pts", C.c_uint32), ("plan_csr", RadixPlan), ("debug_skip_union", C.c_uint32),
                ("vpath", C.c_uint32), ("gbits", C.c_uint32), ("cshift", C.c_uint32), ("nbuckets", C.c_uint32), ("nb_cap", C.c_uint32),
                ("cgrid_req", C.c_uint32), ("cpath", C.c_uint32), ("cmap_words", C.c_uint32), ("c_rowbits", C.c_uint32),
                ("c_plane", C.c_uint32), ("c0", C.c_int32 * 3), ("c_dim", C.c_uint32 * 3), ("maxbin", C.c_uint32),
                ("box_nonfinite", C.c_uint32), ("pad1", C.c_uint32), ("spin_limit", C.c_uint32), ("box_tile", C.c_uint32), ("sbox", C.c_uint32)]


_lib = None


def load_library():
    """Load libpcop.so; raises if the HIP extension has not been built (no CPU stand-in exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libpcop.so not built: run `make -C %s` (or __graft_entry__.build())" %
                           os.path.dirname(LIB_PATH))
    try:  # a process that also uses torch must load torch's HIP runtime first (same soname)
        import torch  # noqa: F401
    except Exception:
        pass
    L = C.CDLL(LIB_PATH)
    vp, fp, ip = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int32)
    L.pcop_create.restype = C.c_int
    L.pcop_create.argtypes = [C.POINTER(vp), C.c_int, C.c_size_t]
    L.pcop_create_batch.restype = C.c_int
    L.pcop_create_batch.argtypes = [C.POINTER(vp), C.c_int, C.c_size_t, C.c_int]
    L.pcop_max_frames.restype = C.c_int
    L.pcop_max_frames.argtypes = [vp]
    L.pcop_submit_batch.restype = C.c_int
    L.pcop_submit_batch.argtypes = [vp, C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_size_t), C.c_int, C.c_int, C.c_int]
    L.pcop_wait_batch.restype = C.c_int
    L.pcop_wait_batch.argtypes = [vp, C.POINTER(Result), C.c_int]
    L.pcop_cluster_summaries_frame.restype = C.c_int
    L.pcop_cluster_summaries_frame.argtypes = [vp, C.c_int, fp, C.c_int32]
    L.pcop_set_cloud_layout.restype = C.c_int
    L.pcop_set_cloud_layout.argtypes = [vp, C.c_int]
    L.pcop_cluster_cloud.restype = C.c_int
    L.pcop_cluster_cloud.argtypes = [vp, C.c_int, C.POINTER(fp), ip]
    L.pcop_process_node.restype = C.c_int
    L.pcop_process_node.argtypes = [vp, C.c_void_p, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.POINTER(Result), C.POINTER(NodeStats)]
    L.pcop_process_node_clouds.restype = C.c_int
    L.pcop_process_node_clouds.argtypes = [vp, C.c_void_p, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.POINTER(Result), C.POINTER(NodeStats), C.c_void_p]
    L.pcop_process_node_pointcloud2.restype = C.c_int
    L.pcop_process_node_pointcloud2.argtypes = [vp, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                                C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p,
                                                C.POINTER(Result), C.POINTER(NodeStats), C.c_void_p]
    L.pcop_segment_planes_clouds.restype = C.c_int
    L.pcop_segment_planes_clouds.argtypes = [vp, C.c_void_p, C.c_int, C.c_size_t, C.POINTER(PlaneParams), C.c_void_p, C.POINTER(C.c_size_t),
                                             C.c_void_p, C.POINTER(C.c_size_t), C.c_void_p, C.POINTER(C.c_size_t), C.c_void_p, C.c_int32, ip, ip, C.c_void_p]
    L.pcop_segment_planes.restype = C.c_int
    L.pcop_segment_planes.argtypes = [vp, C.c_void_p, C.c_int, C.c_size_t, C.POINTER(PlaneParams), C.c_void_p, C.POINTER(C.c_size_t),
                                      C.c_void_p, C.POINTER(C.c_size_t), C.c_void_p, C.c_int32, ip, ip]
    L.pcop_occupancy_grid_dims.restype = None
    L.pcop_occupancy_grid_dims.argtypes = [C.c_float] * 5 + [C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.pcop_occupancy_filter.restype = C.c_int
    L.pcop_occupancy_filter.argtypes = [vp, C.c_void_p, C.c_int, C.c_size_t, C.c_int, C.POINTER(OccParams), C.c_void_p,
                                        C.POINTER(C.c_size_t), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]
    L.pcop_statistical_outlier_removal.restype = C.c_int
    L.pcop_statistical_outlier_removal.argtypes = [vp, C.c_void_p, C.c_int, C.c_size_t, C.c_int, C.c_float, C.c_void_p,
                                                   C.POINTER(C.c_size_t), C.c_void_p, C.POINTER(C.c_double)]
    L.pcop_process_filtered.restype = C.c_int
    L.pcop_process_filtered.argtypes = [vp, C.POINTER(Result)]
    L.pcop_destroy.restype = None
    L.pcop_destroy.argtypes = [vp]
    L.pcop_set_params.restype = C.c_int
    L.pcop_set_params.argtypes = [vp, C.POINTER(Params)]
    L.pcop_get_params.restype = C.c_int
    L.pcop_get_params.argtypes = [vp, C.POINTER(Params)]
    L.pcop_last_error.restype = C.c_char_p
    L.pcop_last_error.argtypes = [vp]
    L.pcop_status_string.restype = C.c_char_p
    L.pcop_status_string.argtypes = [C.c_int]
    L.pcop_set_profiling.restype = C.c_int
    L.pcop_set_profiling.argtypes = [vp, C.c_int]
    L.pcop_get_stage_times.restype = C.c_int
    L.pcop_get_stage_times.argtypes = [vp, C.POINTER(StageTimes)]
    L.pcop_stream.restype = vp
    L.pcop_stream.argtypes = [vp]
    L.pcop_voxel_downsample.restype = C.c_int
    L.pcop_voxel_downsample.argtypes = [vp, fp, C.c_int, C.c_size_t, C.c_int, fp, C.POINTER(C.c_size_t),
                                        C.POINTER(C.c_int)]
    L.pcop_euclidean_clusters.restype = C.c_int
    L.pcop_euclidean_clusters.argtypes = [vp, fp, C.c_int, C.c_size_t, ip, ip, ip]
    L.pcop_process_pointcloud2.restype = C.c_int
    L.pcop_process_pointcloud2.argtypes = [vp, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                           C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.POINTER(Result)]
    L.pcop_process_xyz.restype = C.c_int
    L.pcop_process_xyz.argtypes = [vp, C.c_void_p, C.c_int, C.c_size_t, C.c_int, C.c_int, C.POINTER(Result)]
    L.pcop_submit_xyz.restype = C.c_int
    L.pcop_submit_xyz.argtypes = [vp, C.c_void_p, C.c_int, C.c_size_t, C.c_int, C.c_int]
    L.pcop_wait.restype = C.c_int
    L.pcop_wait.argtypes = [vp, C.POINTER(Result)]
    L.pcop_cluster_summaries.restype = C.c_int
    L.pcop_cluster_summaries.argtypes = [vp, fp, C.c_int32]
    L.pcop_dev_alloc.restype = C.c_int
    L.pcop_dev_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    L.pcop_dev_free.restype = C.c_int
    L.pcop_dev_free.argtypes = [vp, vp]
    L.pcop_dev_upload.restype = C.c_int
    L.pcop_dev_upload.argtypes = [vp, vp, C.c_void_p, C.c_size_t]
    L.pcop_dev_download.restype = C.c_int
    L.pcop_dev_download.argtypes = [vp, C.c_void_p, vp, C.c_size_t]
    L.pcop_device_count.restype = C.c_int
    L.pcop_device_count.argtypes = []
    L.pcop_bind_thread_near_device.restype = C.c_int
    L.pcop_bind_thread_near_device.argtypes = [C.c_int]
    L.pcop_debug_parse_cpulist.restype = C.c_int
    L.pcop_debug_parse_cpulist.argtypes = [C.c_char_p, C.POINTER(C.c_int32), C.c_int32]
    L.pcop_debug_counter.restype = C.c_int
    L.pcop_debug_counter.argtypes = [vp, C.c_char_p, C.POINTER(C.c_uint64)]
    L.pcop_debug_buffer.restype = C.c_int
    L.pcop_debug_buffer.argtypes = [vp, C.c_char_p, C.POINTER(vp), C.POINTER(C.c_size_t)]
    _lib = L
    return L


def bind_thread_near_device(device=0):
    """pcop_bind_thread_near_device: the calling thread (and those it starts) onto the CPUs next to the GPU; the number of CPUs, 0 = unchanged."""
    rc = load_library().pcop_bind_thread_near_device(int(device))
    if rc < 0:
        raise PcopError(rc, "pcop_bind_thread_near_device")
    return rc


def pack_points(xyz, layout):
    """(n,3) float32 -> flat float32 array in `layout`."""
    xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
    n = xyz.shape[0]
    if layout == LAYOUT_AOS12:
        return xyz.reshape(-1).copy(), n
    if layout == LAYOUT_SOA:
        return np.ascontiguousarray(xyz.T).reshape(-1).copy(), n
    out = np.ones((n, 4), dtype=np.float32)
    out[:, :3] = xyz
    return out.reshape(-1), n


def unpack_points(flat, layout, m, stride):
    if layout == LAYOUT_AOS12:
        return flat[:3 * m].reshape(m, 3).copy()
    if layout == LAYOUT_SOA:
        return np.stack([flat[0:m], flat[stride:stride + m], flat[2 * stride:2 * stride + m]], axis=1).copy()
    return flat[:4 * m].reshape(m, 4)[:, :3].copy()


class Context:
    """One GPU context (pcop_ctx)."""

    def __init__(self, max_points, device=0, profiling=False, max_frames=1):
        self.cloud_layout = LAYOUT_AOS12
        self.lib = load_library()
        self.h = C.c_void_p()
        self.max_frames = int(max_frames)
        rc = self.lib.pcop_create_batch(C.byref(self.h), int(device), int(max_points), int(max_frames))
        if rc:
            raise PcopError(rc, self.lib.pcop_status_string(rc).decode())
        self.max_points = int(max_points)
        if profiling:
            self.lib.pcop_set_profiling(self.h, 1)

    def close(self):
        if self.h:
            self.lib.pcop_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc:
            raise PcopError(rc, "%s (%s)" % (self.lib.pcop_status_string(rc).decode(),
                                             self.lib.pcop_last_error(self.h).decode()))

    def set_params(self, leaf=0.015, tol=0.4, min_cluster=5, max_cluster=20000, publish_clouds=1):
        p = Params(np.float32(leaf), np.float32(tol), int(min_cluster), int(max_cluster), int(publish_clouds))
        self._chk(self.lib.pcop_set_params(self.h, C.byref(p)))

    def set_params_struct(self, p):
        self._chk(self.lib.pcop_set_params(self.h, C.byref(p)))

    # ---- stage API --------------------------------------------------------------------
    def voxel_downsample(self, xyz, is_dense=True, layout=LAYOUT_AOS12):
        flat, n = pack_points(xyz, layout)
        out = np.zeros(max(4 * n, 4), dtype=np.float32)
        m, pt = C.c_size_t(0), C.c_int(0)
        self._chk(self.lib.pcop_voxel_downsample(self.h, flat.ctypes.data_as(C.POINTER(C.c_float)), layout, n,
                                                 int(is_dense), out.ctypes.data_as(C.POINTER(C.c_float)),
                                                 C.byref(m), C.byref(pt)))
        return unpack_points(out, layout, m.value, n), bool(pt.value)

    def euclidean_clusters(self, xyz, layout=LAYOUT_AOS12):
        flat, m = pack_points(xyz, layout)
        offsets = np.zeros(m + 1, dtype=np.int32)
        indices = np.zeros(max(m, 1), dtype=np.int32)
        k = C.c_int32(0)
        ip = C.POINTER(C.c_int32)
        self._chk(self.lib.pcop_euclidean_clusters(self.h, flat.ctypes.data_as(C.POINTER(C.c_float)), layout, m,
                                                   offsets.ctypes.data_as(ip), indices.ctypes.data_as(ip),
                                                   C.byref(k)))
        return offsets[:k.value + 1].copy(), indices[:offsets[k.value]].copy()

    # ---- frame API --------------------------------------------------------------------
    def set_cloud_layout(self, layout):
        """Layout of the frame API's voxel cloud: LAYOUT_AOS12 (default) or LAYOUT_AOS16 = pcl::PointXYZ records {x,y,z,1}."""
        self._chk(self.lib.pcop_set_cloud_layout(self.h, int(layout)))
        self.cloud_layout = int(layout)

    def cluster_cloud(self, frame=0):
        """The euc_clusters cloud of a finished frame -> (n, 4) float32 records {x, y, z, 1.0}, gathered on the GPU."""
        p, n = C.POINTER(C.c_float)(), C.c_int32(0)
        self._chk(self.lib.pcop_cluster_cloud(self.h, int(frame), C.byref(p), C.byref(n)))
        if n.value == 0:
            return np.zeros((0, 4), dtype=np.float32)
        return np.ctypeslib.as_array(p, shape=(n.value * 4,)).reshape(n.value, 4).copy()

    def _result(self, r):
        k, t, m = r.num_clusters, r.num_indices, r.num_voxels
        offsets = np.ctypeslib.as_array(r.offsets, shape=(k + 1,)).copy()
        indices = np.ctypeslib.as_array(r.indices, shape=(max(t, 1),))[:t].copy()
        vox = None
        if r.voxel_xyz:
            w = 4 if self.cloud_layout == LAYOUT_AOS16 else 3
            vox = np.ctypeslib.as_array(r.voxel_xyz, shape=(max(m, 1) * w,))[:w * m].reshape(m, w).copy()
        return dict(passthrough=bool(r.passthrough), num_voxels=m, offsets=offsets, indices=indices, voxel_xyz=vox)

    def process_xyz(self, xyz, is_dense=True, layout=LAYOUT_AOS12):
        flat, n = pack_points(xyz, layout)
        r = Result()
        self._chk(self.lib.pcop_process_xyz(self.h, flat.ctypes.data, layout, n, int(is_dense), MEM_HOST, C.byref(r)))
        return self._result(r)

    def occupancy_grid_dims(self, x_min, x_max, y_min, y_max, block_size):
        w, h = C.c_int32(0), C.c_int32(0)
        self.lib.pcop_occupancy_grid_dims(C.c_float(x_min), C.c_float(x_max), C.c_float(y_min), C.c_float(y_max),
                                          C.c_float(block_size), C.byref(w), C.byref(h))
        return w.value, h.value

    def occupancy_filter(self, xyz, occ, layout=LAYOUT_AOS12, want_cloud=True):
        """Bounds filter + occupancy binning -> dict(xyz, counts, row_avg, grid, index_errors); the kept cloud
        also stays resident for process_filtered()."""
        flat, n = pack_points(xyz, layout)
        cells, rows = occ.grid_width * occ.grid_height, occ.grid_height
        out = np.zeros(max(3 * n, 3), dtype=np.float32)
        counts = np.zeros(max(cells, 1), dtype=np.int64)
        row_avg = np.zeros(max(rows, 1), dtype=np.int64)
        grid = np.zeros(max(cells, 1), dtype=np.int8)
        n_out, bad = C.c_size_t(0), C.c_int32(0)
        self._chk(self.lib.pcop_occupancy_filter(self.h, flat.ctypes.data, layout, n, MEM_HOST, C.byref(occ),
                                                 out.ctypes.data if want_cloud else None, C.byref(n_out),
                                                 counts.ctypes.data, row_avg.ctypes.data, grid.ctypes.data, C.byref(bad)))
        k = n_out.value
        return dict(xyz=out[:3 * k].reshape(k, 3).copy() if want_cloud else None, n_out=k, counts=counts[:cells],
                    row_avg=row_avg[:rows], grid=grid[:cells], index_errors=bad.value)

    def occupancy_filter_device(self, dev_ptr, n, occ, layout=LAYOUT_SOA):
        """The same on a cloud already resident in HBM; returns (kept points, index errors) only."""
        n_out, bad = C.c_size_t(0), C.c_int32(0)
        self._chk(self.lib.pcop_occupancy_filter(self.h, C.c_void_p(dev_ptr), layout, int(n), MEM_DEVICE, C.byref(occ),
                                                 None, C.byref(n_out), None, None, None, C.byref(bad)))
        return n_out.value, bad.value

    def statistical_outlier_removal(self, xyz, mean_k, stddev_mul, layout=LAYOUT_AOS12):
        """remove_statistical_outliers -> dict(xyz kept points, distances float32[n], threshold)."""
        flat, n = pack_points(xyz, layout)
        out = np.zeros_like(flat) if n else np.zeros(3, dtype=np.float32)
        dist = np.zeros(max(n, 1), dtype=np.float32)
        n_out, thr = C.c_size_t(0), C.c_double(0.0)
        self._chk(self.lib.pcop_statistical_outlier_removal(self.h, flat.ctypes.data, layout, n, int(mean_k), C.c_float(stddev_mul),
                                                            out.ctypes.data, C.byref(n_out), dist.ctypes.data, C.byref(thr)))
        return dict(xyz=unpack_points(out, layout, n_out.value, n), distances=dist[:n], threshold=thr.value)

    def segment_planes(self, xyz, plane, layout=LAYOUT_AOS12, cap_planes=64):
        """segment_plane_and_extract_indices -> dict(xyz what is left, coeffs (planes, 4), n_planes, found_all)."""
        flat, n = pack_points(xyz, layout)
        out = np.zeros_like(flat) if n else np.zeros(3, dtype=np.float32)
        last = np.zeros_like(out)
        coeffs = np.zeros((cap_planes, 4), dtype=np.float32)
        before = np.zeros_like(out)
        last_c = np.zeros(4, dtype=np.float32)
        n_out, n_last, n_before, n_planes, found = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0), C.c_int32(0), C.c_int32(0)
        self._chk(self.lib.pcop_segment_planes_clouds(self.h, flat.ctypes.data, layout, n, C.byref(plane), out.ctypes.data, C.byref(n_out),
                                                      last.ctypes.data, C.byref(n_last), before.ctypes.data, C.byref(n_before),
                                                      coeffs.ctypes.data, cap_planes, C.byref(n_planes), C.byref(found), last_c.ctypes.data))
        return dict(xyz=unpack_points(out, layout, n_out.value, n), last_plane=unpack_points(last, layout, n_last.value, n),
                    cloud_f=unpack_points(before, layout, n_before.value, n), last_coeffs=last_c,
                    coeffs=coeffs[:min(n_planes.value, cap_planes)].copy(), n_planes=n_planes.value, found_all=bool(found.value))

    def process_node(self, xyz, occ=None, mean_k=0, stddev_mul=1.0, plane=None, is_dense=True, layout=LAYOUT_AOS12, dev_ptr=None,
                     n=None, want_grid=False):
        """The node's chain in one call (od.cpp:727-797): [bounds filter + occupancy binning] -> voxel grid ->
        [statistical outlier removal] -> [RANSAC plane removal] -> clusters.  xyz: (n,3) host array, or dev_ptr/n for SoA planes already in HBM.
        -> dict of the frame result + stats (+ counts, row_avg, grid with want_grid)."""
        if dev_ptr is None:
            flat, n = pack_points(xyz, layout)
            ptr, mem = flat.ctypes.data, MEM_HOST
        else:
            ptr, mem, layout = dev_ptr, MEM_DEVICE, LAYOUT_SOA
        counts = row_avg = grid = None
        if occ is not None and want_grid:
            cells, rows = occ.grid_width * occ.grid_height, occ.grid_height
            counts = np.zeros(max(cells, 1), dtype=np.int64)
            row_avg = np.zeros(max(rows, 1), dtype=np.int64)
            grid = np.zeros(max(cells, 1), dtype=np.int8)
        r, st = Result(), NodeStats()
        self._chk(self.lib.pcop_process_node(self.h, C.c_void_p(ptr), layout, int(n), int(is_dense), mem,
                                             C.cast(C.byref(occ), C.c_void_p) if occ is not None else None,
                                             counts.ctypes.data if counts is not None else None,
                                             row_avg.ctypes.data if row_avg is not None else None,
                                             grid.ctypes.data if grid is not None else None,
                                             int(mean_k), C.c_float(stddev_mul),
                                             C.cast(C.byref(plane), C.c_void_p) if plane is not None else None, C.byref(r), C.byref(st)))
        out = self._result(r)
        out.update(n_filtered=st.n_filtered, index_errors=st.index_errors, n_voxels=st.n_voxels, n_inliers=st.n_inliers,
                   n_planes=st.n_planes, n_clustered=st.n_clustered, planes_found_all=bool(st.planes_found_all), threshold=st.threshold)
        if counts is not None:
            out.update(counts=counts[:occ.grid_width * occ.grid_height], row_avg=row_avg[:occ.grid_height],
                       grid=grid[:occ.grid_width * occ.grid_height])
        return out

    def process_node_pointcloud2(self, data, width, height, point_step, row_step, off_x, off_y, off_z, is_bigendian=False, is_dense=True,
                                 occ=None, mean_k=0, stddev_mul=1.0, plane=None, want_clouds=False):
        """The node's chain fed with the PointCloud2 blob itself (pcop_process_node_pointcloud2).  want_clouds: also the five clouds the
        node publishes between the steps, each as (k, 4) float32 records or None."""
        buf = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
        n = int(width) * int(height)
        r, st = Result(), NodeStats()
        cl, bufs = None, None
        if want_clouds:
            cl = NodeClouds()
            bufs = [np.zeros(4 * max(n, 1), dtype=np.float32) for _ in range(5)]
            cl.voxel_grid, cl.statistical_outliers, cl.planar_cloud, cl.indices_cloud, cl.cloud_f = [b.ctypes.data for b in bufs]
        self._chk(self.lib.pcop_process_node_pointcloud2(self.h, buf.ctypes.data, width, height, point_step, row_step, off_x, off_y, off_z,
                                                         int(is_bigendian), int(is_dense),
                                                         C.cast(C.byref(occ), C.c_void_p) if occ is not None else None, None, None, None,
                                                         int(mean_k), C.c_float(stddev_mul),
                                                         C.cast(C.byref(plane), C.c_void_p) if plane is not None else None, C.byref(r), C.byref(st),
                                                         C.cast(C.byref(cl), C.c_void_p) if cl is not None else None))
        out = self._result(r)
        out.update(n_filtered=st.n_filtered, index_errors=st.index_errors, n_voxels=st.n_voxels, n_inliers=st.n_inliers,
                   n_planes=st.n_planes, n_clustered=st.n_clustered, planes_found_all=bool(st.planes_found_all), threshold=st.threshold)
        if cl is not None:
            cnt = [cl.n_voxel_grid, cl.n_statistical_outliers, cl.n_planar_cloud, cl.n_indices_cloud, cl.n_cloud_f]
            names = ["voxel_grid", "statistical_outliers", "planar_cloud", "indices_cloud", "cloud_f"]
            out["clouds"] = {nm: (b[:4 * k].reshape(k, 4).copy() if k >= 0 else None) for nm, b, k in zip(names, bufs, cnt)}
        return out

    def process_filtered(self):
        r = Result()
        self._chk(self.lib.pcop_process_filtered(self.h, C.byref(r)))
        return self._result(r)

    def process_device(self, dev_ptr, n, is_dense=True, layout=LAYOUT_SOA):
        r = Result()
        self._chk(self.lib.pcop_process_xyz(self.h, C.c_void_p(dev_ptr), layout, int(n), int(is_dense), MEM_DEVICE,
                                            C.byref(r)))
        return self._result(r)

    def submit_device(self, dev_ptr, n, is_dense=True, layout=LAYOUT_SOA):
        self._chk(self.lib.pcop_submit_xyz(self.h, C.c_void_p(dev_ptr), layout, int(n), int(is_dense), MEM_DEVICE))

    def submit_batch_device(self, dev_ptrs, ns, is_dense=True, layout=LAYOUT_SOA):
        """Enqueue len(dev_ptrs) device-resident clouds as one batch."""
        k = len(dev_ptrs)
        ptrs = (C.c_void_p * k)(*[C.c_void_p(p) for p in dev_ptrs])
        cnt = (C.c_size_t * k)(*[int(n) for n in ns])
        self._chk(self.lib.pcop_submit_batch(self.h, ptrs, layout, cnt, k, int(is_dense), MEM_DEVICE))
        self._batch = k

    def submit_batch_host(self, clouds, is_dense=True, layout=LAYOUT_AOS12):
        k = len(clouds)
        packed = [pack_points(c, layout) for c in clouds]
        self._keep = packed
        ptrs = (C.c_void_p * k)(*[C.c_void_p(p[0].ctypes.data) for p in packed])
        cnt = (C.c_size_t * k)(*[p[1] for p in packed])
        self._chk(self.lib.pcop_submit_batch(self.h, ptrs, layout, cnt, k, int(is_dense), MEM_HOST))
        self._batch = k

    def wait_batch(self, want_arrays=True):
        k = self._batch
        res = (Result * k)()
        self._chk(self.lib.pcop_wait_batch(self.h, res, k))
        if not want_arrays:
            return [dict(passthrough=bool(r.passthrough), num_voxels=r.num_voxels, num_clusters=r.num_clusters,
                         num_indices=r.num_indices) for r in res]
        return [self._result(r) for r in res]

    def cluster_summaries_frame(self, frame, k):
        out = np.zeros((max(k, 1), 4), dtype=np.float32)
        self._chk(self.lib.pcop_cluster_summaries_frame(self.h, int(frame), out.ctypes.data_as(C.POINTER(C.c_float)), int(k)))
        return out[:k]

    def wait(self, want_arrays=True):
        r = Result()
        self._chk(self.lib.pcop_wait(self.h, C.byref(r)))
        if not want_arrays:
            return dict(passthrough=bool(r.passthrough), num_voxels=r.num_voxels, num_clusters=r.num_clusters,
                        num_indices=r.num_indices)
        return self._result(r)

    def process_pointcloud2(self, data, width, height, point_step, row_step, off_x, off_y, off_z,
                            is_bigendian=False, is_dense=True):
        buf = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
        r = Result()
        self._chk(self.lib.pcop_process_pointcloud2(self.h, buf.ctypes.data, width, height, point_step, row_step,
                                                    off_x, off_y, off_z, int(is_bigendian), int(is_dense),
                                                    C.byref(r)))
        return self._result(r)

    def cluster_summaries(self, k):
        out = np.zeros((max(k, 1), 4), dtype=np.float32)
        self._chk(self.lib.pcop_cluster_summaries(self.h, out.ctypes.data_as(C.POINTER(C.c_float)), int(k)))
        return out[:k]

    # ---- device memory helpers ----------------------------------------------------------
    def dev_alloc(self, nbytes):
        p = C.c_void_p()
        self._chk(self.lib.pcop_dev_alloc(self.h, int(nbytes), C.byref(p)))
        return p.value

    def dev_free(self, ptr):
        self._chk(self.lib.pcop_dev_free(self.h, C.c_void_p(ptr)))

    def upload_soa(self, xyz):
        """(n,3) float32 -> device pointer of SoA planes x[n] y[n] z[n]."""
        flat, n = pack_points(xyz, LAYOUT_SOA)
        ptr = self.dev_alloc(flat.nbytes)
        self._chk(self.lib.pcop_dev_upload(self.h, C.c_void_p(ptr), flat.ctypes.data, flat.nbytes))
        return ptr, n

    def debug_buffer(self, name, dtype, count=None):
        p, b = C.c_void_p(), C.c_size_t()
        self._chk(self.lib.pcop_debug_buffer(self.h, name.encode(), C.byref(p), C.byref(b)))
        if count is None:   # the whole buffer
            count = b.value // np.dtype(dtype).itemsize
        out = np.zeros(count, dtype=dtype)
        assert out.nbytes <= b.value
        if out.nbytes:
            self._chk(self.lib.pcop_dev_download(self.h, out.ctypes.data, p, out.nbytes))
        return out

    def debug_counter(self, name):
        v = C.c_uint64(0)
        self._chk(self.lib.pcop_debug_counter(self.h, name.encode(), C.byref(v)))
        return int(v.value)

    def debug_state(self):
        p, b = C.c_void_p(), C.c_size_t()
        self._chk(self.lib.pcop_debug_buffer(self.h, b"state", C.byref(p), C.byref(b)))
        st = DevState()
        if os.environ.get("PCOP_LIB") is None:   # (an experiment build may lay the tail of the struct out differently: its head is read)
            assert C.sizeof(st) == b.value, (C.sizeof(st), b.value)
        self._chk(self.lib.pcop_dev_download(self.h, C.addressof(st), p, min(C.sizeof(st), b.value)))
        return st

    def stage_times(self):
        t = StageTimes()
        self._chk(self.lib.pcop_get_stage_times(self.h, C.byref(t)))
        return [(t.name[i].decode(), float(t.ms[i])) for i in range(t.count)]
