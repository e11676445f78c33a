"""CPU-side tests of the C++ host mirror (pointcloud_obstacle_processing_amd/host): the params.yaml
tree of od.cpp:940-975, PointCloud2 field lookup / conversion, and the harness file format."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from pointcloud_obstacle_processing_amd import pc2io, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "pointcloud_obstacle_processing_amd", "host")
REF_YAML = "/root/reference/minibot_cr18/params.yaml"


class HostParams(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("accumulate_count", "downsample_input_data", "passthrough_filter_enable",
                                      "publish_point_clouds", "grid_opacity")] + \
               [(n, C.c_float) for n in ("block_size", "dev_percent", "downsample_size")] + \
               [("statistical_outlier_meanK", C.c_int), ("statistical_outlier_stdDevThres", C.c_float),
                ("plane_segment_dist_thres", C.c_float), ("plane_segment_angle", C.c_int),
                ("euc_cluster_tolerance", C.c_float), ("euc_min_cluster_size", C.c_int),
                ("euc_max_cluster_size", C.c_int)] + \
               [(n, C.c_float) for n in ("convex_hull_alpha", "x_min", "x_max", "y_min", "y_max", "z_min", "z_max")]


@pytest.fixture(scope="module")
def hostlib():
    so = os.path.join(HOST, "libpcop_host.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", HOST, "libpcop_host.so"])
    L = C.CDLL(so)
    L.pcop_host_load_params.restype = C.c_int
    L.pcop_host_load_params.argtypes = [C.c_char_p, C.POINTER(HostParams)]
    L.pcop_host_read_cloud_xyz.restype = C.c_long
    L.pcop_host_read_cloud_xyz.argtypes = [C.c_char_p, C.POINTER(C.c_float), C.c_long, C.POINTER(C.c_int), C.POINTER(C.c_uint)]
    return L


def test_params_tree_c1(hostlib):
    p = HostParams()
    assert hostlib.pcop_host_load_params(os.path.join(ROOT, "tests", "golden", "params_c1.yaml").encode(), C.byref(p)) == 0
    assert np.float32(p.downsample_size) == np.float32(0.05) and np.float32(p.euc_cluster_tolerance) == np.float32(0.1)
    assert (p.euc_min_cluster_size, p.euc_max_cluster_size, p.accumulate_count) == (5, 20000, 1)
    assert (np.float32(p.x_max), np.float32(p.y_max), np.float32(p.z_min)) == (np.float32(4.5), np.float32(3.78), np.float32(-0.5))
    assert p.statistical_outlier_stdDevThres == 4.0            # int in yaml -> float param
    assert p.plane_segment_angle == 20                          # double in yaml -> int param, rounded like roscpp
    # keys absent from the file keep the defaults hard-coded in od.cpp
    assert np.float32(p.block_size) == np.float32(0.15) and p.statistical_outlier_meanK == 15
    assert p.publish_point_clouds == 1


def test_missing_file_is_an_error(hostlib):
    assert hostlib.pcop_host_load_params(b"/nonexistent/params.yaml", C.byref(HostParams())) == -1


@pytest.mark.skipif(not os.path.exists(REF_YAML), reason="reference checkout not present")
def test_params_tree_of_the_reference_file(hostlib):
    """The reference's own params.yaml parses to the values it holds (params.yaml:2-31)."""
    p = HostParams()
    assert hostlib.pcop_host_load_params(REF_YAML.encode(), C.byref(p)) == 0
    assert p.accumulate_count == 200 and np.float32(p.downsample_size) == np.float32(0.015)
    assert np.float32(p.euc_cluster_tolerance) == np.float32(0.4)
    assert (p.euc_min_cluster_size, p.euc_max_cluster_size) == (5, 20000)
    assert np.float32(p.block_size) == np.float32(0.0375) and np.float32(p.dev_percent) == np.float32(0.9)
    assert np.float32(p.statistical_outlier_stdDevThres) == np.float32(4.0) and p.plane_segment_angle == 20
    assert (np.float32(p.x_min), np.float32(p.x_max), np.float32(p.y_max), np.float32(p.z_max)) == \
           (np.float32(0.0), np.float32(4.5), np.float32(3.78), np.float32(0.25))
    assert p.downsample_input_data == 1      # the yaml key is misspelled "downsame_input_data": default stays


def test_pointcloud2_file_roundtrip_and_field_lookup(hostlib, tmp_path):
    pts = synth.uniform_field(1000, 3)
    pts[7] = np.nan
    path = str(tmp_path / "cloud.pc2")
    # Kinect-style layout: xyz not at 0/4/8, 32-byte points, an extra rgb field
    pc2io.write_pointcloud2(path, pts, point_step=32, offsets=(8, 4, 16), is_dense=False, width=50, height=20,
                            extra_fields=[("rgb", 24, 7, 1)])
    out = np.zeros(3 * 1000, dtype=np.float32)
    dense = C.c_int(-1)
    off = (C.c_uint * 3)()
    n = hostlib.pcop_host_read_cloud_xyz(path.encode(), out.ctypes.data_as(C.POINTER(C.c_float)), 1000, C.byref(dense), off)
    assert n == 1000 and dense.value == 0 and list(off) == [8, 4, 16]
    assert out.reshape(-1, 3).tobytes() == pts.tobytes()
    back = pc2io.read_pointcloud2(path)
    assert back["xyz"].tobytes() == pts.tobytes() and back["point_step"] == 32


def test_missing_xyz_field_is_rejected(hostlib, tmp_path):
    path = str(tmp_path / "bad.pc2")
    pc2io.write_pointcloud2(path, synth.uniform_field(10, 1))
    raw = open(path, "rb").read().replace(b"\nz 8 7 1\n", b"\nw 8 7 1\n")
    open(path, "wb").write(raw)
    out = np.zeros(30, dtype=np.float32)
    assert hostlib.pcop_host_read_cloud_xyz(path.encode(), out.ctypes.data_as(C.POINTER(C.c_float)), 10, None, None) == -2


def test_corrupt_cloud_files_are_refused_not_followed(hostlib, tmp_path):
    """A field offset near 2^32 must not wrap past the offset check (off + 4 did), and counts in the header that no file can hold
    are a corrupt header, not an allocation to attempt."""
    path = str(tmp_path / "cloud.pc2")
    pc2io.write_pointcloud2(path, synth.uniform_field(10, 1))
    good = open(path, "rb").read()
    out = np.zeros(30, dtype=np.float32)
    outp = out.ctypes.data_as(C.POINTER(C.c_float))
    assert hostlib.pcop_host_read_cloud_xyz(path.encode(), outp, 10, None, None) == 10
    for patch, want in (((b"\nz 8 7 1\n", b"\nz 4294967294 7 1\n"), -2),          # offset + 4 wraps to 2
                        ((b"\nz 8 7 1\n", b"\nz 4294967293 7 1\n"), -2),
                        ((b"\nfields 3\n", b"\nfields 99999999999999\n"), -1),
                        ((b"\ndata 160\n", b"\ndata 999999999999999\n"), -1),
                        ((b"\ndata 160\n", b"\ndata 161\n"), -1)):                 # one byte more than the file holds
        assert patch[0] in good, patch[0]
        open(path, "wb").write(good.replace(*patch))
        assert hostlib.pcop_host_read_cloud_xyz(path.encode(), outp, 10, None, None) == want, patch[1]
