"""The ROS-free node harness (host/pcop_node, C++ on top of the C ABI) against the oracle: BASELINE config 0
plumbing (params.yaml -> PointCloud2 -> voxel_grid / centroids / euc_clusters / cluster indices)."""
import os
import subprocess

import numpy as np
import pytest

from pointcloud_obstacle_processing_amd import pc2io, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "pointcloud_obstacle_processing_amd", "host")


def read_indices(path):
    lines = open(path).read().split("\n")
    k = int(lines[0])
    return [np.array(lines[1 + c].split()[1:], dtype=np.int32) for c in range(k)]


@pytest.mark.parametrize("mode", ["frame", "stage"])
def test_node_harness_c1(oracle, tmp_path, mode):
    exe = os.path.join(HOST, "pcop_node")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", HOST])
    pts = synth.make_cloud("C1")
    cloud = str(tmp_path / "c1.pc2")
    pc2io.write_pointcloud2(cloud, pts, point_step=32, offsets=(0, 4, 8), width=100, height=100)
    args = [exe, "--params", os.path.join(ROOT, "tests", "golden", "params_c1.yaml"), "--cloud", cloud, "--out", str(tmp_path)]
    if mode == "stage":
        args.append("--stage-api")
    out = subprocess.run(args, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    p = synth.params_of("C1")
    ref = oracle.process_xyz(pts, **p)
    vox = pc2io.read_pointcloud2(str(tmp_path / "voxel_grid.pc2"))
    assert vox["frame_id"] == "world" and vox["point_step"] == 16 and vox["height"] == 1       # od.cpp:290-293
    assert vox["xyz"].tobytes() == ref["voxel_xyz"].tobytes()
    clusters = read_indices(str(tmp_path / "cluster_indices.txt"))
    assert len(clusters) == len(ref["offsets"]) - 1
    for c, idx in enumerate(clusters):
        assert np.array_equal(idx, ref["indices"][ref["offsets"][c]:ref["offsets"][c + 1]])
    if mode == "frame":
        cen = np.loadtxt(str(tmp_path / "centroids.txt"), dtype=np.float32).reshape(-1, 4)
        want = oracle.cluster_summaries(ref["voxel_xyz"], ref["offsets"], ref["indices"])
        assert cen.shape == want.shape and np.max(np.abs(cen - want)) <= 1e-5
        ec = pc2io.read_pointcloud2(str(tmp_path / "euc_clusters.pc2"))
        assert ec["frame_id"] == "/world"                                                      # od.cpp:865
        assert ec["xyz"].tobytes() == ref["voxel_xyz"][ref["indices"]].tobytes()


def test_node_harness_occupancy_then_chain(oracle, tmp_path):
    """--occupancy: build_initial_occupancy_grid_dataset (host mirror, reference argument list) in front of the
    stage API, as cloud_cb runs it (od.cpp:725-797); limits, block_size and dev_percent come from the yaml tree."""
    exe = os.path.join(HOST, "pcop_node")
    subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
    pts = synth.make_cloud("C1")
    pts[::37, 1] = np.nan          # NaN y passes the reference's predicate (only x is tested, od.cpp:196)
    pts[::91, 0] = np.nan
    cloud = str(tmp_path / "c1.pc2")
    pc2io.write_pointcloud2(cloud, pts, point_step=32, offsets=(0, 4, 8), width=100, height=100)
    yaml_path = str(tmp_path / "params.yaml")
    with open(yaml_path, "w") as f:
        f.write("pointcloud_obstacle_processing:\n        x_min: 0.4\n        x_max: 4.0\n        y_min: 0.3\n        y_max: 3.2\n"
                "        z_min: -0.4\n        z_max: 0.2\n        obstacle_detection:\n                block_size: 0.2\n"
                "                dev_percent: 0.4\n                downsample_size: 0.05\n                euc_cluster_tolerance: 0.1\n"
                "                euc_min_cluster_size: 5\n                euc_max_cluster_size: 20000\n")
    out = subprocess.run([exe, "--params", yaml_path, "--cloud", cloud, "--out", str(tmp_path), "--stage-api", "--occupancy"],
                         capture_output=True, text=True, timeout=300)
    # NaN y survives the filter and the voxel grid is then handed a cloud it believes dense: the library refuses it loudly
    assert out.returncode != 0 and "non-finite" in out.stderr
    pts = synth.make_cloud("C1")
    pts[::91, 0] = np.nan
    pc2io.write_pointcloud2(cloud, pts, point_step=32, offsets=(0, 4, 8), width=100, height=100)
    out = subprocess.run([exe, "--params", yaml_path, "--cloud", cloud, "--out", str(tmp_path), "--stage-api", "--occupancy"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    p = oracle.occ_params(np.float32(0.4), np.float32(4.0), np.float32(0.3), np.float32(3.2), np.float32(-0.4), np.float32(0.2),
                          np.float32(0.2), np.float32(0.4))
    ref = oracle.occupancy(pts, p)
    lines = open(str(tmp_path / "occupancy.txt")).read().split("\n")
    gw, gh, kept = (int(v) for v in lines[0].split())
    assert (gw, gh, kept) == (p.grid_width, p.grid_height, ref["n_out"])
    assert np.array_equal(np.array(lines[1].split(), dtype=np.int64), ref["counts"])
    assert np.array_equal(np.array(lines[2].split(), dtype=np.int64), ref["row_avg"])
    assert np.array_equal(np.array(lines[3].split(), dtype=np.int64), ref["grid"].astype(np.int64))
    want = oracle.process_xyz(ref["xyz"], leaf=0.05, tol=0.1, min_cluster=5, max_cluster=20000)
    vox = pc2io.read_pointcloud2(str(tmp_path / "voxel_grid.pc2"))
    assert vox["xyz"].tobytes() == want["voxel_xyz"].tobytes()
    clusters = read_indices(str(tmp_path / "cluster_indices.txt"))
    assert len(clusters) == len(want["offsets"]) - 1
    for c, idx in enumerate(clusters):
        assert np.array_equal(idx, want["indices"][want["offsets"][c]:want["offsets"][c + 1]])


def test_node_harness_sor_between_voxel_grid_and_clusters(oracle, tmp_path):
    """--sor: remove_statistical_outliers (host mirror) between downsample_cloud and extract_euclidian_clusters, as
    cloud_cb runs them (od.cpp:740-797); meanK and the stddev multiplier come from the yaml tree."""
    exe = os.path.join(HOST, "pcop_node")
    subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
    pts = synth.make_cloud("C1")
    cloud = str(tmp_path / "c1.pc2")
    pc2io.write_pointcloud2(cloud, pts, point_step=32, offsets=(0, 4, 8), width=100, height=100)
    yaml_path = str(tmp_path / "params.yaml")
    with open(yaml_path, "w") as f:
        f.write("pointcloud_obstacle_processing:\n        obstacle_detection:\n                downsample_size: 0.05\n"
                "                statistical_outlier_meanK: 15\n                statistical_outlier_stdDevThres: 0.8\n"
                "                euc_cluster_tolerance: 0.1\n                euc_min_cluster_size: 5\n                euc_max_cluster_size: 20000\n")
    out = subprocess.run([exe, "--params", yaml_path, "--cloud", cloud, "--out", str(tmp_path), "--stage-api", "--sor"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    vox, _ = oracle.voxel_downsample(pts, 0.05)
    ref = oracle.sor(vox, 15, np.float32(0.8))
    got = pc2io.read_pointcloud2(str(tmp_path / "sor.pc2"))
    assert got["frame_id"] == "world" and got["xyz"].tobytes() == ref["xyz"].tobytes()
    assert 0 < len(ref["xyz"]) < len(vox)
    want_off, want_idx = oracle.euclidean_clusters(ref["xyz"], 0.1, 5, 20000)
    clusters = read_indices(str(tmp_path / "cluster_indices.txt"))
    assert len(clusters) == len(want_off) - 1
    for c, idx in enumerate(clusters):
        assert np.array_equal(idx, want_idx[want_off[c]:want_off[c + 1]])


def test_node_harness_node_order(oracle, tmp_path):
    """--node-order: the node's own order (od.cpp:727-797) through one pcop_process_node_clouds call: bounds filter + occupancy
    binning -> voxel grid -> statistical outlier removal -> plane removal -> clusters, parameters from the yaml; the clouds of
    the topics voxel_grid, statistical_outliers, planar_cloud, indices_cloud and cloud_f come out of the same call."""
    exe = os.path.join(HOST, "pcop_node")
    subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
    pts = synth.make_cloud("C3", n=90_000, seed=31)
    pts[::91] = np.nan                      # a depth camera's holes
    cloud = str(tmp_path / "c3.pc2")
    pc2io.write_pointcloud2(cloud, pts, point_step=32, offsets=(0, 4, 8), width=300, height=300)
    yaml_path = str(tmp_path / "params.yaml")
    with open(yaml_path, "w") as f:
        f.write("pointcloud_obstacle_processing:\n        x_min: 0.2\n        x_max: 4.3\n        y_min: 0.2\n        y_max: 4.3\n"
                "        z_min: -0.4\n        z_max: 0.7\n        obstacle_detection:\n                block_size: 0.2\n"
                "                dev_percent: 0.4\n                downsample_size: 0.04\n                statistical_outlier_meanK: 12\n"
                "                statistical_outlier_stdDevThres: 1.5\n                plane_segment_dist_thres: 0.04\n"
                "                plane_segment_angle: 20\n                euc_cluster_tolerance: 0.1\n"
                "                euc_min_cluster_size: 5\n                euc_max_cluster_size: 20000\n")
    out = subprocess.run([exe, "--params", yaml_path, "--cloud", cloud, "--out", str(tmp_path), "--node-order"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    f32 = np.float32
    p = oracle.occ_params(f32(0.2), f32(4.3), f32(0.2), f32(4.3), f32(-0.4), f32(0.7), f32(0.2), f32(0.4))
    kept = oracle.occupancy(pts, p)
    vox, _ = oracle.voxel_downsample(kept["xyz"], 0.04)
    s = oracle.sor(vox, 12, 1.5)
    pl = oracle.segment_planes(s["xyz"], 0.04)
    assert pl["n_planes"] >= 1
    s = dict(xyz=pl["xyz"], n_inl=len(s["xyz"]))
    off, idx = oracle.euclidean_clusters(s["xyz"], 0.1, 5, 20000)
    assert "inside the box %d, voxels %d, inliers %d, planes %d, clustered %d" % (kept["n_out"], len(vox), s["n_inl"], pl["n_planes"], len(pl["xyz"])) in out.stdout
    # the five clouds the node publishes on its way (od.cpp:1004-1008), all from the one fused call
    def cloud_of(name):
        m = pc2io.read_pointcloud2(str(tmp_path / (name + ".pc2")))
        assert m["frame_id"] == "world" and m["point_step"] == 16 and m["height"] == 1
        return m["xyz"]
    sor_kept = oracle.sor(vox, 12, 1.5)["xyz"]
    assert cloud_of("voxel_grid").tobytes() == vox.tobytes()                    # before the outlier removal (od.cpp:288-295)
    assert cloud_of("statistical_outliers").tobytes() == sor_kept.tobytes()     # od.cpp:331-338
    assert cloud_of("planar_cloud").tobytes() == pl["xyz"].tobytes()            # what the planes left (od.cpp:401-408)
    assert cloud_of("indices_cloud").tobytes() == pl["last_plane"].tobytes()    # the last plane removed (od.cpp:410-417)
    # cloud_f: the cloud the last plane was cut from = what is left and the last plane, in their original order
    # (planar_cloud.swap(cloud_f), od.cpp:389-392): rebuilt here by walking the outlier removal's output
    rest, last = pl["xyz"], pl["last_plane"]
    i = j = 0
    want = []
    for q in sor_kept:
        if i < len(rest) and q.tobytes() == rest[i].tobytes():
            want.append(q); i += 1
        elif j < len(last) and q.tobytes() == last[j].tobytes():
            want.append(q); j += 1
    assert i == len(rest) and j == len(last)
    assert cloud_of("cloud_f").tobytes() == np.array(want, dtype=np.float32).reshape(-1, 3).tobytes()
    got = pc2io.read_pointcloud2(str(tmp_path / "clustered.pc2"))              # the cloud the cluster indices refer to
    assert got["xyz"].tobytes() == s["xyz"].tobytes()
    clusters = read_indices(str(tmp_path / "cluster_indices.txt"))
    assert len(clusters) == len(off) - 1 and len(clusters) > 0
    for c, ci in enumerate(clusters):
        assert np.array_equal(ci, idx[off[c]:off[c + 1]])
    ec = pc2io.read_pointcloud2(str(tmp_path / "euc_clusters.pc2"))
    assert ec["xyz"].tobytes() == s["xyz"][idx].tobytes()


@pytest.mark.parametrize("gpus,same", [(1, False), (3, True)])
def test_node_harness_pool(oracle, tmp_path, gpus, same):
    """pcop_node --gpus G: the C++ pool of host/pcop_pool.hpp (a feeder thread and a context per GPU, frame f on GPU f mod G,
    the 32-byte parameter struct broadcast from rank 0 - by RCCL when the workers sit on different devices or there is one;
    --same-gpu rehearses G = 3 on this box's one device).  Every frame's voxel cloud and cluster indices are the oracle's."""
    exe = os.path.join(HOST, "pcop_node")
    subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
    pts = synth.make_cloud("C1")
    cloud = str(tmp_path / "c1.pc2")
    pc2io.write_pointcloud2(cloud, pts, point_step=16, offsets=(0, 4, 8), width=len(pts), height=1)
    frames = 7
    args = [exe, "--params", os.path.join(ROOT, "tests", "golden", "params_c1.yaml"), "--cloud", cloud, "--out", str(tmp_path),
            "--gpus", str(gpus), "--frames", str(frames), "--batch", "2"] + (["--same-gpu"] if same else [])
    out = subprocess.run(args, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert ("ncclBroadcast" in out.stdout) == (not same), out.stdout
    p = synth.params_of("C1")
    cut = len(pts) // (4 * frames)
    for i in range(frames):
        ref = oracle.process_xyz(pts[: len(pts) - i * cut], **p)
        vox = np.fromfile(str(tmp_path / ("pool_frame_%d.vox" % i)), dtype=np.float32).reshape(-1, 3)
        assert vox.tobytes() == ref["voxel_xyz"].tobytes(), i
        clusters = read_indices(str(tmp_path / ("pool_frame_%d.txt" % i)))
        assert len(clusters) == len(ref["offsets"]) - 1
        for c, idx in enumerate(clusters):
            assert np.array_equal(idx, ref["indices"][ref["offsets"][c]:ref["offsets"][c + 1]])


def test_binding_a_thread_next_to_the_gpu():
    """pcop_bind_thread_near_device (what host/pcop_pool.cpp's workers and bench.py's ranks call before they allocate pinned memory): a
    thread of its own is bound to a subset of the CPUs it could run on before - the GPU's local_cpulist - or left alone (0: no NUMA
    information, or a cpuset that leaves fewer than four of them); the calling process keeps its own mask; a device that does not
    exist is refused."""
    import threading
    from pointcloud_obstacle_processing_amd import capi
    lib = capi.load_library()
    before = os.sched_getaffinity(0)
    seen = {}

    def worker():
        seen["rc"] = lib.pcop_bind_thread_near_device(0)
        seen["mask"] = os.sched_getaffinity(0)   # (tid 0 = the calling thread)

    t = threading.Thread(target=worker)
    t.start()
    t.join()
    assert seen["rc"] >= 0
    if seen["rc"] > 0:
        assert len(seen["mask"]) == seen["rc"] and seen["mask"] <= before and seen["rc"] >= 4
    else:
        assert seen["mask"] == before
    assert os.sched_getaffinity(0) == before
    assert lib.pcop_bind_thread_near_device(4096) == -3


def test_a_refused_call_leaves_no_hip_error_behind_for_the_next_launch(oracle):
    """The HIP runtime keeps a failed call as the thread's last error, and the library checks hipGetLastError() after its launches: a
    refused pcop_bind_thread_near_device / pcop_create (a device that does not exist) must not turn the next frame of a healthy context
    into PCOP_ERR_HIP."""
    import ctypes as C
    from pointcloud_obstacle_processing_amd import capi
    lib = capi.load_library()
    pts = synth.make_cloud("C1")
    p = synth.params_of("C1")
    ref = oracle.process_xyz(pts, **p)
    c = capi.Context(len(pts))
    c.set_params(publish_clouds=1, **p)
    c.process_xyz(pts)
    assert lib.pcop_bind_thread_near_device(4096) == -3
    h = C.c_void_p()
    assert lib.pcop_create(C.byref(h), 4096, 1000) == -3
    got = c.process_xyz(pts)
    assert np.array_equal(got["offsets"], ref["offsets"]) and np.array_equal(got["indices"], ref["indices"])
    c.close()
