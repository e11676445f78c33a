"""The node's own order on the device (od.cpp:727-797): build_initial_occupancy_grid_dataset -> downsample_cloud ->
remove_statistical_outliers -> kd-tree + extract_euclidian_clusters, one C-ABI call (pcop_process_node), the cloud of
every step staying in HBM.  Checked against the stage-by-stage chain of the CPU restatements: kept points, voxel
centroids, inliers, cluster offsets and indices all bit-exact."""
import numpy as np
import pytest

from pointcloud_obstacle_processing_amd import synth

pytestmark = pytest.mark.gpu
f32 = np.float32


@pytest.fixture(scope="module")
def capi():
    from pointcloud_obstacle_processing_amd import capi as m
    m.load_library()
    return m


def field_cloud(n, seed):
    """C3-like scene (ground plane + hovering boxes on a 4.5 m field) with a margin of points outside the node's box and
    a few NaNs, as a depth camera delivers them."""
    pts = synth.make_cloud("C3", n=n, seed=seed).copy()
    rng = np.random.default_rng(seed + 1000)
    k = n // 20
    pts[rng.choice(n, k, replace=False)] += rng.uniform(-1.5, 1.5, (k, 3)).astype(f32)      # strays, some outside the box
    pts[rng.choice(n, n // 500, replace=False)] = np.nan      # a depth camera's holes: x, y and z all NaN (the reference's
    return pts                                                 # filter tests x only, od.cpp:199)


def oracle_chain(oracle, pts, occ, leaf, mean_k, mul, tol, mn, mx, plane_thr=None):
    cloud = pts
    ref = {}
    if occ is not None:
        r = oracle.occupancy(pts, occ)
        cloud = r["xyz"]
        ref.update(n_filtered=r["n_out"], counts=r["counts"], row_avg=r["row_avg"], grid=r["grid"], index_errors=r["index_errors"])
    vox, pt = oracle.voxel_downsample(cloud, leaf)
    ref["n_voxels"] = len(vox)
    if mean_k:
        s = oracle.sor(vox, mean_k, mul)
        ref["threshold"] = s["threshold"]
        vox = s["xyz"]
    ref["n_inliers"] = len(vox)
    if plane_thr is not None:
        pl = oracle.segment_planes(vox, plane_thr)
        ref["n_planes"], ref["coeffs"] = pl["n_planes"], pl["coeffs"]
        vox = pl["xyz"]
    ref["cloud"] = vox
    ref["offsets"], ref["indices"] = oracle.euclidean_clusters(vox, tol, mn, mx)
    return ref


def occ_of(capi, oracle):
    # the yaml's field box (params.yaml:2-7) with a usable z range
    p = oracle.occ_params(0.0, 4.5, 0.0, 3.78, -0.5, 0.6, 0.15, 0.5)
    return p, capi.OccParams(p.x_min, p.x_max, p.y_min, p.y_max, p.z_min, p.z_max, p.block_size, p.dev_percent, p.grid_width, p.grid_height)


@pytest.mark.parametrize("use_occ,mean_k", [(True, 15), (True, 0), (False, 15), (False, 0)])
@pytest.mark.parametrize("paths", ["stream+grid", "sort+hash"])
def test_node_chain_matches_stage_by_stage_oracle(capi, oracle, monkeypatch, use_occ, mean_k, paths):
    vox_path, cell_path = paths.split("+")
    if cell_path == "hash" and mean_k:
        pytest.skip("the k-NN pass needs the dense cell grid")
    monkeypatch.setenv("PCOP_VOXEL_PATH", vox_path)
    monkeypatch.setenv("PCOP_CELL_PATH", cell_path)
    leaf, tol, mn, mx, mul = 0.05, 0.12, 5, 20000, 1.0
    pts = field_cloud(300_000, 21)
    if not use_occ:
        pts = pts[np.isfinite(pts).all(axis=1)]
    po, pc = occ_of(capi, oracle)
    ref = oracle_chain(oracle, pts, po if use_occ else None, leaf, mean_k, mul, tol, mn, mx)
    c = capi.Context(len(pts))
    c.set_params(leaf=leaf, tol=tol, min_cluster=mn, max_cluster=mx, publish_clouds=1)
    for rep in range(2):
        got = c.process_node(pts, occ=pc if use_occ else None, mean_k=mean_k, stddev_mul=mul, want_grid=True)
        assert got["voxel_xyz"].tobytes() == ref["cloud"].tobytes()
        assert np.array_equal(got["offsets"], ref["offsets"]) and np.array_equal(got["indices"], ref["indices"])
        assert got["n_voxels"] == ref["n_voxels"] and got["n_inliers"] == len(ref["cloud"])
        if use_occ:
            assert got["n_filtered"] == ref["n_filtered"] and got["index_errors"] == ref["index_errors"]
            assert np.array_equal(got["counts"], ref["counts"]) and np.array_equal(got["row_avg"], ref["row_avg"])
            assert np.array_equal(got["grid"], ref["grid"])
        if mean_k:
            assert got["threshold"] == ref["threshold"]
    # the clouds of the topics come off the GPU as well
    clu = c.cluster_cloud()
    assert clu[:, :3].tobytes() == ref["cloud"][ref["indices"]].tobytes()
    k = len(ref["offsets"]) - 1
    if k:
        assert np.max(np.abs(c.cluster_summaries(k) - oracle.cluster_summaries(ref["cloud"], ref["offsets"], ref["indices"]))) <= 1e-5
    # the context goes back to plain frames afterwards
    p1 = synth.params_of("C1")
    c.set_params(publish_clouds=1, **p1)
    small = synth.make_cloud("C1")
    r1, g1 = oracle.process_xyz(small, **p1), c.process_xyz(small)
    assert g1["voxel_xyz"].tobytes() == r1["voxel_xyz"].tobytes() and np.array_equal(g1["indices"], r1["indices"])
    c.close()


@pytest.mark.parametrize("use_occ,mean_k", [(True, 15), (False, 0), (False, 15)])
def test_node_chain_with_plane_removal(capi, oracle, use_occ, mean_k):
    """All five steps of od.cpp:727-797: bounds filter -> voxel grid -> outlier removal -> RANSAC plane removal -> clusters."""
    leaf, tol, mn, mx, mul, pthr = 0.04, 0.12, 5, 20000, 1.0, 0.04
    pts = field_cloud(300_000, 23)
    if not use_occ:
        pts = pts[np.isfinite(pts).all(axis=1)]
    po, pc = occ_of(capi, oracle)
    ref = oracle_chain(oracle, pts, po if use_occ else None, leaf, mean_k, mul, tol, mn, mx, plane_thr=pthr)
    assert ref["n_planes"] >= 1 and len(ref["offsets"]) > 1
    c = capi.Context(len(pts))
    c.set_params(leaf=leaf, tol=tol, min_cluster=mn, max_cluster=mx, publish_clouds=1)
    for rep in range(2):
        got = c.process_node(pts, occ=pc if use_occ else None, mean_k=mean_k, stddev_mul=mul, plane=capi.plane_params(pthr))
        assert got["voxel_xyz"].tobytes() == ref["cloud"].tobytes()
        assert np.array_equal(got["offsets"], ref["offsets"]) and np.array_equal(got["indices"], ref["indices"])
        assert got["n_voxels"] == ref["n_voxels"] and got["n_inliers"] == ref["n_inliers"]
        assert got["n_planes"] == ref["n_planes"] and got["n_clustered"] == len(ref["cloud"]) and got["planes_found_all"]
    assert c.cluster_cloud()[:, :3].tobytes() == ref["cloud"][ref["indices"]].tobytes()
    c.close()


def test_node_chain_on_a_cloud_resident_in_hbm(capi, oracle):
    leaf, tol, mn, mx = 0.04, 0.1, 5, 20000
    pts = field_cloud(500_000, 22)
    po, pc = occ_of(capi, oracle)
    ref = oracle_chain(oracle, pts, po, leaf, 15, 2.0, tol, mn, mx)
    c = capi.Context(len(pts))
    c.set_params(leaf=leaf, tol=tol, min_cluster=mn, max_cluster=mx, publish_clouds=1)
    ptr, n = c.upload_soa(pts)
    got = c.process_node(None, occ=pc, mean_k=15, stddev_mul=2.0, dev_ptr=ptr, n=n)
    assert got["voxel_xyz"].tobytes() == ref["cloud"].tobytes()
    assert np.array_equal(got["offsets"], ref["offsets"]) and np.array_equal(got["indices"], ref["indices"])
    c.close()


def test_node_chain_edge_cases(capi, oracle):
    c = capi.Context(50_000)
    c.set_params(leaf=0.05, tol=0.12, min_cluster=5, max_cluster=20000, publish_clouds=1)
    po, pc = occ_of(capi, oracle)
    # nothing inside the box: empty all the way down (sor2.filter on an empty cloud gives an empty cloud)
    far = synth.uniform_field(5000, 3) + f32(100.0)
    got = c.process_node(far, occ=pc, mean_k=15, stddev_mul=1.0)
    assert got["n_filtered"] == 0 and got["n_voxels"] == 0 and got["n_inliers"] == 0 and len(got["offsets"]) == 1
    # fewer voxels than mean_k + 1: refused, as pcop_statistical_outlier_removal refuses it
    few = synth.uniform_field(8, 4)
    with pytest.raises(capi.PcopError) as e:
        c.process_node(few, mean_k=15, stddev_mul=1.0)
    assert e.value.status == -2
    # and the context still works
    pts = field_cloud(40_000, 5)
    ref = oracle_chain(oracle, pts, po, 0.05, 15, 1.0, 0.12, 5, 20000)
    got = c.process_node(pts, occ=pc, mean_k=15, stddev_mul=1.0)
    assert got["voxel_xyz"].tobytes() == ref["cloud"].tobytes() and np.array_equal(got["indices"], ref["indices"])
    c.close()


@pytest.mark.parametrize("mean_k,use_plane", [(0, True), (15, False), (4, True)])
def test_node_chain_on_a_pass_through_frame(capi, oracle, mean_k, use_plane):
    """A cloud whose voxel grid would overflow (blobs 40 m apart at leaf 0.01: VoxelGrid's "output = input" rule) and the
    filter steps after it: the cloud returned is the one the cluster indices refer to - what the last step kept of the
    INPUT - not the pass-through cloud (found by the randomised campaign, tests/tools/fuzz.py)."""
    rng = np.random.default_rng(77)
    centres = rng.uniform(-20, 20, (25, 3))
    pts = (centres[rng.integers(0, 25, 5000)] + rng.normal(0, 0.03, (5000, 3))).astype(f32)
    leaf, tol, mn, mx, thr = 0.01, 0.1, 2, 20000, 0.4
    vox, passthrough = oracle.voxel_downsample(pts, leaf)
    assert passthrough and len(vox) == len(pts)
    cur = vox
    if mean_k:
        cur = oracle.sor(cur, mean_k, 1.0)["xyz"]
    if use_plane:
        cur = oracle.segment_planes(cur, thr)["xyz"]
    assert 0 < len(cur) < len(pts)
    ro, ri = oracle.euclidean_clusters(cur, tol, mn, mx)
    c = capi.Context(len(pts))
    c.set_params(leaf=leaf, tol=tol, min_cluster=mn, max_cluster=mx, publish_clouds=1)
    for rep in range(2):
        got = c.process_node(pts, mean_k=mean_k, stddev_mul=1.0, plane=capi.plane_params(thr) if use_plane else None)
        assert got["passthrough"] == 1
        assert got["voxel_xyz"].tobytes() == cur.tobytes()
        assert np.array_equal(got["offsets"], ro) and np.array_equal(got["indices"], ri)
        assert got["n_clustered"] == len(cur)
    c.close()


def test_node_chain_error_in_the_middle_leaves_a_usable_context(capi, oracle, monkeypatch):
    """PCOP_DEBUG_PLANE_MISMATCH=1 makes the first plane split of the context "disagree" with the host's selection: the call
    returns an error from the middle of the frame - and the same context then runs the node chain and a plain batch correctly
    (nothing left in flight, the occupancy flags clean)."""
    monkeypatch.setenv("PCOP_DEBUG_PLANE_MISMATCH", "1")
    p = synth.params_of("C3")
    pts = synth.make_cloud("C3", n=60_000, seed=77)
    c = capi.Context(60_000, max_frames=2)
    c.set_params(publish_clouds=1, **p)
    with pytest.raises(capi.PcopError):
        c.process_node(pts, mean_k=8, stddev_mul=1.0, plane=capi.plane_params(0.03))
    vox, _ = oracle.voxel_downsample(pts, p["leaf"])
    cur = oracle.sor(vox, 8, 1.0)["xyz"]
    cur = oracle.segment_planes(cur, 0.03)["xyz"]
    ro, ri = oracle.euclidean_clusters(cur, p["tol"], p["min_cluster"], p["max_cluster"])
    g = c.process_node(pts, mean_k=8, stddev_mul=1.0, plane=capi.plane_params(0.03))
    gv = g["voxel_xyz"] if g["voxel_xyz"] is not None else np.zeros((0, 3), dtype=np.float32)
    assert gv.tobytes() == cur.tobytes() and np.array_equal(g["offsets"], ro) and np.array_equal(g["indices"], ri)
    ref = oracle.process_xyz(pts, **p)
    for _ in range(2):
        c.submit_batch_host([pts, pts[::2].copy()])
        got = c.wait_batch()
        assert got[0]["voxel_xyz"].tobytes() == ref["voxel_xyz"].tobytes() and np.array_equal(got[0]["indices"], ref["indices"])
    c.close()


def kinect_message(pts, width, height, step=32, pad=64, bigendian=False):
    """A sensor_msgs/PointCloud2 blob the way a Kinect driver lays it out: xyz + padding + rgb in 32-byte points (here with the fields
    out of order: y @4, x @8, z @16) and padded rows."""
    row_step = width * step + pad
    buf = np.zeros(height * row_step, dtype=np.uint8)
    rows = buf.reshape(height, row_step)[:, :width * step].reshape(height, width, step)
    for off, a in ((4, 1), (8, 0), (16, 2)):
        v = pts[:, a].reshape(height, width).copy()
        b = v.view(np.uint8).reshape(height, width, 4)
        rows[:, :, off:off + 4] = b[:, :, ::-1] if bigendian else b
    return buf, row_step


@pytest.mark.parametrize("bigendian", [False, True])
def test_node_chain_fed_with_the_pointcloud2_blob(capi, oracle, bigendian):
    """pcop_process_node_pointcloud2 (od.cpp:688-689 in front of od.cpp:727-797): the message goes to the GPU as it is; same results
    as pcop_process_node on the cloud a host loop would have unpacked, and as the stage-by-stage oracle chain."""
    leaf, tol, mn, mx, mul, pthr = 0.04, 0.12, 5, 20000, 1.0, 0.04
    width, height = 640, 400
    pts = field_cloud(width * height, 29)
    buf, row_step = kinect_message(pts, width, height, bigendian=bigendian)
    po, pc = occ_of(capi, oracle)
    ref = oracle_chain(oracle, pts, po, leaf, 15, mul, tol, mn, mx, plane_thr=pthr)
    c = capi.Context(len(pts))
    c.set_params(leaf=leaf, tol=tol, min_cluster=mn, max_cluster=mx, publish_clouds=1)
    want = c.process_node(pts, occ=pc, mean_k=15, stddev_mul=mul, plane=capi.plane_params(pthr))
    got = c.process_node_pointcloud2(buf, width, height, 32, row_step, 8, 4, 16, is_bigendian=bigendian, is_dense=False, occ=pc, mean_k=15,
                                     stddev_mul=mul, plane=capi.plane_params(pthr), want_clouds=True)
    for g in (want, got):
        assert g["voxel_xyz"].tobytes() == ref["cloud"].tobytes()
        assert np.array_equal(g["offsets"], ref["offsets"]) and np.array_equal(g["indices"], ref["indices"])
        assert g["n_voxels"] == ref["n_voxels"] and g["n_inliers"] == ref["n_inliers"] and g["n_planes"] == ref["n_planes"]
        assert g["n_filtered"] == ref["n_filtered"]
    assert got["clouds"]["planar_cloud"][:, :3].tobytes() == ref["cloud"].tobytes()
    # a malformed message is refused before a byte of it is read
    for bad in (dict(point_step=32, off_x=29), dict(point_step=32, off_x=0xFFFFFFFE), dict(point_step=3, off_x=0), dict(row_step=width * 32 - 1)):
        kw = dict(point_step=32, row_step=row_step, off_x=8)
        kw.update(bad)
        with pytest.raises(capi.PcopError) as e:
            c.process_node_pointcloud2(buf, width, height, kw["point_step"], kw["row_step"], kw["off_x"], 4, 16, occ=pc, mean_k=15, plane=capi.plane_params(pthr))
        assert e.value.status == -1
    with pytest.raises(capi.PcopError) as e:   # more points than the context holds
        c.process_node_pointcloud2(buf, width, height + 1, 32, row_step, 8, 4, 16)
    assert e.value.status == -5
    again = c.process_node_pointcloud2(buf, width, height, 32, row_step, 8, 4, 16, is_bigendian=bigendian, is_dense=False, occ=pc, mean_k=15,
                                       stddev_mul=mul, plane=capi.plane_params(pthr))
    assert np.array_equal(again["indices"], ref["indices"])   # the context is as usable as before
    c.close()


def test_node_chain_without_a_plane_publishes_the_outlier_cloud_twice(capi, oracle):
    """No plane removed (a cloud the RANSAC finds no inliers in - here: fewer than three points left - or an empty one): the reference still
    publishes indices_cloud and cloud_f, both the copies of the outlier removal's output cloud_cb made (od.cpp:774-775, 410-426)."""
    leaf, tol = 0.05, 0.2
    c = capi.Context(4096)
    c.set_params(leaf=leaf, tol=tol, min_cluster=1, max_cluster=100, publish_clouds=1)
    pts = np.array([[0.01, 0.01, 0.0], [1.0, 1.0, 0.3]], dtype=f32)   # two voxels: the plane model needs three points
    buf = np.zeros(2 * 16, dtype=np.uint8)
    buf.view(f32).reshape(2, 4)[:, :3] = pts
    got = c.process_node_pointcloud2(buf, 2, 1, 16, 32, 0, 4, 8, plane=capi.plane_params(0.04), want_clouds=True)
    assert got["n_planes"] == 0 and got["n_voxels"] == 2
    cl = got["clouds"]
    vox, _ = oracle.voxel_downsample(pts, leaf)
    assert cl["voxel_grid"][:, :3].tobytes() == vox.tobytes()
    assert cl["statistical_outliers"] is None            # that step was not asked for
    for nm in ("planar_cloud", "indices_cloud", "cloud_f"):
        assert cl[nm] is not None and cl[nm][:, :3].tobytes() == vox.tobytes(), nm
        assert np.all(cl[nm][:, 3] == 1.0)
    # an empty message: five empty clouds where the steps ran
    got = c.process_node_pointcloud2(np.zeros(16, dtype=np.uint8), 0, 1, 16, 0, 0, 4, 8, plane=capi.plane_params(0.04), want_clouds=True)
    for nm in ("planar_cloud", "indices_cloud", "cloud_f"):
        assert got["clouds"][nm] is not None and len(got["clouds"][nm]) == 0, nm
    c.close()
