"""Parity of the HIP path (through the C ABI, libpcop.so) against the CPU oracle.

Integer/index results must be bit-exact; voxel centroids are compared bit-exactly too
(north star allows 1e-4; the canonical summation order makes them equal by construction).
"""
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


@pytest.fixture(scope="module", params=["stream+grid", "sort+hash"])
def ctx(capi, request):
    """Every test that takes `ctx` runs on both sets of paths: the defaults (streaming voxel path, dense cell
    grid) and their fallbacks (PCOP_VOXEL_PATH=sort, PCOP_CELL_PATH=hash; read when the context is created)."""
    import os
    vox, cell = request.param.split("+")
    old = {k: os.environ.get(k) for k in ("PCOP_VOXEL_PATH", "PCOP_CELL_PATH")}
    os.environ["PCOP_VOXEL_PATH"] = vox
    os.environ["PCOP_CELL_PATH"] = cell
    try:
        c = capi.Context(1_100_000)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    c.voxel_path = vox
    c.cell_path = cell
    yield c
    c.close()


def assert_same_cloud(got, ref):
    assert got.shape == ref.shape, (got.shape, ref.shape)
    if got.tobytes() != ref.tobytes():
        bad = np.flatnonzero((got.view(np.uint32) != ref.view(np.uint32)).any(axis=1))
        raise AssertionError("centroids differ at %d rows, first %s: %s vs %s" %
                             (len(bad), bad[:5], got[bad[0]], ref[bad[0]]))
    # the looser bound the north star states, written out
    assert np.max(np.abs(got - ref), initial=0.0) <= 1e-4


def assert_same_clusters(got_off, got_idx, ref_off, ref_idx):
    assert np.array_equal(got_off, ref_off), (np.diff(got_off)[:10], np.diff(ref_off)[:10])
    assert np.array_equal(got_idx, ref_idx)


# ---------------------------------------------------------------------------------------------
# known-answer tests of SURVEY.md §8c, through the GPU
# ---------------------------------------------------------------------------------------------
def test_kat1_strict_radius(ctx):
    pts = np.array([[0, 0, 0], [0.5, 0, 0]], dtype=f32)
    ctx.set_params(leaf=0.05, tol=0.5, min_cluster=1, max_cluster=10)
    off, idx = ctx.euclidean_clusters(pts)
    assert list(np.diff(off)) == [1, 1]
    ctx.set_params(leaf=0.05, tol=np.nextafter(f32(0.5), f32(1)), min_cluster=1, max_cluster=10)
    off, idx = ctx.euclidean_clusters(pts)
    assert list(np.diff(off)) == [2] and list(idx) == [0, 1]


def test_kat2_kat3_order_and_negative(ctx, oracle):
    ctx.set_params(leaf=0.05, tol=0.1)
    pts = np.array([[0.01, 0.01, 0.11], [0.01, 0.06, 0.01], [0.06, 0.01, 0.01], [0.01, 0.01, 0.01]], dtype=f32)
    vox, pt = ctx.voxel_downsample(pts)
    assert not pt and np.array_equal(vox, pts[[3, 2, 1, 0]])
    st = ctx.debug_state()
    assert list(st.div_b) == [2, 2, 3] and list(st.mul) == [1, 2, 4]
    pts = np.array([[-0.01, 0, 0], [0.01, 0, 0]], dtype=f32)
    vox, _ = ctx.voxel_downsample(pts)
    assert vox.shape[0] == 2 and ctx.debug_state().min_b[0] == -1


def test_kat4_centroid_order(ctx):
    ctx.set_params(leaf=0.05, tol=0.1)
    pts = np.array([[0.01, 0, 0], [0.02, 0, 0], [0.04, 0, 0]], dtype=f32)
    vox, _ = ctx.voxel_downsample(pts)
    expect = ((f32(0) + f32(0.01)) + f32(0.02) + f32(0.04)) / f32(3.0)
    assert vox.shape == (1, 3) and vox[0, 0].tobytes() == f32(expect).tobytes()
    ctx.set_params(leaf=1e9, tol=0.1)
    pts2 = np.array([[16777216.0, 0, 0], [1.0, 0, 0], [1.0, 0, 0]], dtype=f32)
    vox2, _ = ctx.voxel_downsample(pts2)
    assert vox2[0, 0] == f32(16777216.0) / f32(3.0)


def test_kat5_overflow_passthrough(ctx, capi):
    ctx.set_params(leaf=0.001, tol=0.1, min_cluster=1, max_cluster=10)
    pts = np.array([[0, 0, 0], [3000, 0, 0], [0, 3000, 0], [0, 0, 3000]], dtype=f32)
    vox, pt = ctx.voxel_downsample(pts)
    assert pt and np.array_equal(vox, pts)
    ctx.set_params(leaf=0.001, tol=10.0, min_cluster=1, max_cluster=10)
    res = ctx.process_xyz(pts)                       # chain: clustering then runs on the input cloud
    assert res["passthrough"] and res["num_voxels"] == 4
    assert list(np.diff(res["offsets"])) == [1, 1, 1, 1]
    assert np.array_equal(res["voxel_xyz"], pts)
    pts1 = np.array([[0, 0, 0], [3000, 0, 0]], dtype=f32)
    vox1, pt1 = ctx.voxel_downsample(pts1)
    assert not pt1 and vox1.shape[0] == 2
    ctx.set_params(leaf=0.001, tol=0.1, min_cluster=1, max_cluster=10)
    res = ctx.process_xyz(pts)                       # 30000 cells apart: still singletons
    assert list(np.diff(res["offsets"])) == [1, 1, 1, 1]
    # documented limit of the neighbour grid: |coordinate| / (tol/1.99) beyond 2^20 cells is a loud error
    far = np.array([[0, 0, 0], [1.0e6, 0, 0]], dtype=f32)
    with pytest.raises(capi.PcopError) as e:
        ctx.euclidean_clusters(far)
    assert e.value.status == -7
    res = ctx.euclidean_clusters(pts)                # and the context is usable again afterwards
    assert list(np.diff(res[0])) == [1, 1, 1, 1]


def test_kat6_kat7_filter_and_order(ctx, oracle):
    tol = f32(0.1)
    chain = np.zeros((7, 3), dtype=f32)
    chain[:, 0] = np.arange(7, dtype=f32) * f32(0.09)
    ctx.set_params(leaf=0.05, tol=tol, min_cluster=5, max_cluster=6)
    off, idx = ctx.euclidean_clusters(chain)
    assert len(off) - 1 == 0
    ctx.set_params(leaf=0.05, tol=tol, min_cluster=5, max_cluster=7)
    off, idx = ctx.euclidean_clusters(chain)
    assert list(np.diff(off)) == [7] and list(idx) == list(range(7))
    pts = []
    for c, s in enumerate([5, 9, 5, 7]):
        for j in range(s):
            pts.append([j * 0.05, c * 1.0, 0.0])
    pts = np.array(pts, dtype=f32)
    ctx.set_params(leaf=0.05, tol=tol, min_cluster=5, max_cluster=100)
    assert_same_clusters(*ctx.euclidean_clusters(pts), *oracle.euclidean_clusters(pts, tol, 5, 100))


def test_kat8_r2_bits(ctx):
    ctx.set_params(leaf=0.05, tol=0.1, min_cluster=1, max_cluster=10)
    ctx.euclidean_clusters(np.zeros((3, 3), dtype=f32))
    assert np.float32(ctx.debug_state().r2).view(np.uint32) == np.uint32(0x3C23D70B)


def test_empty_single_and_duplicates(ctx, oracle):
    ctx.set_params(leaf=0.05, tol=0.1, min_cluster=1, max_cluster=10)
    vox, pt = ctx.voxel_downsample(np.zeros((0, 3), dtype=f32))
    assert vox.shape == (0, 3) and not pt
    off, idx = ctx.euclidean_clusters(np.zeros((0, 3), dtype=f32))
    assert list(off) == [0]
    off, idx = ctx.euclidean_clusters(np.array([[1, 2, 3]], dtype=f32))
    assert list(off) == [0, 1] and list(idx) == [0]
    res = ctx.process_xyz(np.zeros((0, 3), dtype=f32))
    assert res["num_voxels"] == 0 and list(res["offsets"]) == [0]
    dup = np.array([[0, 0, 0], [0, 0, 0], [0, 0, 0], [5, 5, 5]], dtype=f32)
    assert_same_clusters(*ctx.euclidean_clusters(dup), *oracle.euclidean_clusters(dup, 0.1, 1, 10))


def test_zero_tolerance_gives_singletons(ctx, oracle):
    pts = synth.uniform_field(50, 11)
    ctx.set_params(leaf=0.05, tol=0.0, min_cluster=1, max_cluster=10)
    assert_same_clusters(*ctx.euclidean_clusters(pts), *oracle.euclidean_clusters(pts, 0.0, 1, 10))


def test_nan_handling(ctx, oracle, capi):
    pts = synth.uniform_field(5000, 5)
    pts[::7, 0] = np.nan
    pts[3::11, 2] = np.inf
    ctx.set_params(leaf=0.05, tol=0.1, min_cluster=5, max_cluster=20000)
    vox, _ = ctx.voxel_downsample(pts, is_dense=False)
    ref, _ = oracle.voxel_downsample(pts, 0.05, is_dense=False)
    assert_same_cloud(vox, ref)
    with pytest.raises(capi.PcopError) as e:        # dense cloud with NaN: undefined in PCL, loud error here
        ctx.voxel_downsample(pts, is_dense=True)
    assert e.value.status == -6
    with pytest.raises(capi.PcopError):
        ctx.euclidean_clusters(pts)


def test_bad_params_fail_loudly(ctx, capi):
    ctx.set_params(leaf=0.0, tol=0.1)
    with pytest.raises(capi.PcopError) as e:
        ctx.voxel_downsample(np.zeros((3, 3), dtype=f32))
    assert e.value.status == -2
    ctx.set_params(leaf=0.05, tol=np.inf)
    with pytest.raises(capi.PcopError):
        ctx.euclidean_clusters(np.zeros((3, 3), dtype=f32))
    small = capi.Context(10)
    small.set_params(leaf=0.05, tol=0.1)
    with pytest.raises(capi.PcopError) as e:
        small.voxel_downsample(np.zeros((11, 3), dtype=f32))
    assert e.value.status == -5
    small.close()
    with pytest.raises(capi.PcopError) as e:        # no CPU fallback: a missing device is an error
        capi.Context(10, device=-1)
    assert e.value.status == -3


# ---------------------------------------------------------------------------------------------
# stage parity on seeded clouds
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("layout", [0, 1, 2])
@pytest.mark.parametrize("seed,n,leaf", [(1, 10_000, 0.05), (2, 50_000, 0.2), (3, 70_001, 0.013), (4, 4097, 1.0)])
def test_voxel_stage_matches_oracle(ctx, oracle, layout, seed, n, leaf):
    pts = synth.uniform_field(n, seed)
    ctx.set_params(leaf=leaf, tol=0.1)
    got, pt = ctx.voxel_downsample(pts, layout=layout)
    ref, pt_ref = oracle.voxel_downsample(pts, leaf)
    assert pt == pt_ref
    assert_same_cloud(got, ref)


def test_voxel_stage_sorted_records_are_stable(ctx, oracle):
    """Stage-wise: after the partition passes the 12-byte records are ordered by (voxel key, point index)."""
    if ctx.voxel_path != "sort":
        pytest.skip("records of the sort path")
    pts = synth.uniform_field(30_000, 9)
    ctx.set_params(leaf=0.1, tol=0.1)
    ctx.voxel_downsample(pts)
    st = ctx.debug_state()
    assert st.vpath == 0
    npass = st.plan_voxel.npass
    rec = ctx.debug_buffer("rec_a" if npass % 2 == 1 else "rec_b", np.float32, 3 * len(pts)).reshape(-1, 3)
    keys, grid, _ = oracle.voxel_keys(pts, 0.1)
    assert list(grid[:3]) == list(st.min_b) and list(grid[3:6]) == list(st.div_b)
    order = np.lexsort((np.arange(len(pts)), keys.astype(np.int64)))      # by key, ties by point index
    assert rec.tobytes() == pts[order].tobytes()


def test_voxel_stage_stream_buckets_and_records(ctx, oracle):
    """Stage-wise, streaming path: the buckets are runs of consecutive coarse key bins within the two limits
    (points, cells), they tile the cloud, and the partitioned 12-byte records are ordered by (bucket, point index)."""
    if ctx.voxel_path != "stream":
        pytest.skip("buffers of the streaming path")
    pts = synth.uniform_field(30_000, 9)
    ctx.set_params(leaf=0.1, tol=0.1)
    ctx.voxel_downsample(pts)
    st = ctx.debug_state()
    assert st.vpath == 1 and st.m > 0
    keys, grid, _ = oracle.voxel_keys(pts, 0.1)
    assert list(grid[:3]) == list(st.min_b) and list(grid[3:6]) == list(st.div_b)
    assert st.m == len(np.unique(keys))
    nb = st.nbuckets
    info = ctx.debug_buffer("binfo", np.uint32, 4 * nb).reshape(-1, 4).astype(np.int64)   # first record, records, first key, keys
    assert 1 <= nb <= st.nb_cap
    assert info[0, 0] == 0 and np.array_equal(info[1:, 0], np.cumsum(info[:-1, 1])) and info[:, 1].sum() == len(pts)
    assert (info[:, 1] >= 1).all() and (info[:, 1] <= 2048).all() and (info[:, 3] <= (1 << 17)).all()
    assert (info[:, 2] % (1 << st.cshift) == 0).all() and (info[1:, 2] >= info[:-1, 2] + info[:-1, 3]).all()
    k64 = keys.astype(np.int64)
    bucket = np.searchsorted(info[:, 2], k64, side="right") - 1
    assert (k64 < info[bucket, 2] + info[bucket, 3]).all()
    b2b = ctx.debug_buffer("bin2bucket", np.uint16, 1 << st.gbits)
    assert np.array_equal(b2b[k64 >> st.cshift], bucket.astype(np.uint16))
    rec = ctx.debug_buffer("rec_a", np.float32, 3 * len(pts)).reshape(-1, 3)
    order = np.lexsort((np.arange(len(pts)), bucket))                   # by bucket, ties by point index
    assert rec.tobytes() == pts[order].tobytes()


@pytest.mark.parametrize("cfg,n,mn,mx", [("C1", 10_000, 5, 20000), ("C1", 10_000, 1, 10**9), ("C2", 30_000, 5, 10**9),
                                         ("C3", 120_000, 5, 20000), ("C3L10", 150_000, 2, 20000)])
def test_cluster_stage_matches_oracle(ctx, oracle, cfg, n, mn, mx):
    p = synth.params_of(cfg)
    pts = synth.make_cloud(cfg, n=n)
    vox, _ = oracle.voxel_downsample(pts, p["leaf"])
    ctx.set_params(leaf=p["leaf"], tol=p["tol"], min_cluster=mn, max_cluster=mx)
    for layout in (2, 0):
        assert_same_clusters(*ctx.euclidean_clusters(vox, layout=layout),
                             *oracle.euclidean_clusters(vox, p["tol"], mn, mx))


def test_cluster_stage_large_tolerance_ratio(ctx, oracle):
    """The reference's deployed parameters (params.yaml:16,26): tol/leaf = 26.7, hundreds of points per cell."""
    pts = synth.make_cloud("C3", n=200_000)
    vox, _ = oracle.voxel_downsample(pts, 0.015)
    ctx.set_params(leaf=0.015, tol=0.4, min_cluster=5, max_cluster=20000)
    assert_same_clusters(*ctx.euclidean_clusters(vox), *oracle.euclidean_clusters(vox, 0.4, 5, 20000))
    ctx.set_params(leaf=0.015, tol=0.4, min_cluster=5, max_cluster=10**9)
    assert_same_clusters(*ctx.euclidean_clusters(vox), *oracle.euclidean_clusters(vox, 0.4, 5, 10**9))


def test_cluster_stage_sparse_points(ctx, oracle):
    rng = np.random.default_rng(3)
    pts = rng.uniform(-50, 50, (20_000, 3)).astype(f32)
    ctx.set_params(leaf=0.05, tol=2.0, min_cluster=2, max_cluster=10**9)
    assert_same_clusters(*ctx.euclidean_clusters(pts), *oracle.euclidean_clusters(pts, 2.0, 2, 10**9))


# ---------------------------------------------------------------------------------------------
# whole chain at the BASELINE configs
# ---------------------------------------------------------------------------------------------
def run_chain(ctx, oracle, pts, p):
    ctx.set_params(publish_clouds=1, **p)
    got = ctx.process_xyz(pts)
    ref = oracle.process_xyz(pts, **p)
    assert got["passthrough"] == ref["passthrough"]
    assert_same_cloud(got["voxel_xyz"], ref["voxel_xyz"])
    assert_same_clusters(got["offsets"], got["indices"], ref["offsets"], ref["indices"])
    return got, ref


def test_chain_c1(ctx, oracle):
    got, _ = run_chain(ctx, oracle, synth.make_cloud("C1"), synth.params_of("C1"))
    assert 100 < len(got["offsets"]) - 1 < 300


def test_chain_c2(ctx, oracle):
    p = synth.params_of("C2")
    pts = synth.make_cloud("C2")
    got, _ = run_chain(ctx, oracle, pts, p)
    assert len(got["offsets"]) - 1 == 0                    # one giant component, dropped by max=20000
    p["max_cluster"] = 10**9
    got, _ = run_chain(ctx, oracle, pts, p)
    assert len(got["offsets"]) - 1 >= 1


def test_chain_c3_full_size(ctx, oracle):
    """BASELINE config 2 at its full size: 1M points, 20 box clusters, ground dropped by max=20000."""
    got, _ = run_chain(ctx, oracle, synth.make_cloud("C3"), synth.params_of("C3"))
    assert len(got["offsets"]) - 1 == 20


def test_chain_c3_l10(ctx, oracle):
    run_chain(ctx, oracle, synth.make_cloud("C3L10", n=400_000), synth.params_of("C3L10"))


def test_chain_device_resident_and_repeatable(ctx, oracle):
    """The benchmark path: cloud resident in HBM as SoA planes; repeated frames give identical output."""
    p = synth.params_of("C3")
    pts = synth.make_cloud("C3", n=300_000, seed=31)
    ref = oracle.process_xyz(pts, **p)
    ctx.set_params(publish_clouds=1, **p)
    ptr, n = ctx.upload_soa(pts)
    for _ in range(3):
        got = ctx.process_device(ptr, n)
        assert_same_cloud(got["voxel_xyz"], ref["voxel_xyz"])
        assert_same_clusters(got["offsets"], got["indices"], ref["offsets"], ref["indices"])
    ctx.dev_free(ptr)


def test_frames_in_flight_on_several_contexts(capi, oracle):
    p = synth.params_of("C3")
    clouds = [synth.make_cloud("C3", n=100_000, seed=100 + i) for i in range(4)]
    refs = [oracle.process_xyz(c, **p) for c in clouds]
    ctxs = [capi.Context(100_000) for _ in clouds]
    ptrs = []
    for c, pts in zip(ctxs, clouds):
        c.set_params(publish_clouds=0, **p)
        ptrs.append(c.upload_soa(pts))
    for rep in range(2):
        for c, (ptr, n) in zip(ctxs, ptrs):
            c.submit_device(ptr, n)
        for c, ref in zip(ctxs, refs):
            got = c.wait()
            assert got["voxel_xyz"] is None
            assert_same_clusters(got["offsets"], got["indices"], ref["offsets"], ref["indices"])
    for c in ctxs:
        c.close()


def test_batched_frames_match_oracle(capi, oracle):
    """pcop_submit_batch: frames of different sizes and shapes in one set of launches."""
    p = synth.params_of("C3")
    clouds = [synth.make_cloud("C3", n=120_000, seed=200), synth.make_cloud("C3", n=60_000, seed=201),
              synth.uniform_field(30_000, 7), synth.make_cloud("C3", n=200_000, seed=203), np.zeros((0, 3), dtype=f32)]
    refs = [oracle.process_xyz(c, **p) for c in clouds]
    c = capi.Context(200_000, max_frames=6)
    c.set_params(publish_clouds=1, **p)
    for rep in range(2):
        c.submit_batch_host(clouds)
        got = c.wait_batch()
        for g, r in zip(got, refs):
            if len(r["voxel_xyz"]):
                assert_same_cloud(g["voxel_xyz"], r["voxel_xyz"])
            assert g["num_voxels"] == len(r["voxel_xyz"])
            assert_same_clusters(g["offsets"], g["indices"], r["offsets"], r["indices"])
    k = len(got[3]["offsets"]) - 1
    xyzr = c.cluster_summaries_frame(3, k)
    ref = oracle.cluster_summaries(got[3]["voxel_xyz"], got[3]["offsets"], got[3]["indices"])
    assert np.max(np.abs(xyzr - ref)) <= 1e-5
    with pytest.raises(capi.PcopError):              # more frames than the context holds
        c.submit_batch_host(clouds + clouds)
    c.close()


def test_pointcloud2_entry(ctx, oracle):
    """sensor_msgs/PointCloud2 with extra fields (Kinect: xyz + pad + rgb, point_step 32) and row padding."""
    p = synth.params_of("C1")
    pts = synth.make_cloud("C1", n=6000)
    pts[5] = np.nan
    width, height, step = 100, 60, 32
    row_step = width * step + 64
    buf = np.zeros(height * row_step, dtype=np.uint8)
    rows = buf.reshape(height, row_step)[:, :width * step].reshape(height, width, step)
    rows[:, :, 4:8] = pts[:, 1].reshape(height, width).copy().view(np.uint8).reshape(height, width, 4)    # y @4
    rows[:, :, 8:12] = pts[:, 0].reshape(height, width).copy().view(np.uint8).reshape(height, width, 4)   # x @8
    rows[:, :, 16:20] = pts[:, 2].reshape(height, width).copy().view(np.uint8).reshape(height, width, 4)  # z @16
    ctx.set_params(publish_clouds=1, **p)
    got = ctx.process_pointcloud2(buf, width, height, step, row_step, 8, 4, 16, is_bigendian=False, is_dense=False)
    ref = oracle.process_xyz(pts, is_dense=False, **p)
    assert_same_cloud(got["voxel_xyz"], ref["voxel_xyz"])
    assert_same_clusters(got["offsets"], got["indices"], ref["offsets"], ref["indices"])
    # big-endian payload
    be = buf.copy()
    r2 = be.reshape(height, row_step)[:, :width * step].reshape(height, width, step)
    for o in (4, 8, 16):
        r2[:, :, o:o + 4] = r2[:, :, o:o + 4][:, :, ::-1]
    got2 = ctx.process_pointcloud2(be, width, height, step, row_step, 8, 4, 16, is_bigendian=True, is_dense=False)
    assert_same_clusters(got2["offsets"], got2["indices"], ref["offsets"], ref["indices"])
    # unaligned field offsets (the message format allows them), the smallest point_step there is
    odd = np.zeros(6000 * 13, dtype=np.uint8).reshape(6000, 13)
    for a, o in ((0, 1), (1, 5), (2, 9)):
        odd[:, o:o + 4] = pts[:, a].copy().view(np.uint8).reshape(6000, 4)
    got3 = ctx.process_pointcloud2(odd.reshape(-1), 6000, 1, 13, 6000 * 13, 1, 5, 9, is_dense=False)
    assert_same_clusters(got3["offsets"], got3["indices"], ref["offsets"], ref["indices"])
    # fields that do not fit the point, a row shorter than its points: refused, offsets near 2^32 included (no wrap past the test)
    from pointcloud_obstacle_processing_amd import capi as m
    for args in ((buf, width, height, step, row_step, 29, 4, 16), (buf, width, height, step, row_step, 8, 0xFFFFFFFE, 16),
                 (buf, width, height, step, row_step, 8, 4, 0xFFFFFFFC), (buf, width, height, step, width * step - 1, 8, 4, 16),
                 (buf, width, height, 3, row_step, 0, 0, 0)):
        with pytest.raises(m.PcopError):
            ctx.process_pointcloud2(*args, is_dense=False)
    assert_same_clusters(*[ctx.process_pointcloud2(buf, width, height, step, row_step, 8, 4, 16, is_dense=False)[k] for k in ("offsets", "indices")],
                         ref["offsets"], ref["indices"])   # (and the context is as usable as before)


def test_cluster_summaries(ctx, oracle):
    p = synth.params_of("C3")
    pts = synth.make_cloud("C3", n=200_000)
    ctx.set_params(publish_clouds=1, **p)
    got = ctx.process_xyz(pts)
    k = len(got["offsets"]) - 1
    assert k > 0
    xyzr = ctx.cluster_summaries(k)
    ref = oracle.cluster_summaries(got["voxel_xyz"], got["offsets"], got["indices"])
    assert np.max(np.abs(xyzr - ref)) <= 1e-5              # PointWithRad floats (centre: double mean; r: float chain)


def test_radix_replanning_and_digit_widths(capi, oracle, monkeypatch):
    """Pass counts adapt after the first frame; every digit width gives the same answer."""
    p = synth.params_of("C3")
    pts = synth.make_cloud("C3", n=150_000, seed=77)
    ref = oracle.process_xyz(pts, **p)
    monkeypatch.setenv("PCOP_VOXEL_PATH", "sort")   # the pass plan under test belongs to the sort path
    for bits in ("5", "8", "11"):
        monkeypatch.setenv("PCOP_RADIX_BITS", bits)
        c = capi.Context(len(pts))
        c.set_params(publish_clouds=1, **p)
        for rep in range(3):
            got = c.process_xyz(pts)
            assert_same_cloud(got["voxel_xyz"], ref["voxel_xyz"])
            assert_same_clusters(got["offsets"], got["indices"], ref["offsets"], ref["indices"])
        # a different cloud on the same context, with a much larger key space, forces a re-plan
        big = synth.uniform_field(100_000, 5) * f32(20.0)
        c.set_params(publish_clouds=1, leaf=0.02, tol=0.1, min_cluster=2, max_cluster=20000)
        got = c.process_xyz(big)
        refb = oracle.process_xyz(big, leaf=0.02, tol=0.1, min_cluster=2, max_cluster=20000)
        assert_same_cloud(got["voxel_xyz"], refb["voxel_xyz"])
        assert_same_clusters(got["offsets"], got["indices"], refb["offsets"], refb["indices"])
        c.close()


# ---------------------------------------------------------------------------------------------
# streaming voxel path: its own re-planning (buckets provided for, coarse bins) and its fallback
# ---------------------------------------------------------------------------------------------
def test_stream_path_grows_and_falls_back(capi, oracle, monkeypatch):
    monkeypatch.setenv("PCOP_VOXEL_PATH", "stream")
    c = capi.Context(1_000_000)
    # (1) small cloud: streaming path, smallest bucket provision
    p = synth.params_of("C3")
    pts = synth.make_cloud("C3", n=150_000, seed=5)
    c.set_params(publish_clouds=1, **p)
    got = c.process_xyz(pts)
    ref = oracle.process_xyz(pts, **p)
    assert_same_cloud(got["voxel_xyz"], ref["voxel_xyz"])
    assert_same_clusters(got["offsets"], got["indices"], ref["offsets"], ref["indices"])
    st = c.debug_state()
    assert st.vpath == 1 and st.nb_cap == 1024 and st.nbuckets <= 1024
    # (2) points scattered over a wide, sparse grid need more than 1024 buckets (a bucket spans at most 2^17 cells):
    #     one re-run with more of them, then it stays there
    big = synth.uniform_field(300_000, 11) * f32(6.0)
    pl = dict(leaf=0.02, tol=0.05, min_cluster=2, max_cluster=20000)
    c.set_params(publish_clouds=1, **pl)
    refb = oracle.process_xyz(big, **pl)
    for rep in range(2):
        got = c.process_xyz(big)
        assert_same_cloud(got["voxel_xyz"], refb["voxel_xyz"])
        assert_same_clusters(got["offsets"], got["indices"], refb["offsets"], refb["indices"])
    st = c.debug_state()
    assert st.vpath == 1 and st.nbuckets > 1024 and st.nb_cap >= st.nbuckets
    # (3) the small cloud again on the grown context
    c.set_params(publish_clouds=1, **p)
    got = c.process_xyz(pts)
    assert_same_cloud(got["voxel_xyz"], ref["voxel_xyz"])
    assert_same_clusters(got["offsets"], got["indices"], ref["offsets"], ref["indices"])
    # (4) crowded voxels: more points in one voxel than a bucket holds, whatever the bins: sort path (and it stays
    #     there for this leaf size)
    dense = synth.uniform_field(300_000, 13) * f32(0.05)
    pd = dict(leaf=0.05, tol=0.1, min_cluster=1, max_cluster=20000)
    c.set_params(publish_clouds=1, **pd)
    refd = oracle.process_xyz(dense, **pd)
    for rep in range(2):
        got = c.process_xyz(dense)
        assert_same_cloud(got["voxel_xyz"], refd["voxel_xyz"])
        assert_same_clusters(got["offsets"], got["indices"], refd["offsets"], refd["indices"])
    assert c.debug_state().vpath == 0
    # (5) a new leaf size puts the context back on the streaming path
    c.set_params(publish_clouds=1, **p)
    got = c.process_xyz(pts)
    assert_same_cloud(got["voxel_xyz"], ref["voxel_xyz"])
    assert c.debug_state().vpath == 1
    c.close()


@pytest.mark.parametrize("bits,n", [(11, 100_000), (12, 250_000), (13, 250_000), (14, 250_000)])
def test_stream_path_any_bin_count_same_result(capi, oracle, monkeypatch, bits, n):
    """The coarse bin count is a tuning knob: every value gives the same bytes."""
    monkeypatch.setenv("PCOP_VOXEL_PATH", "stream")
    monkeypatch.setenv("PCOP_BIN_BITS", str(bits))
    c = capi.Context(300_000)
    p = synth.params_of("C3")
    pts = synth.make_cloud("C3", n=n, seed=6)
    c.set_params(publish_clouds=1, **p)
    got = c.process_xyz(pts)
    ref = oracle.process_xyz(pts, **p)
    st = c.debug_state()
    assert st.vpath == 1 and st.gbits == bits
    assert_same_cloud(got["voxel_xyz"], ref["voxel_xyz"])
    assert_same_clusters(got["offsets"], got["indices"], ref["offsets"], ref["indices"])
    c.close()


def test_stream_path_heavy_bins_are_split(capi, oracle, monkeypatch):
    """A dense sheet inside a tall, nearly empty box: the sheet's points fall into a few rows of the key space, 5000 to a
    coarse bin even at the finest bins, where a bucket holds 2048.  Such bins are cut into buckets along the keys
    (k_heavy_split); the frame stays on the streaming path and the bytes are the oracle's."""
    monkeypatch.setenv("PCOP_VOXEL_PATH", "stream")
    rng = np.random.default_rng(31)
    sheet = np.concatenate([rng.uniform(0, 2, (400_000, 2)), rng.normal(0.5, 0.001, (400_000, 1))], axis=1)
    air = rng.uniform(0, 2, (2_000, 3))
    pts = np.concatenate([sheet, air]).astype(f32)[rng.permutation(402_000)]
    p = dict(leaf=0.01, tol=0.05, min_cluster=5, max_cluster=1_000_000)
    c = capi.Context(len(pts))
    c.set_params(publish_clouds=1, **p)
    ref = oracle.process_xyz(pts, **p)
    for rep in range(3):
        got = c.process_xyz(pts)
        assert_same_cloud(got["voxel_xyz"], ref["voxel_xyz"])
        assert_same_clusters(got["offsets"], got["indices"], ref["offsets"], ref["indices"])
    st = c.debug_state()
    assert st.vpath == 1 and st.maxbin > 2048
    # an ordinary cloud on the same context afterwards
    p3 = synth.params_of("C3")
    small = synth.make_cloud("C3", n=150_000, seed=5)
    c.set_params(publish_clouds=1, **p3)
    got = c.process_xyz(small)
    ref3 = oracle.process_xyz(small, **p3)
    assert_same_cloud(got["voxel_xyz"], ref3["voxel_xyz"])
    assert_same_clusters(got["offsets"], got["indices"], ref3["offsets"], ref3["indices"])
    c.close()


def test_stream_path_without_lane_ordered_atomics(capi, oracle, monkeypatch):
    """PCOP_LDS_ORDER=0: the instances of k_bin_scatter / k_bucket_reduce that build their stable ranks from peer masks
    instead of the return values of LDS atomic adds (what a device failing the probe at context creation would run)."""
    monkeypatch.setenv("PCOP_VOXEL_PATH", "stream")
    monkeypatch.setenv("PCOP_LDS_ORDER", "0")
    c = capi.Context(400_000, max_frames=2)
    for cfg, n, seed in (("C3", 300_000, 8), ("C2", 100_000, 9), ("C1", 10_000, 10)):
        p = synth.params_of(cfg)
        pts = synth.make_cloud(cfg, n=n, seed=seed)
        c.set_params(publish_clouds=1, **p)
        ref = oracle.process_xyz(pts, **p)
        got = c.process_xyz(pts)
        assert c.debug_state().vpath == 1
        assert_same_cloud(got["voxel_xyz"], ref["voxel_xyz"])
        assert_same_clusters(got["offsets"], got["indices"], ref["offsets"], ref["indices"])
    # two frames in one batch
    p = synth.params_of("C3")
    c.set_params(publish_clouds=1, **p)
    frames = [synth.make_cloud("C3", n=200_000, seed=11), synth.make_cloud("C3", n=120_000, seed=12)]
    c.submit_batch_host(frames)
    for got, pts in zip(c.wait_batch(), frames):
        ref = oracle.process_xyz(pts, **p)
        assert_same_cloud(got["voxel_xyz"], ref["voxel_xyz"])
        assert_same_clusters(got["offsets"], got["indices"], ref["offsets"], ref["indices"])
    c.close()


def test_stream_path_huge_sparse_grid(capi, oracle, monkeypatch):
    """A voxel grid of more than 2^26 cells (tens of metres at leaf 0.02): still the streaming path, same answer."""
    monkeypatch.setenv("PCOP_VOXEL_PATH", "stream")
    c = capi.Context(200_000)
    pts = synth.uniform_field(100_000, 21) * f32(10.0)
    c.set_params(publish_clouds=1, leaf=0.02, tol=0.1, min_cluster=2, max_cluster=20000)
    ref = oracle.process_xyz(pts, leaf=0.02, tol=0.1, min_cluster=2, max_cluster=20000)
    for rep in range(2):
        got = c.process_xyz(pts)
        st = c.debug_state()
        assert int(st.div_b[0]) * int(st.div_b[1]) * int(st.div_b[2]) > (1 << 26)
        assert_same_cloud(got["voxel_xyz"], ref["voxel_xyz"])
        assert_same_clusters(got["offsets"], got["indices"], ref["offsets"], ref["indices"])
    c.close()


def test_stream_path_nonfinite_points_dropped(capi, oracle, monkeypatch):
    monkeypatch.setenv("PCOP_VOXEL_PATH", "stream")
    c = capi.Context(100_000)
    pts = synth.make_cloud("C3", n=60_000, seed=8).copy()
    pts[::97, 1] = np.nan
    pts[5::211, 0] = np.inf
    p = synth.params_of("C3")
    c.set_params(publish_clouds=1, **p)
    got = c.process_xyz(pts, is_dense=False)
    ref = oracle.process_xyz(pts, is_dense=False, **p)
    assert c.debug_state().vpath == 1
    assert_same_cloud(got["voxel_xyz"], ref["voxel_xyz"])
    assert_same_clusters(got["offsets"], got["indices"], ref["offsets"], ref["indices"])
    c.close()


def test_egress_in_toROSMsg_layout(capi, oracle):
    """od.cpp:288-295, 861-866: the voxel_grid and euc_clusters clouds as pcl::toROSMsg lays them out, 16-byte
    {x, y, z, 1.0f} records, written by the GPU (no host re-pack)."""
    p = synth.params_of("C3")
    pts = synth.make_cloud("C3", n=200_000, seed=12)
    ref = oracle.process_xyz(pts, **p)
    c = capi.Context(len(pts))
    c.set_params(publish_clouds=1, **p)
    c.set_cloud_layout(capi.LAYOUT_AOS16)
    got = c.process_xyz(pts)
    want = np.ones((len(ref["voxel_xyz"]), 4), dtype=np.float32)
    want[:, :3] = ref["voxel_xyz"]
    assert got["voxel_xyz"].tobytes() == want.tobytes()
    assert_same_clusters(got["offsets"], got["indices"], ref["offsets"], ref["indices"])
    clu = c.cluster_cloud()
    assert clu.tobytes() == want[ref["indices"]].tobytes() and len(clu) == len(ref["indices"])
    c.set_cloud_layout(capi.LAYOUT_AOS12)                      # and back
    got = c.process_xyz(pts)
    assert_same_cloud(got["voxel_xyz"], ref["voxel_xyz"])
    # a frame without kept clusters gives an empty cloud
    c.set_params(publish_clouds=1, leaf=p["leaf"], tol=p["tol"], min_cluster=10**6, max_cluster=10**7)
    c.process_xyz(pts)
    assert len(c.cluster_cloud()) == 0
    with pytest.raises(capi.PcopError):
        c.set_cloud_layout(capi.LAYOUT_SOA)
    c.close()


def test_error_frame_leaves_no_marks_behind(capi, oracle):
    """A frame given up with an error (NaN in a cloud declared dense) on a cell grid larger than the occupancy scan was
    launched for must not leave occupancy flags set: the next frame on the same context equals the oracle."""
    c = capi.Context(300_000)
    small = synth.make_cloud("C1")
    c.set_params(publish_clouds=1, **synth.params_of("C1"))
    for rep in range(3):
        c.process_xyz(small)                       # the scan grid shrinks to what this small cloud needs
    wide = synth.uniform_field(250_000, 31) * f32(4.0)
    bad = wide.copy()
    bad[123_456, 2] = np.nan
    p = dict(leaf=0.05, tol=0.05, min_cluster=2, max_cluster=20000)
    c.set_params(publish_clouds=1, **p)
    with pytest.raises(capi.PcopError) as e:
        c.euclidean_clusters(bad)                  # cell marking sees the NaN while the scan grid is too small
    assert e.value.status == -6
    ref = oracle.process_xyz(wide, **p)
    got = c.process_xyz(wide)
    assert_same_cloud(got["voxel_xyz"], ref["voxel_xyz"])
    assert_same_clusters(got["offsets"], got["indices"], ref["offsets"], ref["indices"])
    assert not c.debug_buffer("cflag", np.uint8, 1 << 24).any()
    c.close()


def _poke_flags(capi, c, name, positions):
    """Fault injection: occupancy flag bytes set behind the library's back (what a given-up frame could leave behind)."""
    import ctypes as C
    p, b = C.c_void_p(), C.c_size_t()
    c._chk(c.lib.pcop_debug_buffer(c.h, name.encode(), C.byref(p), C.byref(b)))
    one = np.ones(1, dtype=np.uint8)
    for pos in positions:
        assert pos < b.value
        c._chk(c.lib.pcop_dev_upload(c.h, C.c_void_p(p.value + int(pos)), one.ctypes.data, 1))


def test_stale_occupancy_flags_cost_a_rerun_not_a_fault(capi, oracle):
    """The dense cell grid relies on 16 MiB of flag bytes being all zero between frames.  A byte that is not (poked here) used to
    give a cell without points - cell ids shifted, per-cell arrays indexed past what the frame owns.  Now the frame that meets it
    raises kErrStale before anything is indexed by the ids, the host scrubs the flags and runs the batch again: oracle-identical
    results, one counted re-run, no fault."""
    p = synth.params_of("C1")
    pts = synth.make_cloud("C1")
    ref = oracle.process_xyz(pts, **p)
    # (a) one flag inside the grid (position 3: the margin of the first row - no point ever lies there), single frames
    c = capi.Context(len(pts))
    c.set_params(publish_clouds=1, **p)
    got = c.process_xyz(pts)
    assert_same_clusters(got["offsets"], got["indices"], ref["offsets"], ref["indices"])
    assert c.debug_counter("stale_reruns") == 0
    _poke_flags(capi, c, "cflag", [3])
    got = c.process_xyz(pts)
    assert_same_cloud(got["voxel_xyz"], ref["voxel_xyz"])
    assert_same_clusters(got["offsets"], got["indices"], ref["offsets"], ref["indices"])
    assert c.debug_counter("stale_reruns") == 1
    got = c.process_xyz(pts)
    assert_same_clusters(got["offsets"], got["indices"], ref["offsets"], ref["indices"])
    assert c.debug_counter("stale_reruns") == 1
    assert not c.debug_buffer("cflag", np.uint8, 1 << 24).any()
    c.close()
    # (b) more stale flags than the frame has points (cell ids would leave the per-cell arrays), in slot 1 of a batch of three
    tiny = synth.make_cloud("C1", n=300, seed=9)
    rt = oracle.process_xyz(tiny, **p)
    c = capi.Context(1024, max_frames=3)
    c.set_params(publish_clouds=1, **p)
    c.submit_batch_host([tiny, tiny, tiny])
    c.wait_batch()
    _poke_flags(capi, c, "cflag@1", list(range(3, 9000, 2)))
    for rep in range(2):
        c.submit_batch_host([tiny, tiny, tiny])
        for g in c.wait_batch():
            assert_same_cloud(g["voxel_xyz"], rt["voxel_xyz"])
            assert_same_clusters(g["offsets"], g["indices"], rt["offsets"], rt["indices"])
    assert c.debug_counter("stale_reruns") == 1
    for k in range(3):
        assert not c.debug_buffer("cflag@%d" % k, np.uint8, 1 << 24).any()
    c.close()


# ---------------------------------------------------------------------------------------------
# C5 at full size (BASELINE.json configs[4]): 5,000,000 points, leaf 0.01 m, tolerance 0.05 m
# ---------------------------------------------------------------------------------------------
_c5_cache = {}


def _c5(oracle, seed):
    if seed not in _c5_cache:
        pts = synth.make_cloud("C5", seed=seed)
        _c5_cache[seed] = (pts, oracle.process_xyz(pts, **synth.params_of("C5")))
    return _c5_cache[seed]


@pytest.mark.parametrize("paths", ["stream+grid", "sort+hash"])
def test_c5_full_size_chain(capi, oracle, monkeypatch, paths):
    vox, cell = paths.split("+")
    monkeypatch.setenv("PCOP_VOXEL_PATH", vox)
    monkeypatch.setenv("PCOP_CELL_PATH", cell)
    pts, ref = _c5(oracle, 5)
    assert len(pts) == 5_000_000 and len(ref["voxel_xyz"]) > 4_000_000
    c = capi.Context(len(pts))
    c.set_params(publish_clouds=1, **synth.params_of("C5"))
    for rep in range(2):   # the first call grows the bucket provision, the second runs the settled plan
        got = c.process_xyz(pts)
        assert_same_cloud(got["voxel_xyz"], ref["voxel_xyz"])
        assert_same_clusters(got["offsets"], got["indices"], ref["offsets"], ref["indices"])
    st = c.debug_state()
    # C5's ground plane puts about 30 000 points into one coarse key bin (2 x 10^8 cells / 2^14 bins) where a bucket holds
    # 2048.  Bins up to four buckets' worth are cut into buckets along the keys (k_heavy_split); fifteen times over, with
    # nearly every point a voxel of its own, the sort path is the faster one (measured: 1.77 ms against 2.29 ms a frame),
    # so the frame is re-run there
    assert st.vpath == 0 and st.passthrough == 0
    c.close()


def test_c5_batch_of_two(capi, oracle):
    """Two different 5M-point frames in the same launches."""
    c = capi.Context(5_000_000, max_frames=2)
    c.set_params(publish_clouds=1, **synth.params_of("C5"))
    frames = [_c5(oracle, 5), _c5(oracle, 6)]
    c.submit_batch_host([f[0] for f in frames])
    res = c.wait_batch()
    for got, (_, ref) in zip(res, frames):
        assert_same_cloud(got["voxel_xyz"], ref["voxel_xyz"])
        assert_same_clusters(got["offsets"], got["indices"], ref["offsets"], ref["indices"])
    c.close()
    _c5_cache.clear()


# ---------------------------------------------------------------------------------------------
# dense cell grid: stage-wise check and its fallback to the hash table
# ---------------------------------------------------------------------------------------------
def test_cell_grid_ids_and_metadata(capi, oracle, monkeypatch):
    """Cell ids are the ranks of the occupied cells in grid order; per-cell metadata holds count and priority."""
    monkeypatch.setenv("PCOP_CELL_PATH", "grid")
    c = capi.Context(100_000)
    pts = synth.make_cloud("C3", n=50_000, seed=12)
    tol = 0.1
    c.set_params(leaf=0.02, tol=tol, min_cluster=1, max_cluster=10**9)
    c.euclidean_clusters(pts)
    st = c.debug_state()
    assert st.cpath == 1 and st.n_cells > 0
    ig = 1.99 / abs(np.float32(tol).astype(np.float64))
    cell = np.floor(pts.astype(np.float64) * ig).astype(np.int64)
    g = cell - np.array(list(st.c0), dtype=np.int64)
    assert (g >= 3).all() and (g[:, 0] <= st.c_dim[0] - 4).all() and (g[:, 1] <= st.c_dim[1] - 4).all() and (g[:, 2] <= st.c_dim[2] - 4).all()
    pos = g[:, 2] * st.c_plane + g[:, 1] * st.c_rowbits + g[:, 0]
    upos, ids_ref, cnt_ref = np.unique(pos, return_inverse=True, return_counts=True)
    assert st.n_cells == len(upos)
    ids = c.debug_buffer("pt_slot", np.uint32, len(pts))
    assert np.array_equal(ids, ids_ref.astype(np.uint32))
    meta = c.debug_buffer("slot_meta", np.uint32, 4 * st.n_cells).reshape(-1, 4)
    assert np.array_equal(meta[:, 1], cnt_ref.astype(np.uint32))                 # points per cell
    assert np.array_equal(meta[:, 3], upos.astype(np.uint32))                    # grid position
    first = np.full(len(upos), len(pts), dtype=np.int64)
    np.minimum.at(first, ids_ref, np.arange(len(pts)))
    assert np.array_equal(meta[:, 2], np.arange(len(upos), dtype=np.uint32))     # priority = the cell's id (ids follow the grid)
    assert np.array_equal(meta[:, 0], np.concatenate(([0], np.cumsum(cnt_ref)[:-1])).astype(np.uint32))
    assert not c.debug_buffer("cflag", np.uint8, 1 << 24).any()
    c.close()


def test_cell_grid_too_large_falls_back_to_hash(capi, oracle, monkeypatch):
    monkeypatch.setenv("PCOP_CELL_PATH", "grid")
    c = capi.Context(100_000)
    rng = np.random.default_rng(5)
    # a few compact blobs spread over hundreds of metres: > 2^24 cells at tolerance 0.1
    centres = rng.uniform(-300.0, 300.0, size=(40, 3)).astype(f32)
    pts = (centres[rng.integers(0, 40, 60_000)] + rng.normal(0.0, 0.08, size=(60_000, 3))).astype(f32)
    p = dict(leaf=0.02, tol=0.1, min_cluster=3, max_cluster=10**9)
    c.set_params(publish_clouds=1, **p)
    ref = oracle.process_xyz(pts, **p)
    for rep in range(2):
        got = c.process_xyz(pts)
        assert_same_cloud(got["voxel_xyz"], ref["voxel_xyz"])
        assert_same_clusters(got["offsets"], got["indices"], ref["offsets"], ref["indices"])
    assert c.debug_state().cpath == 0
    # a compact cloud with another tolerance: back on the grid, and the borrowed hash arrays were left clean
    q = synth.params_of("C3")
    small = synth.make_cloud("C3", n=80_000, seed=4)
    q2 = dict(q, tol=0.11)
    c.set_params(publish_clouds=1, **q2)
    got = c.process_xyz(small)
    ref2 = oracle.process_xyz(small, **q2)
    assert c.debug_state().cpath == 1
    assert_same_cloud(got["voxel_xyz"], ref2["voxel_xyz"])
    assert_same_clusters(got["offsets"], got["indices"], ref2["offsets"], ref2["indices"])
    # and the hash path once more on the same context (PCOP_ERR-free, same answers as the oracle)
    c.set_params(publish_clouds=1, **p)
    got = c.process_xyz(pts)
    assert_same_clusters(got["offsets"], got["indices"], ref["offsets"], ref["indices"])
    c.close()


def test_crowded_cells_switch_to_the_hash_kernels(capi, oracle, monkeypatch):
    """Many points per cell (large tolerance): after one batch the context runs the wave-per-cell kernels of the
    hash path; a sparse cloud brings it back to the grid.  Same answers either way."""
    monkeypatch.setenv("PCOP_CELL_PATH", "grid")
    c = capi.Context(200_000)
    pts = synth.make_cloud("C3", n=150_000, seed=31)
    p = dict(leaf=0.015, tol=0.4, min_cluster=5, max_cluster=10**9)     # the deployed params.yaml values
    c.set_params(publish_clouds=1, **p)
    ref = oracle.process_xyz(pts, **p)
    paths = []
    for rep in range(3):
        got = c.process_xyz(pts)
        assert_same_cloud(got["voxel_xyz"], ref["voxel_xyz"])
        assert_same_clusters(got["offsets"], got["indices"], ref["offsets"], ref["indices"])
        paths.append(c.debug_state().cpath)
    assert paths[0] == 1 and paths[-1] == 0
    q = synth.params_of("C3")
    c.set_params(publish_clouds=1, **q)
    ref2 = oracle.process_xyz(pts, **q)
    for rep in range(2):
        got = c.process_xyz(pts)
        assert_same_clusters(got["offsets"], got["indices"], ref2["offsets"], ref2["indices"])
    assert c.debug_state().cpath == 1
    c.close()


@pytest.mark.parametrize("vox,cell", [("stream", "hash"), ("sort", "grid")])
def test_mixed_path_combinations(capi, oracle, monkeypatch, vox, cell):
    """The `ctx` fixture covers stream+grid and sort+hash; the two paths are independent, so the other two
    combinations must give the same answers too (also without the hipGraph, launches issued eagerly)."""
    monkeypatch.setenv("PCOP_VOXEL_PATH", vox)
    monkeypatch.setenv("PCOP_CELL_PATH", cell)
    for no_graph in ("0", "1"):
        monkeypatch.setenv("PCOP_NO_GRAPH", no_graph)
        c = capi.Context(200_000, max_frames=2)
        for cfg, n in (("C1", None), ("C3", 150_000)):
            p = synth.params_of(cfg)
            pts = synth.make_cloud(cfg, n=n, seed=17)
            c.set_params(publish_clouds=1, **p)
            ref = oracle.process_xyz(pts, **p)
            got = c.process_xyz(pts)
            st = c.debug_state()
            assert st.vpath == (1 if vox == "stream" else 0) and st.cpath == (1 if cell == "grid" else 0)
            assert_same_cloud(got["voxel_xyz"], ref["voxel_xyz"])
            assert_same_clusters(got["offsets"], got["indices"], ref["offsets"], ref["indices"])
        c.close()


def test_hash_table_sub_table_overflow(capi, oracle, monkeypatch):
    """Hash fallback, a sparse axis-aligned plane: every cell has the same z index inside its 4x4x4 block, so only 16 of
    the table's 64 sub-tables are addressed and they overflow; probing must spill into the next sub-table (it used to
    walk its own one forever).  Also a line (4 sub-tables) and a single column of cells."""
    monkeypatch.setenv("PCOP_CELL_PATH", "hash")
    c = capi.Context(20_000)
    rng = np.random.default_rng(13)
    plane = np.zeros((4000, 3), dtype=f32)
    plane[:, :2] = rng.uniform(-20, 20, (4000, 2))
    line = np.zeros((3000, 3), dtype=f32)
    line[:, 0] = rng.uniform(-30, 30, 3000)
    for pts, tol in ((plane, 0.02), (line, 0.004), (plane * f32(0.01), 0.0002)):
        c.set_params(leaf=0.001, tol=tol, min_cluster=1, max_cluster=10**9)
        got = c.euclidean_clusters(pts)
        assert c.debug_state().cpath == 0
        assert_same_clusters(*got, *oracle.euclidean_clusters(pts, tol, 1, 10**9))
    c.close()
