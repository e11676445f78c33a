"""The streaming voxel path with its keys formed against the context's box (no bounds pass: csrc/pcop_voxel_box.hpp,
k_box_setup mode 2 + bin_hist_on_box) against the CPU oracle, through the C ABI.

The first batch of a context (or after a new leaf size) runs k_minmax and leaves a box; from then on k_bin_hist takes the true
bounds on its way and checks every point against the box.  A point outside sends the batch back through the bounds pass.
Every path must leave PCL's own min_b / div_b in DevState and give the same bytes.
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


def same(got, ref):
    assert got["num_voxels"] == len(ref["voxel_xyz"])
    if len(ref["voxel_xyz"]):
        assert got["voxel_xyz"].tobytes() == ref["voxel_xyz"].tobytes(), "centroids differ"
        assert np.max(np.abs(got["voxel_xyz"] - ref["voxel_xyz"]), initial=0.0) <= 1e-4   # the north star's bound, written out
    assert np.array_equal(got["offsets"], ref["offsets"]) and np.array_equal(got["indices"], ref["indices"])


def run(c, clouds, refs, is_dense=True):
    c.submit_batch_host(clouds, is_dense=is_dense)
    got = c.wait_batch()
    for g, r in zip(got, refs):
        same(g, r)
    return c.debug_state()


def grid_of(pts, leaf):
    ok = np.isfinite(pts).all(axis=1)
    inv = f32(1.0) / f32(leaf)
    lo = np.floor(pts[ok].min(axis=0) * inv).astype(np.int64)
    hi = np.floor(pts[ok].max(axis=0) * inv).astype(np.int64)
    return list(lo), list(hi - lo + 1)


def test_box_path_matches_oracle(capi, oracle):
    p = synth.params_of("C3")
    clouds = [synth.make_cloud("C3", n=150_000, seed=410), synth.make_cloud("C3", n=9_000, seed=411),
              synth.uniform_field(40_000, 9), np.zeros((0, 3), dtype=f32), synth.make_cloud("C3", n=150_000, seed=412)]
    refs = [oracle.process_xyz(c, **p) for c in clouds]
    c = capi.Context(150_000, max_frames=5)
    c.set_params(publish_clouds=1, **p)
    seen = []
    for _ in range(3):
        st = run(c, clouds, refs)
        seen.append(int(st.sbox))
        lo, div = grid_of(clouds[0], p["leaf"])
        assert list(st.min_b) == lo and list(st.div_b) == div
        assert st.n_valid == len(clouds[0]) and st.passthrough == 0
    assert seen == [0, 1, 1], seen
    c.close()


def test_box_path_point_outside_and_new_leaf(capi, oracle):
    p = synth.params_of("C3")
    a = [synth.make_cloud("C3", n=80_000, seed=420 + i) for i in range(2)]
    b = [a[0].copy(), a[1].copy()]
    b[0][777] = (-1.5, 9.0, -0.75)
    ra = [oracle.process_xyz(c, **p) for c in a]
    rb = [oracle.process_xyz(c, **p) for c in b]
    c = capi.Context(80_000, max_frames=2)
    c.set_params(publish_clouds=1, **p)
    assert run(c, a, ra).sbox == 0
    assert run(c, a, ra).sbox == 1
    assert run(c, b, rb).sbox == 0      # a point outside: re-run with the bounds pass
    assert run(c, b, rb).sbox == 1      # the box has grown
    q = dict(p, leaf=0.05)
    c.set_params(publish_clouds=1, **q)   # a new leaf size is a new grid: the box starts over
    rq = [oracle.process_xyz(x, **q) for x in a]
    assert run(c, a, rq).sbox == 0
    assert run(c, a, rq).sbox == 1
    c.close()


def test_box_path_nan_holes_and_all_nan_frame(capi, oracle):
    p = synth.params_of("C3")
    a = synth.make_cloud("C3", n=100_000, seed=430)
    a[::53] = np.nan
    b = np.full((5_000, 3), np.nan, dtype=f32)
    cl = [a, b, synth.make_cloud("C3", n=60_000, seed=431)]
    refs = [oracle.process_xyz(x, is_dense=False, **p) for x in cl]
    c = capi.Context(100_000, max_frames=3)
    c.set_params(publish_clouds=1, **p)
    assert run(c, cl, refs, is_dense=False).sbox == 0
    for _ in range(2):
        st = run(c, cl, refs, is_dense=False)
        assert st.sbox == 1 and st.n_valid == int(np.isfinite(a).all(axis=1).sum())
    with pytest.raises(capi.PcopError):   # declared dense: undefined in the reference, an error on every path
        run(c, cl, refs, is_dense=True)
    st = run(c, cl, refs, is_dense=False)   # and the context works on
    assert st.sbox == 1
    c.close()


def test_box_path_heavy_bins(capi, oracle):
    """A dense sheet (a coarse bin above a bucket's capacity: k_heavy_split) with the keys on the box."""
    rng = np.random.default_rng(44)
    n = 200_000
    pts = np.empty((n, 3), dtype=f32)
    pts[:, 0] = rng.uniform(0.0, 1.2, n); pts[:, 1] = rng.uniform(0.0, 1.2, n); pts[:, 2] = rng.normal(0.0, 0.002, n)
    p = dict(leaf=0.02, tol=0.1, min_cluster=5, max_cluster=10**9)
    cl = [pts, pts[::-1].copy()]
    refs = [oracle.process_xyz(x, **p) for x in cl]
    c = capi.Context(n, max_frames=2)
    c.set_params(publish_clouds=1, **p)
    run(c, cl, refs)
    for _ in range(2):
        assert run(c, cl, refs).sbox == 1
    c.close()


def test_box_path_switched_off(capi, oracle, monkeypatch):
    monkeypatch.setenv("PCOP_VOXEL_BOX", "0")
    p = synth.params_of("C3")
    clouds = [synth.make_cloud("C3", n=70_000, seed=460 + i) for i in range(2)]
    refs = [oracle.process_xyz(c, **p) for c in clouds]
    c = capi.Context(70_000, max_frames=2)
    c.set_params(publish_clouds=1, **p)
    for _ in range(3):
        assert run(c, clouds, refs).sbox == 0
    c.close()


def test_bounded_wait_that_runs_out_costs_a_rerun_not_the_frame(capi, oracle, monkeypatch):
    """PCOP_DEBUG_SPIN0=1: the first batch of the context gives up at the first poll of a lower block's publish word
    (kErrInternal); the host queues it once more by eager launches and the results are the oracle's."""
    monkeypatch.setenv("PCOP_DEBUG_SPIN0", "1")
    p = synth.params_of("C3")
    clouds = [synth.make_cloud("C3", n=120_000, seed=470 + i) for i in range(3)]
    refs = [oracle.process_xyz(c, **p) for c in clouds]
    c = capi.Context(120_000, max_frames=3)
    c.set_params(publish_clouds=1, **p)
    for _ in range(2):
        run(c, clouds, refs)
    assert c.debug_counter("internal_retries") == 1
    single = capi.Context(120_000)            # and a single frame (sort path, one-pass CSR: more look-back kernels)
    single.set_params(publish_clouds=1, **p)
    same(single.process_xyz(clouds[0]), refs[0])
    same(single.process_xyz(clouds[1]), refs[1])
    assert single.debug_counter("internal_retries") == 1
    single.close()
    c.close()


def test_streamed_partition_checks_its_own_order_once(capi, oracle, monkeypatch):
    """The first streamed batch of a context writes the partition twice (ranks from lane-ordered LDS atomics, ranks from peer
    masks) and compares the bytes; they agree on this hardware, and the later batches do not pay for it.  With
    PCOP_DEBUG_ORDER_MISMATCH=1 the comparison "fails" once: the batch is re-run with peer-mask ranks, which the context keeps
    from then on - same bytes out either way."""
    p = synth.params_of("C3")
    clouds = [synth.make_cloud("C3", n=110_000, seed=475 + i) for i in range(3)]
    refs = [oracle.process_xyz(c, **p) for c in clouds]
    c = capi.Context(110_000, max_frames=3)
    c.set_params(publish_clouds=1, **p)
    assert c.debug_counter("lds_ordered") == 1 and c.debug_counter("stream_checked") == 0
    run(c, clouds, refs)
    assert c.debug_counter("lds_ordered") == 1 and c.debug_counter("stream_checked") == 1
    before = c.debug_counter("replans")
    run(c, clouds, refs)
    assert c.debug_counter("replans") == before
    c.close()
    monkeypatch.setenv("PCOP_DEBUG_ORDER_MISMATCH", "1")
    c = capi.Context(110_000, max_frames=3)
    c.set_params(publish_clouds=1, **p)
    run(c, clouds, refs)
    assert c.debug_counter("lds_ordered") == 0 and c.debug_counter("replans") >= 1
    for _ in range(2):
        run(c, clouds, refs)
    assert c.debug_counter("lds_ordered") == 0
    c.close()


@pytest.mark.parametrize("tile", ["4096", "8192", "32768"])
def test_any_tile_size_same_result(capi, oracle, monkeypatch, tile):
    """Points per block of the histogram / partition passes (the host picks the largest that still fills the chip: 2^15 on the
    benchmark's batches): a launch parameter, not part of the result.  Ragged sizes, NaN holes, with and without the box."""
    monkeypatch.setenv("PCOP_BIN_TILE", tile)
    p = synth.params_of("C3")
    clouds = [synth.make_cloud("C3", n=150_000, seed=490), synth.make_cloud("C3", n=98_305, seed=491), synth.make_cloud("C3", n=32_769, seed=492)]
    clouds[1][::77] = np.nan
    refs = [oracle.process_xyz(c, is_dense=False, **p) for c in clouds]
    c = capi.Context(150_000, max_frames=3)
    c.set_params(publish_clouds=1, **p)
    seen = []
    for _ in range(3):
        st = run(c, clouds, refs, is_dense=False)
        seen.append((int(st.sbox), int(st.tile)))
    assert seen == [(0, int(tile)), (1, int(tile)), (1, int(tile))], seen
    c.close()


def test_box_path_single_frames_on_the_sort_path(capi, oracle):
    """One frame at a time (the node's own mode): the sort path; from the second frame on its pass-0 histogram takes the
    bounds (no k_minmax).  The frame API, the stage API (voxel grid alone), NaN holes, a point outside the box."""
    p = synth.params_of("C3")
    a = synth.make_cloud("C3", n=140_000, seed=480)
    b = synth.make_cloud("C3", n=90_000, seed=481)
    c = capi.Context(140_000)
    c.set_params(publish_clouds=1, **p)
    seen = []
    for pts in (a, b, a):
        same(c.process_xyz(pts), oracle.process_xyz(pts, **p))
        st = c.debug_state()
        seen.append(int(st.sbox))
        lo, div = grid_of(pts, p["leaf"])
        assert list(st.min_b) == lo and list(st.div_b) == div and st.vpath == 0
    assert seen == [0, 1, 1], seen
    vox, pt = c.voxel_downsample(b)                     # downsample_cloud alone (od.cpp:271-296)
    want, wpt = oracle.voxel_downsample(b, p["leaf"])
    assert vox.tobytes() == want.tobytes() and pt == wpt and c.debug_state().sbox == 1
    q = a.copy(); q[::61] = np.nan                       # is_dense = false
    same(c.process_xyz(q, is_dense=False), oracle.process_xyz(q, is_dense=False, **p))
    assert c.debug_state().sbox == 1 and c.debug_state().n_valid == int(np.isfinite(q).all(axis=1).sum())
    with pytest.raises(capi.PcopError):
        c.process_xyz(q, is_dense=True)
    far = b.copy(); far[5] = (9.0, 9.0, 3.0)
    same(c.process_xyz(far), oracle.process_xyz(far, **p))
    assert c.debug_state().sbox == 0                     # re-run with the bounds pass
    same(c.process_xyz(far), oracle.process_xyz(far, **p))
    assert c.debug_state().sbox == 1
    empty = np.zeros((0, 3), dtype=f32)
    same(c.process_xyz(empty), oracle.process_xyz(empty, **p))
    same(c.process_xyz(a[:3]), oracle.process_xyz(a[:3], **p))
    c.close()


def test_the_configuration_the_bench_times(capi, oracle):
    """The bench line's configuration at its full size (bench.py: a context of 32 slots, 32 distinct 1 M-point C3 clouds resident in
    HBM, three submit_batch_device rounds: the bounds pass first, then twice with the keys on the context's box, 2^15 points a tile, the
    partition in chunks of 8192 points): frames 0, 15 and 31 of every round bit-exact against the oracle.  Reference call sites:
    od.cpp:740-741, 791-797."""
    p = synth.params_of("C3")
    n = synth.CONFIGS["C3"]["n"]
    B = 32
    seeds = [3000 + i for i in range(B)]
    c = capi.Context(n, max_frames=B)
    c.set_params(publish_clouds=1, **p)
    ptrs, ns, refs = [], [], {}
    for i, s in enumerate(seeds):
        cloud = synth.make_cloud("C3", seed=s)
        q, k = c.upload_soa(cloud)
        ptrs.append(q); ns.append(k)
        if i in (0, 15, B - 1):
            refs[i] = oracle.process_xyz(cloud, **p)
        del cloud
    for rnd in range(3):
        c.submit_batch_device(ptrs, ns)
        got = c.wait_batch()
        st = c.debug_state()
        assert st.vpath == 1, "a batch takes the streaming voxel path"
        if rnd >= 1:
            assert st.sbox == 1 and st.tile == 32768, (rnd, st.sbox, st.tile)
        else:
            assert st.sbox == 0
        for i, r in refs.items():
            same(got[i], r)
    c.close()


def test_single_frames_on_a_box_one_key_bit_longer_do_not_replan_every_call(capi, oracle):
    """A grid of 40^3 voxels has 16-bit keys (two 8-bit passes); the context's box around it (a margin of one voxel at least) has
    42^3 cells = 17 bits (three passes).  Single frames keyed on the box plan on the box's bits: the pass count must stay at what
    the box needs, or every call raises kErrReplanVoxel and runs twice."""
    rng = np.random.default_rng(5)
    pts = rng.uniform(0.0005, 0.3995, (60_000, 3)).astype(f32)
    pts[0] = (0.0005, 0.0005, 0.0005); pts[1] = (0.3995, 0.3995, 0.3995)
    p = dict(leaf=0.01, tol=0.03, min_cluster=1, max_cluster=100000)
    ref = oracle.process_xyz(pts, **p)
    c = capi.Context(len(pts))
    c.set_params(publish_clouds=1, **p)
    seen = []
    for rep in range(6):
        same(c.process_xyz(pts), ref)
        seen.append(c.debug_counter("replans"))
    st = c.debug_state()
    assert tuple(st.div_b) == (40, 40, 40) and st.sbox == 1
    assert seen[-1] == seen[2], seen   # (the first boxed frame may move the pass count once; after that nothing moves)
    c.close()


def test_cloud_far_from_the_origin_keeps_its_bounds_pass(capi, oracle):
    """Voxel coordinates beyond +-2^24 are not exact as floats: (float)box_lo would round and a key formed against the box could be off by
    one.  Such a cloud never gets a box (DevState::sbox stays 0, the bounds pass runs every time) and still equals the oracle."""
    rng = np.random.default_rng(17)
    leaf = 0.001
    base = np.array([17000.0, -17500.0, 16900.0], dtype=np.float64)     # |x / leaf| about 1.7e7 > 2^24 = 1.68e7
    pts = (base + rng.uniform(0.0, 0.35, (40_000, 3))).astype(f32)
    p = dict(leaf=leaf, tol=0.02, min_cluster=1, max_cluster=100000)
    ref = oracle.process_xyz(pts, **p)
    c = capi.Context(len(pts), max_frames=2)
    c.set_params(publish_clouds=1, **p)
    for rep in range(3):
        st = run(c, [pts, pts[::-1].copy()], [ref, oracle.process_xyz(pts[::-1].copy(), **p)])
        assert st.sbox == 0, rep
    c.close()


def test_contexts_fed_from_threads_of_one_process(capi, oracle):
    """Three batch contexts of one process, each fed by its own thread (how bench.py and host/pcop_pool.cpp drive a GPU), with
    batches of different shapes in flight at once: every frame of every round bit-exact against the oracle."""
    import threading
    p = synth.params_of("C3")
    shapes = [(6, 120_000), (4, 200_000), (8, 60_000)]
    ctxs, work = [], []
    for t, (B, n) in enumerate(shapes):
        c = capi.Context(n, max_frames=B)
        c.set_params(publish_clouds=1, **p)
        clouds = [synth.make_cloud("C3", seed=7100 + 10 * t + i, n=n) for i in range(B)]
        refs = [oracle.process_xyz(x, **p) for x in clouds]
        ctxs.append(c); work.append((clouds, refs))
    errs = []

    def feed(c, clouds, refs):
        try:
            for rnd in range(6):
                k = len(clouds) if rnd % 3 else max(2, len(clouds) // 2)   # a shorter batch now and then: another graph
                c.submit_batch_host(clouds[:k])
                got = c.wait_batch()
                for g, r in zip(got, refs[:k]):
                    same(g, r)
        except BaseException as e:   # noqa: BLE001 - handed to the main thread
            errs.append(e)

    th = [threading.Thread(target=feed, args=(c, *w)) for c, w in zip(ctxs, work)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    for c in ctxs:
        c.close()
    if errs:
        raise errs[0]
