"""Parity of statistical outlier removal on the GPU (pcop_statistical_outlier_removal, through the C ABI) against the
CPU restatement of pcl::StatisticalOutlierRemoval (remove_statistical_outliers, od.cpp:316-340).  The per-point mean
k-NN distances are float values computed by the same operations in the same order: bit-exact; so are the threshold
(formed on the host in the reference's order) and therefore the selection."""
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


@pytest.fixture(scope="module")
def ctx(capi):
    c = capi.Context(400_000)
    yield c
    c.close()


def check(ctx, oracle, pts, k, mul, layout=2):
    got = ctx.statistical_outlier_removal(pts, k, mul, layout=layout)
    ref = oracle.sor(pts, k, mul)
    assert got["distances"].tobytes() == ref["distances"].tobytes()
    assert got["threshold"] == ref["threshold"]
    assert got["xyz"].tobytes() == ref["xyz"].tobytes()          # same points kept, input order
    return got, ref


def test_voxel_cloud_with_the_nodes_parameters(ctx, oracle):
    """The reference's call: meanK 15 on the voxel-grid output (od.cpp:751-754)."""
    vox, _ = oracle.voxel_downsample(synth.make_cloud("C3", n=300_000, seed=2), 0.02)
    for mul in (1.0, 4.0):
        got, ref = check(ctx, oracle, vox, 15, mul)
        assert 0 < len(got["xyz"]) < len(vox)


@pytest.mark.parametrize("layout", [0, 1, 2])
def test_blob_and_stragglers_all_layouts(ctx, oracle, layout):
    rng = np.random.default_rng(4)
    pts = np.concatenate([rng.normal(0, 0.3, (20_000, 3)), rng.uniform(-3, 3, (500, 3))]).astype(f32)
    got, ref = check(ctx, oracle, pts, 8, 0.5, layout=layout)
    assert len(got["xyz"]) < len(pts)


def test_degenerate_shapes_and_duplicates(ctx, oracle):
    line = np.zeros((300, 3), dtype=f32)
    line[:, 0] = np.linspace(0, 10, 300, dtype=f32) ** 2 / f32(10)            # uneven spacing along one axis
    check(ctx, oracle, line, 5, 1.0)
    plane = np.zeros((4000, 3), dtype=f32)
    plane[:, :2] = np.random.default_rng(5).uniform(0, 2, (4000, 2))
    check(ctx, oracle, plane, 15, 1.0)
    dup = np.repeat(np.random.default_rng(6).uniform(0, 1, (500, 3)).astype(f32), 4, axis=0)   # every point four times
    got, _ = check(ctx, oracle, dup, 3, 1.0)
    assert (got["distances"] == 0).all() and len(got["xyz"]) == len(dup)       # three neighbours at distance 0 each
    tiny = np.random.default_rng(7).uniform(0, 1, (16, 3)).astype(f32)         # exactly mean_k + 1 points
    check(ctx, oracle, tiny, 15, 1.0)


@pytest.mark.parametrize("mean_k", [1, 2, 7, 14, 15, 16, 23, 31])
def test_every_neighbour_count(ctx, oracle, mean_k):
    """mean_k + 1 <= 16 and > 16 run different instances of the k-NN kernel (16 / 32 values in registers, the unused
    places pre-filled); a dense blob is settled by the 27 cells around a point, the stragglers and the sparse shell
    around it walk outwards shell by shell."""
    rng = np.random.default_rng(40 + mean_k)
    pts = np.concatenate([rng.normal(0, 0.05, (6000, 3)), rng.uniform(-2, 2, (300, 3)),
                          np.array([[9.0, 9.0, 9.0], [-9.0, 9.0, -9.0]])]).astype(f32)
    check(ctx, oracle, pts[rng.permutation(len(pts))], mean_k, 1.0)


def test_empty_cloud_gives_empty_cloud(ctx):
    """sor2.filter on an empty input returns an empty cloud (od.cpp:316-340 after a bounds filter that kept nothing)."""
    got = ctx.statistical_outlier_removal(np.zeros((0, 3), dtype=f32), 15, 4.0)
    assert len(got["xyz"]) == 0 and len(got["distances"]) == 0


def test_params_cannot_change_under_a_batch(ctx, capi):
    """pcop_set_params between submit and wait would mix two parameter sets in the re-planning and the ordering."""
    pts = synth.make_cloud("C1")
    ctx.set_params(publish_clouds=1, **synth.params_of("C1"))
    ctx.submit_batch_host([pts])
    with pytest.raises(capi.PcopError) as e:
        ctx.set_params(leaf=0.5, tol=0.1)
    assert e.value.status == -8          # PCOP_ERR_STATE
    assert ctx.wait_batch()[0]["num_voxels"] > 0


def test_refusals(ctx, capi):
    pts = np.random.default_rng(8).uniform(0, 1, (100, 3)).astype(f32)
    with pytest.raises(capi.PcopError) as e:
        ctx.statistical_outlier_removal(pts[:10], 15, 1.0)
    assert e.value.status == -2          # PCOP_ERR_PARAM: fewer than mean_k + 1 points
    with pytest.raises(capi.PcopError) as e:
        ctx.statistical_outlier_removal(pts, 40, 1.0)
    assert e.value.status == -2
    bad = pts.copy()
    bad[3, 1] = np.nan
    with pytest.raises(capi.PcopError) as e:
        ctx.statistical_outlier_removal(bad, 15, 1.0)
    assert e.value.status == -6          # PCOP_ERR_NONFINITE
    # the context still works, and the chain still works on it after the k-NN passes borrowed its buffers
    got = ctx.statistical_outlier_removal(pts, 15, 1.0)
    assert len(got["distances"]) == 100
    ctx.set_params(publish_clouds=1, **synth.params_of("C1"))
    r = ctx.process_xyz(synth.make_cloud("C1"))
    assert r["num_voxels"] > 0


def test_product_against_the_numpy_formulation(capi):
    """The product against the brute-force NumPy formulation of tests/test_oracle_sor.py directly (the host's two sequential
    double sums share their text with the C++ oracle): mean distances, threshold and kept points bit for bit."""
    from tests.test_oracle_sor import brute
    rng = np.random.default_rng(21)
    pts = np.concatenate([rng.normal(0, 0.3, (1200, 3)), rng.uniform(-3, 3, (90, 3))]).astype(np.float32)
    c = capi.Context(len(pts))
    for k, mul in ((15, 1.0), (4, 0.25)):
        g = c.statistical_outlier_removal(pts, k, mul)
        dist, thr, keep = brute(pts, k, mul)
        assert g["distances"].tobytes() == dist.tobytes()
        assert g["threshold"] == thr
        assert g["xyz"].tobytes() == pts[keep].tobytes()
    c.close()
