"""CPU tests of the occupancy oracle (oracle/occupancy_restatement.cpp): the bounds filter and 2-D binning the
reference runs right before the hot path (od.cpp:134-269).  Hand-computable cases + an independent NumPy form."""
import numpy as np
import pytest

f32 = np.float32


@pytest.fixture(scope="module")
def occ(oracle):
    # the defaults of the reference's main(): x -1..1, y -0.5..0.6, z 0..-0.5 (sic: lower 0, upper -0.5), block 0.15
    return oracle.occ_params(-1.0, 1.0, -0.5, 0.6, -0.5, 0.0, 0.15, 0.5)


def np_occupancy(pts, p):
    """Independent formulation: vectorised walks in float32, bincount, integer row means."""
    pts = np.asarray(pts, dtype=f32)
    x, y, z = pts[:, 0], pts[:, 1], pts[:, 2]
    with np.errstate(invalid="ignore"):
        drop = np.isnan(x) | (x < f32(p.x_min)) | (x > f32(p.x_max)) | (z < f32(p.z_min)) | (z > f32(p.z_max)) | \
            (y < f32(p.y_min)) | (y > f32(p.y_max))
    keep = ~drop
    xs, ys = x[keep], y[keep]
    bs = f32(p.block_size)
    col = np.zeros(len(xs), dtype=np.int64)
    row = np.zeros(len(xs), dtype=np.int64)
    with np.errstate(invalid="ignore"):
        for c in range(4096):   # lo + (c+1)*bs < a, as float32 operations
            step = f32(p.y_min) + f32(c + 1) * bs
            go = (col == c) & (step < ys)
            if not go.any():
                break
            col[go] += 1
        for c in range(4096):
            step = f32(p.x_max) - f32(c + 1) * bs
            go = (row == c) & (step > xs)
            if not go.any():
                break
            row[go] += 1
    index = row * p.grid_width + col
    size = p.grid_width * p.grid_height
    ok = index < size
    counts = np.bincount(index[ok], minlength=size).astype(np.int64)
    row_avg = counts.reshape(p.grid_height, p.grid_width).sum(axis=1) // p.grid_width
    limit = row_avg.astype(f32) * (f32(1) - f32(p.dev_percent))
    grid = np.where(counts.reshape(p.grid_height, p.grid_width).astype(f32) < limit[:, None], 100, 0).astype(np.int8).ravel()
    return dict(xyz=pts[keep], counts=counts, row_avg=row_avg, grid=grid, index_errors=int((~ok).sum()))


def same(a, b):
    assert a["xyz"].tobytes() == b["xyz"].tobytes()
    assert np.array_equal(a["counts"], b["counts"]) and np.array_equal(a["row_avg"], b["row_avg"])
    assert np.array_equal(a["grid"], b["grid"]) and a["index_errors"] == b["index_errors"]


def test_grid_dims_follow_the_limits(oracle, occ):
    # ceil((|-0.5| + |0.6|) / 0.15) = ceil(7.33) = 8 columns; ceil((1 + 1) / 0.15) = ceil(13.3) = 14 rows
    assert (occ.grid_width, occ.grid_height) == (8, 14)


def test_cells_by_hand(oracle, occ):
    # column walks up from y_min along the point's y, row walks down from x_max along the point's x
    assert oracle.occ_cell(0.95, -0.45, occ) == 0                 # first column, first row
    assert oracle.occ_cell(-1.0, 0.6, occ) == 13 * 8 + 7          # last cell
    assert oracle.occ_cell(0.0, 0.0, occ) == 6 * 8 + 3            # 1 - 7*0.15 = -0.05 <= 0 -> row 6 ; -0.5 + 4*0.15 = 0.1 >= 0 -> col 3
    assert oracle.occ_cell(0.5, float("nan"), occ) == 3 * 8 + 0   # a NaN y never starts the walk: column 0


def test_nan_quirks_and_order(oracle, occ):
    nan = np.nan
    pts = np.array([[0.0, 0.0, -0.1], [nan, 0.0, -0.1], [0.1, nan, -0.1], [0.2, 0.1, nan], [2.0, 0.0, -0.1],
                    [0.3, 0.0, 0.1], [0.3, 0.7, -0.1], [-0.9, -0.4, -0.5]], dtype=f32)
    r = oracle.occupancy(pts, occ)
    # only x is tested for NaN; NaN y or z fails every comparison and passes (od.cpp:196-199)
    kept = [0, 2, 3, 7]
    assert r["n_out"] == 4 and r["xyz"].tobytes() == pts[kept].tobytes()
    assert r["counts"].sum() == 4 and r["index_errors"] == 0


def test_row_mean_and_hole_marks(oracle):
    p = oracle.OccParams(0.0, 3.0, 0.0, 4.0, -1.0, 1.0, 1.0, 0.5, 4, 3)
    # row 0 is x in (2,3]: columns get 8, 4, 3, 1 points -> mean 16 // 4 = 4 -> limit 2.0 -> only the 1 is a hole
    pts = []
    for col, k in enumerate((8, 4, 3, 1)):
        pts += [[2.5, col + 0.5, 0.0]] * k
    r = oracle.occupancy(np.array(pts, dtype=f32), p)
    assert list(r["counts"][:4]) == [8, 4, 3, 1] and list(r["row_avg"]) == [4, 0, 0]
    assert list(r["grid"][:4]) == [0, 0, 0, 100]
    assert not r["grid"][4:].any()      # empty rows: mean 0, 0 < 0 is false


def test_index_errors_are_kept_but_not_counted(oracle):
    p = oracle.OccParams(-1.0, 1.0, -0.5, 0.6, -0.5, 0.0, 0.15, 0.5, 8, 10)   # four rows short of what the limits need
    pts = np.array([[-0.95, 0.0, -0.1], [0.9, 0.0, -0.1]], dtype=f32)
    r = oracle.occupancy(pts, p)
    assert r["n_out"] == 2 and r["index_errors"] == 1 and r["counts"].sum() == 1


@pytest.mark.parametrize("seed,block", [(1, 0.15), (2, 0.05), (3, 0.011)])
def test_matches_numpy_formulation(oracle, seed, block):
    rng = np.random.default_rng(seed)
    p = oracle.occ_params(-1.0, 1.0, -0.5, 0.6, -0.5, 0.0, block, 0.5)
    pts = np.stack([rng.uniform(-1.3, 1.3, 20000), rng.uniform(-0.7, 0.8, 20000), rng.uniform(-0.6, 0.1, 20000)], axis=1).astype(f32)
    pts[::41, 1] = np.nan
    pts[::53, 0] = np.nan
    pts[::67, 2] = np.nan
    # points exactly on cell edges, as the float walk computes them
    k = np.arange(1, 8, dtype=f32)
    edge = np.stack([f32(1.0) - k * f32(block), f32(-0.5) + k * f32(block), np.full(7, -0.1, dtype=f32)], axis=1)
    pts = np.concatenate([pts, edge]).astype(f32)
    same(oracle.occupancy(pts, p), np_occupancy(pts, p))
