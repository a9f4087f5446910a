"""Edge cases of the scan chain against the oracle: non-finite points (lidar no-returns) through the croppers,
the fused crop + VoxelDownSample and the float32 ingest; f32 normals on EVERY point whose neighbourhood the stored values define."""
import numpy as np
import pytest

from open3d_slam_amd import backend, synthetic as syn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scan():
    return syn.os128_scan(syn.make_scene(), np.eye(4), n_az=256)  # 32 768 raw points


def _with_no_returns(scan, seed=5):
    """a scan in which every 7th..13th ray has no return, encoded every way a driver does it: NaN in all fields, NaN in one, +-inf"""
    pts = scan.copy()
    rng = np.random.default_rng(seed)
    idx = rng.permutation(len(pts))[: len(pts) // 9]
    kinds = rng.integers(0, 6, size=len(idx))
    pts[idx[kinds == 0]] = np.nan
    pts[idx[kinds == 1], 0] = np.nan
    pts[idx[kinds == 2], 2] = np.nan
    pts[idx[kinds == 3], 1] = np.inf
    pts[idx[kinds == 4], 2] = -np.inf
    pts[idx[kinds == 5]] = np.inf
    return pts, np.sort(idx)


CROPS = [
    (backend.CROP_MAX_RADIUS, dict(rmax=12.0)),
    (backend.CROP_MIN_RADIUS, dict(rmin=6.0)),  # keeps +-inf points: inf >= r (croppers.cpp:150)
    (backend.CROP_MIN_MAX_RADIUS, dict(rmin=2.0, rmax=30.0)),
    (backend.CROP_CYLINDER, dict(rmax=15.0, zmin=-1.0, zmax=3.0)),
    (backend.CROP_MAX_RADIUS, dict(rmax=12.0, invert=True)),  # inverted: everything the predicate rejects, NaN included (croppers.cpp:53-55)
    (backend.CROP_MIN_MAX_RADIUS, dict(rmin=2.0, rmax=30.0, invert=True)),
    (backend.CROP_CYLINDER, dict(rmax=15.0, zmin=-1.0, zmax=3.0, invert=True)),
    (backend.CROP_NONE, dict()),
]


@pytest.mark.parametrize("kind,kw", CROPS)
@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_crop_with_non_finite_points_matches_oracle(backend_f64, backend_f32, oracle, scan, kind, kw, prec):
    """croppers.cpp:121-165: every comparison with NaN is false, so a plain volume drops NaN points and an inverted one keeps them;
    |inf| passes `>= radiusMin`.  The device compaction must keep exactly the oracle's index list, in order, non-finite rows included."""
    be = backend_f64 if prec == "f64" else backend_f32
    pts, bad = _with_no_returns(scan)
    stored = pts if prec == "f64" else pts.astype(np.float32).astype(np.float64)
    nrm = np.roll(stored, 1, axis=1)
    c = be.upload(pts, nrm)
    crop = backend.make_crop(kind, center=(0.5, -0.25, 0.1), **kw)
    out = be.crop_cloud(c, crop)
    xyz, n = be.download(out)
    keep = oracle.crop_indices(stored, oracle.make_crop(kind, center=(0.5, -0.25, 0.1), **kw))
    np.testing.assert_array_equal(xyz, stored[keep])  # assert_array_equal: NaN == NaN at the same places
    np.testing.assert_array_equal(n, (nrm if prec == "f64" else nrm.astype(np.float32).astype(np.float64))[keep])
    finite_kept = np.isfinite(stored[keep]).all(axis=1)
    if not kw.get("invert") and kind in (backend.CROP_MAX_RADIUS, backend.CROP_MIN_MAX_RADIUS, backend.CROP_CYLINDER):
        assert finite_kept.all()  # these volumes are a non-finite filter
    else:
        assert (~finite_kept).any()  # ... and these let no-returns through, as the reference does
    be.free(c)
    be.free(out)


@pytest.mark.parametrize("kind,kw", [c for c in CROPS if c[0] in (backend.CROP_MAX_RADIUS, backend.CROP_MIN_MAX_RADIUS, backend.CROP_CYLINDER)
                                      and not c[1].get("invert")])
def test_crop_voxel_down_sample_with_non_finite_points_is_the_oracles_array(backend_f64, oracle, scan, kind, kw):
    """the first two steps of both scan chains (ScanToMapRegistration.cpp:36-37, Odometry.cpp:26-27) on a scan with no-returns: the
    volume removes them, the voxel grid is anchored at the minimum of what is LEFT -- output equal to the oracle's crop -> VoxelDownSample
    bit for bit, order included (a bounding box that looked at the dropped rows would move the grid or poison it)."""
    pts, bad = _with_no_returns(scan)
    c = backend_f64.upload(pts)
    crop = backend.make_crop(kind, **kw)
    out = backend_f64.crop_voxel_down_sample(c, crop, 0.1)
    got, _ = backend_f64.download(out)
    keep = oracle.crop_indices(pts, oracle.make_crop(kind, **kw))
    ref = oracle.voxel_down_sample(pts[keep], 0.1)
    np.testing.assert_array_equal(got, ref)
    clean = backend_f64.upload(pts[np.isfinite(pts).all(axis=1)])
    out2 = backend_f64.crop_voxel_down_sample(clean, crop, 0.1)
    np.testing.assert_array_equal(backend_f64.download(out2)[0], ref)  # and equal to the same scan without the no-return rows
    # the two-call route gives the same
    cropped = backend_f64.crop_cloud(c, crop)
    out3 = backend_f64.voxel_down_sample(cropped, 0.1)
    np.testing.assert_array_equal(backend_f64.download(out3)[0], ref)
    for x in (c, out, clean, out2, cropped, out3):
        backend_f64.free(x)


def test_f32_ingest_with_non_finite_records_then_scan_chain(backend_f32, backend_f64, oracle, scan):
    """open3d_conversions.cpp:59-68 copies NaN / inf fields through unchanged; so must o3ds_cloud_upload_f32.  Then the chain on the
    ingested cloud equals the oracle's on the widened values."""
    pts, bad = _with_no_returns(scan)
    rec = np.zeros((len(pts), 4), dtype=np.float32)
    rec[:, :3] = pts.astype(np.float32)
    rec[:, 3] = 17.0  # intensity
    wide = rec[:, :3].astype(np.float64)
    for be in (backend_f32, backend_f64):
        c = be.upload_f32(rec)
        xyz, _ = be.download(c)
        np.testing.assert_array_equal(xyz, wide)
        crop = backend.make_crop(backend.CROP_MIN_MAX_RADIUS, rmin=2.0, rmax=30.0)
        out = be.crop_cloud(c, crop)
        keep = oracle.crop_indices(wide, oracle.make_crop(oracle.CROP_MIN_MAX_RADIUS, rmin=2.0, rmax=30.0))
        np.testing.assert_array_equal(be.download(out)[0], wide[keep])
        assert np.isfinite(wide[keep]).all() and len(keep) < len(wide) - len(bad) + 1
        vox = be.crop_voxel_down_sample(c, crop, 0.1)
        ref = oracle.voxel_down_sample(wide[keep], 0.1)
        got = be.download(vox)[0]
        if be is backend_f64:
            np.testing.assert_array_equal(got, ref)
        else:
            assert len(got) == len(ref)
            np.testing.assert_allclose(got, ref, atol=4e-6)  # the means are rounded to f32 on the device, same order
        # the whole chain survives: normals on the filtered cloud are finite unit vectors
        be.estimate_normals(vox, 3.0, 20)
        _, n = be.download(vox)
        assert np.isfinite(n).all()
        np.testing.assert_allclose(np.linalg.norm(n, axis=1), 1.0, atol=1e-6)
        for x in (c, out, vox):
            be.free(x)


def _neighbourhoods(stored, radius, knn):
    """For every point of `stored` (the values the device holds): the oracle's hybrid neighbourhood (the knn smallest of d2 < r2 in the
    order (d2, index), computed in f64) and whether the f32 distances the device computes select the SAME set -- true when the gap
    between the last kept and the first rejected candidate, and between every candidate and r2, exceeds the f32 rounding of d2."""
    from scipy.spatial import cKDTree

    tree = cKDTree(stored)
    d, j = tree.query(stored, k=knn + 1, distance_upper_bound=radius * (1 + 1e-6))
    d2 = np.where(np.isfinite(d), d, np.inf) ** 2
    r2 = radius * radius
    eps = 4 * 2.0 ** -23  # three products and two sums in f32 on values <= r2, with margin
    inside = d2 < r2
    count = np.minimum(inside[:, :knn].sum(axis=1), knn)
    # boundary at the radius: no candidate within eps*r2 of r2
    near_r = (np.abs(d2 - r2) <= eps * r2).any(axis=1)
    # boundary at the knn-th place: the (knn+1)-th candidate (if inside) is clearly farther than the knn-th
    last, nxt = d2[:, knn - 1], d2[:, knn]
    both = np.isfinite(last) & np.isfinite(nxt)  # (inf - inf is not a comparison: a missing candidate is no tie)
    gap = np.where(both, np.where(both, nxt, 1.0) - np.where(both, last, 0.0), np.inf)
    tie = inside[:, knn] & both & (gap <= eps * np.where(both, nxt, 1.0))
    return j[:, :knn], inside[:, :knn], count, ~(near_r | tie)


def test_estimate_normals_f32_every_defined_point(backend_f32, oracle, scan):
    """f32 storage, ALL points: wherever the stored values select the same neighbour set under f32 and f64
    distances, the device normal must be the oracle's normal of the stored values -- as a direction wherever the covariance has a
    simple smallest eigenvalue (|dot| >= 1 - 1e-9 x conditioning), and as an eigenvector of the oracle's covariance everywhere else
    (residual test), so degenerate neighbourhoods are covered too.  No fractions: every such point is asserted."""
    pts = oracle.voxel_down_sample(scan, 0.1)
    c = backend_f32.upload(pts)
    stored, _ = backend_f32.download(c)
    for radius, knn in ((3.0, 20), (1.0, 5), (0.5, 30)):
        backend_f32.estimate_normals(c, radius, knn)
        _, got = backend_f32.download(c)
        ref = oracle.estimate_normals(stored, radius, knn)
        j, inside, count, same_set = _neighbourhoods(stored, radius, knn)
        assert same_set.mean() > 0.98, same_set.mean()
        np.testing.assert_allclose(np.linalg.norm(got, axis=1), 1.0, atol=1e-6)
        # covariance of the oracle's neighbourhood, exactly as EstimateNormals forms it (cumulants of the raw coordinates)
        nb = np.where(inside[:, :, None], stored[np.where(inside, j, 0)], 0.0)
        k = np.maximum(count, 1)[:, None]
        mu = nb.sum(axis=1) / k
        cov = np.einsum("nki,nkj->nij", nb, nb) / k[:, :, None] - np.einsum("ni,nj->nij", mu, mu)
        w = np.linalg.eigvalsh(cov)
        enough = count >= 3
        gap = (w[:, 1] - w[:, 0]) / np.maximum(w[:, 2], 1e-300)
        sin_angle = np.linalg.norm(np.cross(got, ref), axis=1) / np.linalg.norm(got, axis=1)
        # (1) simple smallest eigenvalue: same direction.  What may differ: the f32 rounding of the stored normal (<= 1.1e-7 rad) and
        #     the summation order of the cumulants where f32 distances rank near-equal candidates differently (entries of E[x x^T] carry
        #     ~1e-16 * |x|^2 <= 1e-13 m^2 of rounding; an eigenvector moves by that over the eigenvalue gap)
        simple = same_set & enough & (gap > 1e-3)
        tol = 3e-7 + 1e-11 / np.maximum(w[:, 1] - w[:, 0], 1e-300)
        bad = np.flatnonzero(simple & (sin_angle > tol))
        assert len(bad) == 0, (radius, knn, len(bad), bad[:5], sin_angle[bad[:5]], gap[bad[:5]])
        assert simple.mean() > 0.25  # the rest are ring segments and other neighbourhoods without a defined direction: covered by (2)
        # (2) everywhere the set is the same: the device normal lies in the oracle covariance's smallest eigenspace up to the gap
        res = np.einsum("nij,nj->ni", cov, got) - w[:, 0:1] * got
        rel = np.linalg.norm(res, axis=1) / np.maximum(w[:, 2], 1e-300)
        loose = same_set & enough
        bad2 = np.flatnonzero(loose & (rel > np.maximum(2.0 * gap, 1e-5) + 1e-5))
        assert len(bad2) == 0, (radius, knn, len(bad2), bad2[:5], rel[bad2[:5]], gap[bad2[:5]])
        # (3) fewer than three neighbours: the defined fallback (0, 0, 1) oriented towards the origin
        few = same_set & ~enough
        np.testing.assert_array_equal(got[few], ref[few])
        # (4) orientation towards the sensor for every point with a defined direction that is not edge-on
        view = np.einsum("ij,ij->i", got, -stored / np.linalg.norm(stored, axis=1, keepdims=True))
        edge_on = np.abs(np.einsum("ij,ij->i", ref, stored / np.linalg.norm(stored, axis=1, keepdims=True))) < 1e-4
        assert (view[simple & ~edge_on] > 0).all()
    backend_f32.free(c)
