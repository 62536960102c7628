"""CPU: the host-side build of the cell-ordered cloud (read_splat_cells_build_host): permutation, padding, bounding
boxes, Morton locality.  The GPU passes that consume it are covered by tests/test_gpu_splat.py (bit-exact vs oracle)."""
import numpy as np

from read_amd import _lib, synthetic
from read_amd.raster import build_cells


def _parse(blob, n):
    nc = (n + 1023) // 1024
    hdr_n = int(np.frombuffer(blob[:8].tobytes(), np.int64)[0])
    nchunks, version = np.frombuffer(blob[8:16].tobytes(), np.int32)
    bbox = np.frombuffer(blob[16:40].tobytes(), np.float32)
    density = float(np.frombuffer(blob[40:44].tobytes(), np.float32)[0])
    o = 256
    rec = np.frombuffer(blob[o:o + nc * 1024 * 16].tobytes(), np.float32).reshape(-1, 4)     # (x, y, z, id bits)
    xs = np.ascontiguousarray(rec[:, :3])
    ids = np.ascontiguousarray(rec[:, 3]).view(np.uint32)
    o += nc * 1024 * 16
    aabb = np.frombuffer(blob[o:o + nc * 32].tobytes(), np.float32).reshape(-1, 8)
    scratch = ((8 * 8 * nc * 4 + 255) // 256) * 256 + 8 * nc * 16 + ((nc + 255) // 256) * 256   # chunk lists: 8 strips x (8 bands of A, B); sticky flags
    assert o + nc * 32 + scratch == len(blob)
    return hdr_n, int(nchunks), int(version), bbox, density, xs, ids, aabb


def test_cells_build_is_a_permutation_with_exact_boxes():
    n = 5000                                                   # 5 chunks, the last one padded
    xyz = synthetic.make_cloud(n, 11)
    blob = build_cells(xyz)
    assert len(blob) == _lib.lib().read_splat_cells_bytes(n)
    hdr_n, nchunks, version, bbox, density, xs, ids, aabb = _parse(blob, n)
    assert (hdr_n, nchunks, version) == (n, 5, 2)
    assert np.array_equal(np.sort(ids[:n]), np.arange(n, dtype=np.uint32))
    assert np.array_equal(xs[:n], xyz[ids[:n]])                # sorted copy = original points, bit for bit
    assert (ids[n:] == ids[n - 1]).all() and (xs[n:] == xs[n - 1]).all()      # tail = copies of the last point
    assert np.array_equal(bbox[:3], xyz.min(0)) and np.array_equal(bbox[3:], xyz.max(0))
    vol = np.prod((xyz.max(0) - xyz.min(0)).astype(np.float64))
    assert abs(density - n / vol) <= 1e-5 * n / vol
    c = xs.reshape(nchunks, 1024, 3)
    assert np.array_equal(aabb[:, :3], c.min(1)) and np.array_equal(aabb[:, 3:6], c.max(1))


def test_cells_are_spatially_compact():
    n = 1 << 16
    xyz = synthetic.make_cloud(n, 5)
    _, nchunks, _, bbox, _, xs, ids, aabb = _parse(build_cells(xyz), n)
    whole = np.prod((bbox[3:] - bbox[:3]).astype(np.float64))
    vols = np.prod((aabb[:, 3:6] - aabb[:, :3]).astype(np.float64), 1)
    assert nchunks == 64 and np.median(vols) < whole / 16      # a random order would give boxes ~ the whole cloud


def test_cells_equal_codes_keep_ascending_ids_and_degenerate_clouds():
    xyz = np.zeros((3000, 3), np.float32)                      # all points identical: stable sort keeps the order
    _, _, _, _, density, xs, ids, _ = _parse(build_cells(xyz), 3000)
    assert np.array_equal(ids[:3000], np.arange(3000, dtype=np.uint32)) and np.isfinite(density) and density > 0
    flat = synthetic.make_cloud(4096, 2)
    flat[:, 1] = 3.0                                           # a plane: finite density
    assert np.isfinite(_parse(build_cells(flat), 4096)[4])
    bad = flat.copy()
    bad[7, 2] = np.nan
    try:
        build_cells(bad)
    except _lib.ReadHipError as e:
        assert "not finite" in str(e)
    else:
        raise AssertionError("non-finite point accepted")


def test_cells_build_threaded_equals_the_stable_sort():
    """Large enough for the multi-threaded radix sort: the order must be the stable sort by Morton code."""
    n = 300_000
    xyz = synthetic.make_cloud(n, 3)
    _, _, _, bbox, _, xs, ids, _ = _parse(build_cells(xyz), n)
    lo = bbox[:3]
    ext = np.float32((bbox[3:] - bbox[:3]).max())
    scale = np.float32(1023.999) / ext
    q = np.clip(((xyz - lo) * scale).astype(np.int64), 0, 1023)           # (x - lo) * scale in fp32, truncated

    def spread(v):
        out = np.zeros_like(v)
        for b in range(10):
            out |= ((v >> b) & 1) << (3 * b)
        return out
    code = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    order = np.argsort(code, kind="stable")
    assert np.array_equal(ids[:n], order.astype(np.uint32))
    assert np.array_equal(xs[:n], xyz[order])
