"""Parity of the HIP rasteriser (read_splat_forward through the C ABI) with the oracle:
bit-exact int32 index + fp32 depth bit patterns."""
import os

import numpy as np
import pytest
import torch

import oracle
from read_amd import camera, synthetic
from read_amd.raster import PointCloudRasterizer, index_to_float

pytestmark = pytest.mark.gpu


def _check(xyz, Ms, W, H, levels=5, threads=8):
    r = PointCloudRasterizer(xyz)
    idx, dep = r.render(Ms, W, H, levels)
    torch.cuda.synchronize()
    Ms = np.asarray(Ms, np.float32).reshape(-1, 4, 4)
    for b in range(Ms.shape[0]):
        oi, od = oracle.raster_multiscale(xyz, Ms[b], W, H, levels, threads=threads)
        for l in range(levels):
            gi, gd = idx[l][b].cpu().numpy(), dep[l][b].cpu().numpy()
            assert gi.shape == oi[l].shape
            assert np.array_equal(gi, oi[l]), f"index mismatch cam {b} level {l}: {(gi != oi[l]).sum()} px"
            assert np.array_equal(gd.view(np.uint32), od[l].view(np.uint32)), f"depth mismatch cam {b} level {l}"
    return r, idx, dep


def test_config0_100k_256(golden_dir, hip):
    """BASELINE.json configs[0]: 100k random points, 256x256 — vs the oracle AND the committed golden
    (= the reference's DepthProject source run serially)."""
    g = np.load(os.path.join(golden_dir, "raster_256_100k.npz"))
    xyz = synthetic.make_cloud(int(g["N"]), int(g["seed"]))
    r, idx, dep = _check(xyz, g["M"], 256, 256)
    for l in range(5):
        assert np.array_equal(idx[l].cpu().numpy(), g[f"index{l}"])
        assert np.array_equal(dep[l].cpu().numpy().view(np.uint32), g[f"depth{l}"].view(np.uint32))
    # API-edge float index (point_render.cu:158)
    assert np.array_equal(index_to_float(idx[0]).cpu().numpy(), g["index0"].astype(np.float32))


def test_1216x352_2M_two_cameras(hip):
    W, H = 1216, 352
    xyz = synthetic.make_cloud(2_000_003)                 # n % 4 != 0 exercises the scalar tail
    Ms = camera.total_matrix(synthetic.make_proj(W, H), np.stack([synthetic.sweep_pose(0), synthetic.sweep_pose(33)]))
    _check(xyz, Ms, W, H)


def test_more_cameras_than_one_pass(hip):
    W, H = 64, 48
    xyz = synthetic.make_cloud(20_000)
    Ms = camera.total_matrix(synthetic.make_proj(W, H, f=40.0), np.stack([synthetic.sweep_pose(k) for k in range(11)]))
    _check(xyz, Ms, W, H, threads=1)


def test_generic_size_falls_back_to_per_level_passes(hip):
    W, H = 250, 130                                        # not multiples of 16: pyramid identity does not hold
    xyz = synthetic.make_cloud(150_000)
    Ms = camera.total_matrix(synthetic.make_proj(W, H, f=200.0), synthetic.sweep_pose(5))
    _check(xyz, Ms, W, H)


def test_single_level_any_size(hip):
    W, H = 101, 77
    xyz = synthetic.make_cloud(30_000)
    Ms = camera.total_matrix(synthetic.make_proj(W, H, f=80.0), np.eye(4, dtype=np.float32))
    _check(xyz, Ms, W, H, levels=1)


def test_ties_resolve_to_smallest_index(hip):
    W, H = 64, 64
    base = synthetic.make_cloud(4_000, seed=7)
    xyz = np.concatenate([base, base[::-1], base])         # every point three times -> exact depth ties
    Ms = camera.total_matrix(synthetic.make_proj(W, H, f=64.0), np.eye(4, dtype=np.float32))
    _check(xyz, Ms, W, H)


def test_empty_and_invisible_clouds(hip):
    W, H = 64, 32
    Ms = camera.total_matrix(synthetic.make_proj(W, H, f=64.0), np.eye(4, dtype=np.float32))
    behind = synthetic.make_cloud(5_000)
    behind[:, 2] *= -1                                      # all behind the camera
    for xyz in (np.zeros((0, 3), np.float32), behind):
        r = PointCloudRasterizer(xyz)
        idx, dep = r.render(Ms, W, H, 5)
        for i, d in zip(idx, dep):
            assert int(i.abs().sum()) == 0 and float(d.abs().sum()) == 0.0


def test_workspace_is_left_clean_and_render_is_idempotent(hip):
    W, H = 256, 256
    xyz = synthetic.make_cloud(100_000)
    Ms = camera.total_matrix(synthetic.make_proj(W, H, f=256.0), synthetic.sweep_pose(9))
    r = PointCloudRasterizer(xyz)
    a_i, a_d = r.render(Ms, W, H, 5)
    b_i, b_d = r.render(Ms, W, H, 5)
    for x, y in zip(a_i + a_d, b_i + b_d):
        assert torch.equal(x, y)
    keys = r._ws[8192:8192 + W * H * 8].view(torch.int64)          # behind the 8192-byte header
    assert bool((keys == -1).all()), "key images must be EMPTY after a frame"


def test_warm_start_sequence_is_exact(hip):
    """Consecutive poses through ONE rasteriser: from the second frame on, the previous winners seed the
    key image and the LDS hierarchical-Z rejects most points — results must stay bit-exact."""
    W, H = 304, 176
    xyz = synthetic.make_cloud(1_500_000)
    proj = synthetic.make_proj(W, H, f=180.0)
    r = PointCloudRasterizer(xyz)
    for k in (0, 1, 2, 40, 41, 200):                       # small steps and large jumps
        M = camera.total_matrix(proj, synthetic.sweep_pose(k))
        idx, dep = r.render(M, W, H, 5)
        oi, od = oracle.raster_multiscale(xyz, M[0], W, H, 5, threads=8)
        for l in range(5):
            assert np.array_equal(idx[l][0].cpu().numpy(), oi[l]), f"pose {k} level {l}"
            assert np.array_equal(dep[l][0].cpu().numpy().view(np.uint32), od[l].view(np.uint32))
    # a batch of two cameras through the same object uses its own workspace and the plain path
    Ms = camera.total_matrix(proj, np.stack([synthetic.sweep_pose(3), synthetic.sweep_pose(77)]))
    idx, dep = r.render(Ms, W, H, 5)
    for b in range(2):
        oi, od = oracle.raster_multiscale(xyz, Ms[b], W, H, 5, threads=8)
        assert np.array_equal(idx[0][b].cpu().numpy(), oi[0])


def _exact(r, xyz, proj, k, W, H, nxt=None, what=""):
    M = camera.total_matrix(proj, synthetic.sweep_pose(k))
    Mn = None if nxt is None else camera.total_matrix(proj, synthetic.sweep_pose(nxt))
    idx, dep = r.render(M, W, H, 5, next_total=Mn)
    oi, od = oracle.raster_multiscale(xyz, M[0], W, H, 5, threads=8)
    for l in range(5):
        assert np.array_equal(idx[l][0].cpu().numpy(), oi[l]), f"{what} pose {k} (announced next {nxt}) level {l}"
        assert np.array_equal(dep[l][0].cpu().numpy().view(np.uint32), od[l].view(np.uint32)), f"{what} pose {k} level {l} depth"


def test_announced_next_camera_is_exact_whatever_comes_next(hip):
    """read_splat_hint_next_camera (round 5): with the next camera announced, a cell-path frame's resolve launch also classifies
    the chunks and seeds the depth bounds of the next frame (4 dependent launches per frame instead of 5).  Bit-exact against the
    oracle for: a correctly announced sweep; WRONG announcements (the prepared set must be wiped, not used); an announcement
    followed by a batch call and by another size on the same object; hints withdrawn; the knob off; and a profiled frame reports
    no seed / classification launch when the previous frame did that work."""
    import ctypes as C
    from read_amd import _lib
    L = _lib.lib()
    W, H = 304, 176
    xyz = synthetic.make_cloud(1_500_000)
    proj = synthetic.make_proj(W, H, f=180.0)
    r = PointCloudRasterizer(xyz)
    assert r.cells is not None
    seq = [0, 1, 2, 3, 40, 41, 42, 200, 201]
    for i, k in enumerate(seq):                                  # every frame announces the true next pose
        _exact(r, xyz, proj, k, W, H, nxt=seq[i + 1] if i + 1 < len(seq) else None, what="announced")
    for k, wrong in ((5, 90), (6, 6), (7, 150), (150, 8)):      # announcements that do not come true (and one that repeats itself)
        _exact(r, xyz, proj, k, W, H, nxt=wrong, what="wrong announcement")
    _exact(r, xyz, proj, 9, W, H, nxt=10, what="before a batch")
    Ms = camera.total_matrix(proj, np.stack([synthetic.sweep_pose(3), synthetic.sweep_pose(77)]))
    idx, _ = r.render(Ms, W, H, 5)                               # two cameras: own workspace, two cell-path frames over the SAME blob
    for b in range(2):
        assert np.array_equal(idx[0][b].cpu().numpy(), oracle.raster_multiscale(xyz, Ms[b], W, H, 5, threads=8)[0][0])
    _exact(r, xyz, proj, 10, W, H, what="announced, then a batch over the blob: the preparation is stale")
    try:                                                         # ... and the same batch on the plain pass (rounds 1-4)
        _lib.check(L.read_tuning_set(b"splat_cells_batch", 0))
        idx, _ = r.render(Ms, W, H, 5)
        for b in range(2):
            assert np.array_equal(idx[0][b].cpu().numpy(), oracle.raster_multiscale(xyz, Ms[b], W, H, 5, threads=8)[0][0])
    finally:
        _lib.check(L.read_tuning_set(b"splat_cells_batch", 1))
    proj2 = synthetic.make_proj(256, 128, f=150.0)
    _exact(r, xyz, proj, 11, W, H, nxt=12, what="before another size")
    _exact(r, xyz, proj2, 12, 256, 128, nxt=13, what="other size")
    _exact(r, xyz, proj2, 13, 256, 128, what="other size, announced")
    _exact(r, xyz, proj, 12, W, H, what="back: the old preparation is stale (its chunk lists live in the shared cell blob)")
    # the profile shows which launches a frame made
    try:
        _lib.check(L.read_tuning_set(b"splat_prof", 1))
        ms = (C.c_float * 5)()
        _exact(r, xyz, proj, 20, W, H, nxt=21, what="profiled")
        _lib.check(L.read_splat_profile_last(ms), "read_splat_profile_last")
        assert ms[0] > 0 and all(ms[i] > 0 for i in (1, 2, 3, 4)), list(ms)     # nobody prepared pose 20: five launches
        _exact(r, xyz, proj, 21, W, H, nxt=22, what="profiled")
        _lib.check(L.read_splat_profile_last(ms), "read_splat_profile_last")
        assert ms[0] == 0 and all(ms[i] > 0 for i in (1, 2, 3, 4)), list(ms)    # pose 21 was prepared by pose 20's resolve: four
        _lib.check(L.read_tuning_set(b"splat_ahead", 0))
        _exact(r, xyz, proj, 22, W, H, nxt=23, what="knob off")               # prepared by 21 -> consumed; prepares nothing
        _exact(r, xyz, proj, 23, W, H, nxt=24, what="knob off")
        _lib.check(L.read_splat_profile_last(ms), "read_splat_profile_last")
        assert ms[0] > 0, list(ms)
        # the promotion of front chunks into list A (splat_mark / splat_sticky): every setting, small steps, a jump, and back
        for mark, sticky in ((0, 8), (1, 1), (1, 8), (1, 0), (0, 0)):
            _lib.check(L.read_tuning_set(b"splat_mark", mark))
            _lib.check(L.read_tuning_set(b"splat_sticky", sticky))
            _lib.check(L.read_tuning_set(b"splat_compact", sticky & 1))          # candidates compacted before binning / four masked slots
            seq = [60, 61, 62, 63, 64, 180, 181, 64, 65]
            for i, k in enumerate(seq):
                _exact(r, xyz, proj, k, W, H, nxt=seq[i + 1] if (i + 1 < len(seq) and i % 3 != 2) else None, what=f"mark={mark} sticky={sticky}")
    finally:
        _lib.check(L.read_tuning_set(b"splat_prof", 0))
        _lib.check(L.read_tuning_set(b"splat_ahead", 1))
        _lib.check(L.read_tuning_set(b"splat_mark", 1))
        _lib.check(L.read_tuning_set(b"splat_sticky", 1))
        _lib.check(L.read_tuning_set(b"splat_compact", 1))


def test_camera_plane_sides_of_the_chunk_boxes(hip):
    """The chunk classification examines a box wholly BEHIND the camera plane (every corner w < 0) like one in front of it
    (round 5; rounds 2-4 walked such chunks point by point).  (1) Camera deep inside the cloud, most of it behind the camera:
    bit-exact, and pass A now lists only what lies ahead.  (2) The negated matrix -M gives every point the opposite sign of w and
    the same ratios clip / w — the reference's acceptance rule (point_render.cu:139) only sees the ratios, so -M must render
    exactly M's image: here the boxes ahead of the camera are the ones with w < 0."""
    from read_amd import _lib
    L = _lib.lib()
    W, H = 304, 176
    xyz = synthetic.make_cloud(1_500_000)
    proj = synthetic.make_proj(W, H, f=180.0)
    r = PointCloudRasterizer(xyz)
    assert r.cells is not None
    try:
        _lib.check(L.read_tuning_set(b"splat_stats", 1))
        items = {}
        for k in (5, 250):                                        # sweep_pose(250): camera at z = -75 of a slab that spans -120 .. -1
            r.render(camera.total_matrix(proj, synthetic.sweep_pose(k)), W, H, 5)         # settle lists / seeds
            torch.cuda.synchronize()
            before = r._ws[64:64 + 128].view(torch.int64).clone()
            _exact(r, xyz, proj, k, W, H, what="inside the cloud")
            torch.cuda.synchronize()
            items[k] = int((r._ws[64:64 + 128].view(torch.int64) - before)[8])               # pass-A items of that frame
        assert items[250] < 0.8 * items[5], items                  # fewer chunks ahead, not more (rounds 2-4: 3.5x as many)
    finally:
        _lib.check(L.read_tuning_set(b"splat_stats", 0))
    for k in (3, 130, 250):
        M = camera.total_matrix(proj, synthetic.sweep_pose(k))
        oi, od = oracle.raster_multiscale(xyz, M[0], W, H, 5, threads=8)
        oin, odn = oracle.raster_multiscale(xyz, -M[0], W, H, 5, threads=8)
        idx, dep = r.render(-M, W, H, 5)
        for l in range(5):
            assert np.array_equal(oi[l], oin[l]) and np.array_equal(od[l].view(np.uint32), odn[l].view(np.uint32)), "the oracle itself"
            assert np.array_equal(idx[l][0].cpu().numpy(), oi[l]), f"-M pose {k} level {l}"
            assert np.array_equal(dep[l][0].cpu().numpy().view(np.uint32), od[l].view(np.uint32))


def test_full_size_30M_properties(hip):
    """BASELINE configs[2] size: properties that do not need the (slow) full oracle — the on-device
    pyramid identity, and exactness against the oracle on a 3 M-point subset merged by key-min."""
    W, H, N = 1216, 352, 30_000_000
    xyz = synthetic.make_cloud(N)
    Ms = camera.total_matrix(synthetic.make_proj(W, H), np.eye(4, dtype=np.float32))
    r = PointCloudRasterizer(xyz)
    idx, dep = r.render(Ms, W, H, 5)
    key = (dep[0][0].view(torch.int32).to(torch.int64) << 32) | idx[0][0].to(torch.int64)
    key[(idx[0][0] == 0) & (dep[0][0] == 0)] = torch.iinfo(torch.int64).max
    for l in range(1, 5):
        h, w = key.shape
        key = key.view(h // 2, 2, w // 2, 2).amin(dim=(1, 3))
        kl = (dep[l][0].view(torch.int32).to(torch.int64) << 32) | idx[l][0].to(torch.int64)
        kl[(idx[l][0] == 0) & (dep[l][0] == 0)] = torch.iinfo(torch.int64).max
        assert torch.equal(key, kl), f"level {l} is not the 2x2 key-min of level {l-1}"
    # exactness: the winner over all N equals the key-min of the winners of 10 disjoint 3M slices (oracle)
    best_d = np.full((H, W), np.inf, np.float32)
    best_i = np.zeros((H, W), np.int64)
    step = 3_000_000
    for s in range(0, N, step):
        oi, od = oracle.raster_level(xyz[s:s + step], Ms[0], W, H, threads=8)
        hit = (od > 0) | (oi > 0)
        gi = oi.astype(np.int64) + s
        better = hit & ((od < best_d) | ((od == best_d) & (gi < best_i)))
        best_d[better] = od[better]
        best_i[better] = gi[better]
    best_d[np.isinf(best_d)] = 0
    assert np.array_equal(idx[0][0].cpu().numpy().astype(np.int64), best_i)
    assert np.array_equal(dep[0][0].cpu().numpy().view(np.uint32), best_d.view(np.uint32))


def test_cell_ordered_passes_are_exact_for_hard_cameras(hip):
    """The cell-ordered path (Morton chunks, frustum / hi-Z chunk culling; n >= 2^20) against the oracle for cameras
    that stress the conservative chunk tests: inside the cloud (boxes straddling the camera plane), rolled and pitched
    views, a far-away camera (everything beyond the pass-A split), large world offsets (fp32 projection noise), and
    every tuning of the split; then against the plain path on the same object."""
    from read_amd import _lib
    W, H = 304, 176
    rng = np.random.default_rng(17)
    xyz = synthetic.make_cloud(1_200_000, seed=23)
    proj = synthetic.make_proj(W, H, f=180.0)

    def pose(tx, ty, tz, yaw, pitch, roll):
        cy, sy, cp, sp, cr, sr = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
        Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
        Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
        Rz = np.array([[cr, -sr, 0], [sr, cr, 0], [0, 0, 1]])
        m = np.eye(4, dtype=np.float32)
        m[:3, :3] = (Ry @ Rx @ Rz).astype(np.float32)
        m[:3, 3] = [tx, ty, tz]
        return m

    poses = [pose(0, 0, 0, 0, 0, 0), pose(3, 2, -60, 0.4, -0.1, 0.3), pose(-20, 5, -100, 2.5, 0.2, -1.0),
             pose(0, 0, 400, 0, 0, 0), pose(10, 30, -60, 0.1, -1.2, 0.0), pose(0, 4, -119.5, 3.14, 0, 0)]
    r = PointCloudRasterizer(xyz)
    assert r.cells is not None
    L = _lib.lib()
    try:
        for near in (12, 1, 200):
            _lib.check(L.read_tuning_set(b"splat_near", near))
            for k, p in enumerate(poses):
                M = camera.total_matrix(proj, p)
                idx, dep = r.render(M, W, H, 5)
                oi, od = oracle.raster_multiscale(xyz, M[0], W, H, 5, threads=8)
                for l in range(5):
                    assert np.array_equal(idx[l][0].cpu().numpy(), oi[l]), f"near {near} pose {k} level {l}"
                    assert np.array_equal(dep[l][0].cpu().numpy().view(np.uint32), od[l].view(np.uint32))
    finally:
        _lib.check(L.read_tuning_set(b"splat_near", 12))
    # work-item granularity of the striped passes, no warm start, every 32nd chunk in pass A on every frame (rounds 2-4)
    try:
        for key, val in ((b"splat_items", 2), (b"splat_items", 1), (b"splat_seeds", 0), (b"splat_cells_sub", 32), (b"splat_strips", 8),
                         (b"splat_strips", 2), (b"splat_zl2", 1), (b"splat_lds", 0), (b"splat_kslot", 1), (b"splat_kslot", 2), (b"splat_bins", 0)):
            _lib.check(L.read_tuning_set(key, val))
            for k in (1, 2, 5):
                M = camera.total_matrix(proj, poses[k])
                idx, dep = r.render(M, W, H, 5)
                oi, od = oracle.raster_multiscale(xyz, M[0], W, H, 5, threads=8)
                for l in range(5):
                    assert np.array_equal(idx[l][0].cpu().numpy(), oi[l]), f"{key} {val} pose {k} level {l}"
                    assert np.array_equal(dep[l][0].cpu().numpy().view(np.uint32), od[l].view(np.uint32))
            for k_, v_ in ((b"splat_items", 4), (b"splat_seeds", 1), (b"splat_cells_sub", 0), (b"splat_strips", 1), (b"splat_zl2", 0), (b"splat_lds", 1), (b"splat_kslot", 0), (b"splat_bins", 1)):
                _lib.check(L.read_tuning_set(k_, v_))
    finally:
        for k_, v_ in ((b"splat_items", 4), (b"splat_seeds", 1), (b"splat_cells_sub", 0), (b"splat_strips", 1), (b"splat_zl2", 0), (b"splat_lds", 1), (b"splat_kslot", 0), (b"splat_bins", 1)):
            _lib.check(L.read_tuning_set(k_, v_))
    # large world coordinates: the same cloud and camera moved 5 km away (projection rounding grows ~100x)
    off = np.array([5000.0, -3000.0, 4000.0], np.float32)
    far = PointCloudRasterizer(xyz + off)
    p = pose(3, 2, -60, 0.4, -0.1, 0.3)
    p[:3, 3] += off
    M = camera.total_matrix(proj, p)
    for _ in range(2):                                        # cold, then warm-started
        idx, dep = far.render(M, W, H, 5)
        oi, od = oracle.raster_multiscale(xyz + off, M[0], W, H, 5, threads=8)
        assert np.array_equal(idx[0][0].cpu().numpy(), oi[0]) and np.array_equal(idx[4][0].cpu().numpy(), oi[4])
        assert np.array_equal(dep[0][0].cpu().numpy().view(np.uint32), od[0].view(np.uint32))
    # and the plain path of the same object agrees bit for bit
    try:
        _lib.check(L.read_tuning_set(b"splat_cells", 0))
        idx2, dep2 = far.render(M, W, H, 5)
    finally:
        _lib.check(L.read_tuning_set(b"splat_cells", 1))
    assert all(torch.equal(a, b) for a, b in zip(idx, idx2)) and all(torch.equal(a, b) for a, b in zip(dep, dep2))


def test_bins_overflow_falls_back_to_atomics(hip):
    """2 M points on a 64x32 image: two 32x32 tiles, so every sub-bin of pass A (256 records) overflows many times over and
    most candidates take the fallback atomic on the key image; the merge must combine both (bit-exact, cold and warm)."""
    W, H = 64, 32
    xyz = synthetic.make_cloud(1 << 21, 77)
    proj = synthetic.make_proj(W, H)
    r = PointCloudRasterizer(xyz)
    assert r.cells is not None
    for k in (0, 1, 1, 30):
        M = camera.total_matrix(proj, synthetic.sweep_pose(k))
        idx, dep = r.render(M, W, H, 5)
        oi, od = oracle.raster_multiscale(xyz, M[0], W, H, 5, threads=8)
        for l in range(5):
            assert np.array_equal(idx[l][0].cpu().numpy(), oi[l]), f"pose {k} level {l}"
            assert np.array_equal(dep[l][0].cpu().numpy().view(np.uint32), od[l].view(np.uint32))


def _assert_frame(idx, dep, xyz, M, W, H, what, threads):
    oi, od = oracle.raster_multiscale(xyz, M, W, H, 5, threads=threads)
    for l in range(5):
        gi, gd = idx[l][0].cpu().numpy(), dep[l][0].cpu().numpy()
        assert np.array_equal(gi, oi[l]), f"{what} level {l}: {(gi != oi[l]).sum()} index px differ"
        assert np.array_equal(gd.view(np.uint32), od[l].view(np.uint32)), f"{what} level {l}: depth differs"
    return oi, od


def test_surface_like_street_scene_sequence(hip):
    """A surface-like cloud (road, facades with recesses, vehicles, foliage blobs: synthetic.make_street_cloud — the
    stand-in for BASELINE configs[1]) at the kitti6 viewport 1216x368 (downloads/kitti6.yaml:1; 368 is not a multiple
    of 32): real occlusion, empty sky, two orders of magnitude of density contrast.  Warm-started pose sequence with
    small steps and a jump, bit-exact against the oracle; then the same poses backwards (seeds from a farther view)."""
    W, H = 1216, 368
    xyz = synthetic.make_street_cloud(3_000_000)
    proj = synthetic.make_proj(W, H)
    r = PointCloudRasterizer(xyz)
    assert r.cells is not None
    cov = None
    for k in (0, 1, 2, 40, 41, 2, 0):
        M = camera.total_matrix(proj, synthetic.sweep_pose(k))
        idx, dep = r.render(M, W, H, 5)
        oi, od = _assert_frame(idx, dep, xyz, M[0], W, H, f"street pose {k}", threads=8)
        cov = float((od[0] > 0).mean())
    assert 0.2 < cov < 0.95                                   # there is sky and there is geometry


def test_full_size_30M_warm_started_sweep_vs_oracle(hip):
    """BASELINE configs[2] exactly as bench.py times it: 30 M points, 1216x352, consecutive poses of the sweep through
    ONE warm rasteriser (poses 0, 1, 2, then a jump to 40, 41) — every level bit-exact against the threaded oracle."""
    W, H, N = 1216, 352, 30_000_000
    xyz = synthetic.make_cloud(N)
    proj = synthetic.make_proj(W, H)
    r = PointCloudRasterizer(xyz)
    threads = min(os.cpu_count() or 1, 64)
    for k in (0, 1, 2, 40, 41):
        M = camera.total_matrix(proj, synthetic.sweep_pose(k))
        idx, dep = r.render(M, W, H, 5)
        _assert_frame(idx, dep, xyz, M[0], W, H, f"30M pose {k}", threads=threads)


def _project_on_device(xyz, M, W, H):
    import ctypes as C
    from read_amd import _lib
    pts = torch.from_numpy(np.ascontiguousarray(xyz, np.float32)).cuda()
    pix = torch.empty(pts.shape[0], dtype=torch.int32, device="cuda")
    dep = torch.empty(pts.shape[0], dtype=torch.float32, device="cuda")
    Mh = np.ascontiguousarray(M, np.float32).reshape(16)
    _lib.check(_lib.lib().read_splat_project_points(pts.data_ptr(), pts.shape[0], Mh.ctypes.data_as(C.POINTER(C.c_float)), W, H,
                                                    pix.data_ptr(), dep.data_ptr(), _lib.stream_ptr()), "read_splat_project_points")
    torch.cuda.synchronize()
    return pix.cpu().numpy(), dep.cpu().numpy()


def test_shared_reciprocal_projection_is_ieee_division(hip):
    """the rasteriser computes c0/c3, c1/c3, c2/c3 with ONE reciprocal (csrc/splat.hip div3_ieee) instead of three
    IEEE divisions.  More than 10^8 points, device against the oracle's IEEE divisions (oracle/raster.c project_point, the
    restatement pinned to the reference's own source): the accept / reject decision and the pixel of EVERY point, the depth bits
    of every accepted point.  Operands are adversarial on purpose: a matrix that hands (x, y, z) straight to the divisions
    (numerators and divisor chosen bit by bit: divisors across and exactly on the edges 2^-40 / 2^40 of the fast path's window,
    zero, denormal, inf, NaN; numerators zero, denormal, around 2^-103, 2^56 |b| and beyond, equal to the divisor and one ulp
    off it, quotients within an ulp of +-1), real cameras, and real cameras scaled by 2^k so that whole clouds sit on the
    window's edges (the projection is invariant under the scale, the instruction path is not)."""
    W, H = 1216, 352
    cpu = min(os.cpu_count() or 1, 32)
    rng = np.random.default_rng(2024)
    total = 0

    def compare(xyz, M, what):
        nonlocal total
        got_p, got_d = _project_on_device(xyz, M, W, H)
        with np.errstate(all="ignore"):
            ref_p, ref_d = oracle.project_points(xyz, M, W, H, threads=cpu)
        bad = np.flatnonzero(got_p != ref_p)
        assert bad.size == 0, (what, bad.size, xyz[bad[:4]], got_p[bad[:4]], ref_p[bad[:4]])
        acc = ref_p >= 0
        badd = np.flatnonzero(acc & (got_d.view(np.uint32) != ref_d.view(np.uint32)))
        assert badd.size == 0, (what, badd.size, xyz[badd[:4]], got_d[badd[:4]], ref_d[badd[:4]])
        total += xyz.shape[0]
        return float(acc.mean())

    # ---- (x, y, z) -> numerators x, y, 0.5 x + 0.25 y over the divisor z
    direct = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0.5, 0.25, 0, 0], [0, 0, 1, 0]], np.float32)
    n = 1 << 25
    edges = np.array([2.0 ** -40, 2.0 ** 40, 2.0 ** -126, 2.0 ** 126, 2.0 ** -103, 2.0 ** -41, 2.0 ** 41], np.float32)
    edges = np.concatenate([edges, np.nextafter(edges, np.float32(0)), np.nextafter(edges, np.float32(np.inf))])
    specials = np.concatenate([edges, -edges, np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 1e-39, 3.4e38], np.float32)])
    for batch in range(3):
        # divisor: sign * mantissa * 2^e, e across the window edges; a slice of exact special values
        e = rng.integers(-46, 47, n) if batch < 2 else rng.integers(-8, 12, n)
        b = (np.ldexp(1.0 + rng.random(n), e) * rng.choice([-1.0, 1.0], n)).astype(np.float32)
        sl = rng.random(n) < 0.02
        b[sl] = rng.choice(specials, int(sl.sum()))
        # numerators: quotient uniform a little beyond [-1, 1] ...
        u = rng.uniform(-1.0005, 1.0005, (2, n))
        with np.errstate(all="ignore"):
            a = (u * b.astype(np.float64)).astype(np.float32)
            # ... within a few ulps of +-1 (the clip edge), exactly the divisor, special values, far too small / large
            k = rng.random(n)
            near1 = (k < 0.05)
            steps = rng.integers(-3, 4, int(near1.sum()))
            a1 = b[near1] * rng.choice([-1.0, 1.0], int(near1.sum())).astype(np.float32)
            for _ in range(3):
                a1 = np.where(steps > 0, np.nextafter(a1, np.float32(np.inf)), np.where(steps < 0, np.nextafter(a1, np.float32(-np.inf)), a1))
                steps = steps - np.sign(steps)
            a[1, near1] = a1
            sp = (k >= 0.05) & (k < 0.08)
            a[1, sp] = rng.choice(specials, int(sp.sum()))
            tiny = (k >= 0.08) & (k < 0.10)
            a[1, tiny] = (b[tiny].astype(np.float64) * np.ldexp(1.0, -rng.integers(50, 140, int(tiny.sum())))).astype(np.float32)
            huge = (k >= 0.10) & (k < 0.12)
            a[0, huge] = (b[huge].astype(np.float64) * np.ldexp(1.0, rng.integers(1, 100, int(huge.sum())))).astype(np.float32)
        xyz = np.stack([a[0], a[1], b], 1)
        frac = compare(xyz, direct, f"direct operands, batch {batch}")
        assert frac > 0.2, frac                                     # most quotient triples really are accepted points
    # ---- real cameras, and the same cameras scaled onto the window's edges
    cloud = synthetic.make_cloud(1 << 23)
    proj = synthetic.make_proj(W, H)
    for pose in (0, 17, 100):
        M = camera.total_matrix(proj, synthetic.sweep_pose(pose))[0]
        assert compare(cloud, M, f"camera pose {pose}") > 0.3
        for k in (-46, -41, -40, -39, -36, 34, 38, 39, 40, 41):
            # c3 is ~1 .. 120 for visible points: these scales put the cloud's divisors below, across and above both edges
            compare(cloud[: 1 << 21], (M.astype(np.float64) * 2.0 ** k).astype(np.float32), f"camera pose {pose} scaled by 2^{k}")
    assert total > 100_000_000, total
    print(f"shared-reciprocal projection == IEEE division on {total} points")
