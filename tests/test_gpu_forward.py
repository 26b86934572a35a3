"""GPU parity of the forward path against the oracle (oracle/gs_oracle_np.py), stage by stage
through the C ABI, then end to end.  Scene: BASELINE.json configs[0] (10k Gaussians, 256x256)
and small variants; tolerances are written next to each assertion.

Integer outputs derived from float thresholds (radii, tile rectangles) can legitimately
differ on a measure-zero set of Gaussians whose 3*sqrt(lambda) sits within rounding of an
integer; such cases are counted and bounded, and every downstream stage is checked against
the oracle run on the HIP stage's own outputs so that a flip cannot mask a real error.
"""
import math

import numpy as np
import pytest
import torch

from oracle import gs_oracle_np as O
from robosimgs_amd import camera_ring, synthetic_scene

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _scene(n=10_000, mu=0.05, deg=0, w=256, h=256, theta=0.3, seed=0):
    g = synthetic_scene(n, math.log(mu), deg, seed)
    cam = camera_ring(1, w, h, thetas=[theta])[0]
    return g, cam


def _f32(a):
    """What the GPU is given: the matrix rounded to fp32 (the oracle then computes in fp64 from it)."""
    return np.asarray(a, dtype=np.float32).astype(np.float64)


def _t(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype).to(DEV)


@pytest.fixture(scope="module")
def ops():
    from robosimgs_amd import ops as _ops
    return _ops


def test_single_hip_runtime_loaded(ops):
    """libmgs.so must bind to the HIP runtime torch loaded (one libamdhip64 in the process)."""
    from robosimgs_amd import _lib
    assert _lib.lib().mgs_version() == _lib.MGS_VERSION
    torch.zeros(1, device=DEV)
    with open("/proc/self/maps") as f:
        libs = {line.split()[-1] for line in f if "libamdhip64" in line}
    assert len(libs) == 1, f"two HIP runtimes mapped: {libs}"
    # (MGS_USE_DEBUG_LIB=1 runs the suite on libmgs_debug.so: the knob sweeps of profiles/)
    import os
    want = "libmgs_debug.so" if os.environ.get("MGS_USE_DEBUG_LIB") else "libmgs.so"
    assert any(want in line for line in open("/proc/self/maps"))


@pytest.mark.parametrize("n,mu,w,h,theta", [(10_000, 0.05, 256, 256, 0.3),
                                             (3_000, 0.2, 200, 120, 2.1),
                                             (500, 0.6, 64, 48, 4.0)])
def test_projection_matches_oracle(ops, n, mu, w, h, theta):
    g, cam = _scene(n, mu, 0, w, h, theta)
    ref = O.project(g.means, g.quats, g.scales, cam.viewmat(), cam.K, w, h)
    radii, means2d, depths, conics, comps = ops.fully_fused_projection(
        _t(g.means), None, _t(g.quats), _t(g.scales), _t(cam.viewmat())[None], _t(cam.K)[None],
        w, h, calc_compensations=True)
    radii = radii[0].cpu().numpy()
    vis_ref, vis = ref["radii"] > 0, radii > 0
    flips = int((vis_ref != vis).sum())
    assert flips <= max(1, n // 5000), f"{flips} visibility flips of {n}"
    both = vis_ref & vis
    dr = np.abs(radii[both] - ref["radii"][both])
    assert dr.max() <= 1 and (dr > 0).sum() <= max(1, n // 2000), \
        f"radius mismatches: {(dr > 0).sum()} (max {dr.max()})"
    # fp32 vs fp64: relative 2e-5 on screen-space quantities (|means2d| up to ~1e3 px)
    np.testing.assert_allclose(means2d[0].cpu().numpy()[both], ref["means2d"][both], rtol=2e-5, atol=2e-3)
    np.testing.assert_allclose(depths[0].cpu().numpy()[both], ref["depths"][both], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(conics[0].cpu().numpy()[both], ref["conics"][both], rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(comps[0].cpu().numpy()[both], ref["compensations"][both], rtol=2e-4, atol=1e-6)
    # culled rows are zeroed
    assert np.all(means2d[0].cpu().numpy()[~vis] == 0) and np.all(conics[0].cpu().numpy()[~vis] == 0)


@pytest.mark.parametrize("deg,K", [(0, 1), (1, 4), (2, 9), (3, 16), (1, 16), (2, 16), (0, 16)])
def test_spherical_harmonics_matches_oracle(ops, deg, K):
    rng = np.random.default_rng(5)
    n = 4097                                   # ragged: not a multiple of the 64-row wave slab
    dirs = rng.normal(size=(n, 3))
    coeffs = rng.normal(size=(n, K, 3))
    masks = rng.random(n) > 0.3
    Y = O.sh_basis(deg, dirs / np.linalg.norm(dirs, axis=1, keepdims=True))
    ref = np.einsum("nk,nkc->nc", Y, coeffs[:, :(deg + 1) ** 2])
    out = ops.spherical_harmonics(deg, _t(dirs), _t(coeffs)).cpu().numpy()
    np.testing.assert_allclose(out, ref, rtol=1e-4, atol=2e-5)
    out_m = ops.spherical_harmonics(deg, _t(dirs), _t(coeffs), torch.from_numpy(masks).to(DEV)).cpu().numpy()
    np.testing.assert_allclose(out_m[masks], ref[masks], rtol=1e-4, atol=2e-5)
    assert np.all(out_m[~masks] == 0)


@pytest.mark.parametrize("n,mu,w,h", [(10_000, 0.05, 256, 256), (2_000, 0.3, 200, 120),
                                       (64, 0.05, 16, 16), (5, 0.05, 256, 256)])
def test_isect_tiles_bit_exact(ops, n, mu, w, h):
    """Integer path: given the same projected inputs the sorted lists must be identical."""
    g, cam = _scene(n, mu, 0, w, h)
    radii, means2d, depths, conics, _ = ops.fully_fused_projection(
        _t(g.means), None, _t(g.quats), _t(g.scales), _t(cam.viewmat())[None], _t(cam.K)[None], w, h)
    tw, th = -(-w // 16), -(-h // 16)
    tpg, isect_ids, flatten_ids = ops.isect_tiles(means2d, radii, depths, 16, tw, th)
    offs = ops.isect_offset_encode(isect_ids, 1, tw, th)
    r_tpg, r_ids, r_flat = O.isect_tiles(means2d[0].cpu().numpy(), radii[0].cpu().numpy(),
                                         depths[0].cpu().numpy(), 16, tw, th, dtype=np.float32)
    np.testing.assert_array_equal(tpg[0].cpu().numpy(), r_tpg)
    np.testing.assert_array_equal(isect_ids.cpu().numpy(), r_ids)
    np.testing.assert_array_equal(flatten_ids.cpu().numpy(), r_flat)
    np.testing.assert_array_equal(offs.cpu().numpy(), O.isect_offsets(r_ids, 1, tw, th))


@pytest.mark.parametrize("n,span,kind", [(1500, 40.0, "ties"), (1500, 40.0, "clustered"), (30_000, 40.0, "ties"),
                                         (30_000, 40.0, "clustered"), (30_000, 40.0, "uniform"),
                                         (9_000, 6.0, "equal")])
def test_tile_depth_order_long_lists_and_depth_ties(ops, n, span, kind):
    """The per-tile depth sort (csrc/tile_sort.hip) on inputs built to leave its common path: lists far
    longer than its LDS-resident fast path (30 k entries per tile), thousands of bit-identical depths
    (order then falls to the Gaussian index), depths clustered in a few narrow groups with one far
    outlier (buckets that must be refined more than once).  Lists must be bit-identical to the stable
    sort on (tile, depth bits)."""
    rng = np.random.default_rng(n + len(kind))
    w = h = 32                                            # 2 x 2 tiles, every Gaussian covers all of them
    means2d = rng.uniform(0, 32, size=(n, 2)).astype(np.float32)
    radii = np.full(n, int(span), np.int32)
    if kind == "equal":
        depths = np.full(n, 3.25, np.float32)
    elif kind == "ties":
        depths = rng.choice(np.array([1.5, 2.0, 2.0000002, 7.25, 7.2500005], np.float32), size=n)
        depths[rng.random(n) < 0.3] = rng.uniform(1.0, 9.0, size=int((rng.random(n) < 0.3).sum()) or 1)[0]
    elif kind == "clustered":
        base = rng.choice(np.array([4.0, 4.0001, 4.0002], np.float32), size=n)
        depths = (base + rng.integers(0, 40, size=n).astype(np.float32) * np.float32(4.76837158203125e-07)).astype(np.float32)
        depths[0] = 0.011                                  # one far outlier stretches the key range
        depths[1] = 9.0e9
    else:
        depths = rng.uniform(0.5, 20.0, size=n).astype(np.float32)
    radii[rng.random(n) < 0.05] = 0                        # some culled
    tl = ops.isect_tiles_raw(_t(means2d), torch.from_numpy(radii).to(DEV), _t(depths), 2, 2, 4 * n + 16,
                             want_isect_ids=True)
    cnt = int(tl.n_isect.item())
    r_tpg, r_ids, r_flat = O.isect_tiles(means2d, radii, depths, 16, 2, 2, dtype=np.float32)
    assert cnt == len(r_flat) and int(tl.status.item()) == 0
    np.testing.assert_array_equal(tl.flatten_ids[:cnt].cpu().numpy(), r_flat)
    np.testing.assert_array_equal(tl.isect_ids[:cnt].cpu().numpy(), r_ids)
    np.testing.assert_array_equal(tl.tile_ids[:cnt].cpu().numpy(), (r_ids >> 32).astype(np.int32))


@pytest.mark.parametrize("per_list,tiles_w,kind", [(9_000, 16, "clustered"), (12_000, 16, "clustered"), (3_000, 4, "uniform"),
                                                   (12_000, 4, "clustered"), (1_700, 2, "ties")])
def test_tile_depth_order_uneven_lists_under_each_capacity_rule(ops, per_list, tiles_w, kind):
    """Two full tiles among empty ones: the AVERAGE the capacity allows picks the per-tile sort's variant (1,024-entry LDS
    list and no second launch up to 640 per tile, 1,536 entries up to 1,000, 2,048 beyond), the two lists are far longer
    than any of them -- the main kernel's own generic path under the short rule, the long-list kernel under the others
    (its LDS fast path up to 8,192 entries; beyond it the levels through global scratch with heavy buckets finished out of
    LDS).  Lists bit-identical to the stable sort whatever the variant."""
    rng = np.random.default_rng(per_list + tiles_w)
    n = 2 * per_list
    tw, th = tiles_w, 2
    means2d = np.empty((n, 2), np.float32)
    means2d[:per_list] = (8.0, 8.0)                        # tile (0, 0)
    means2d[per_list:] = (16.0 * (tw - 1) + 8.0, 24.0)     # tile (tw - 1, 1)
    means2d += rng.uniform(-2, 2, size=(n, 2)).astype(np.float32)
    radii = np.full(n, 3, np.int32)
    if kind == "clustered":
        base = rng.choice(np.array([4.0, 4.0001, 4.0002, 6.5], np.float32), size=n)
        depths = (base + rng.integers(0, 300, size=n).astype(np.float32) * np.float32(4.76837158203125e-07)).astype(np.float32)
        depths[0], depths[1] = 0.011, 9.0e9
    elif kind == "ties":
        depths = rng.choice(np.array([1.5, 2.0, 2.0000002, 7.25], np.float32), size=n)
    else:
        depths = rng.uniform(0.5, 20.0, size=n).astype(np.float32)
    cap = n + 16
    assert {(16, 9_000): cap <= 640 * tw * th, (16, 12_000): 640 * tw * th < cap <= 1000 * tw * th}.get((tiles_w, per_list), cap > 640 * tw * th)
    tl = ops.isect_tiles_raw(_t(means2d), torch.from_numpy(radii).to(DEV), _t(depths), tw, th, cap, want_isect_ids=True)
    cnt = int(tl.n_isect.item())
    r_tpg, r_ids, r_flat = O.isect_tiles(means2d, radii, depths, 16, tw, th, dtype=np.float32)
    assert cnt == len(r_flat) == n and int(tl.status.item()) == 0
    np.testing.assert_array_equal(tl.flatten_ids[:cnt].cpu().numpy(), r_flat)
    np.testing.assert_array_equal(tl.isect_ids[:cnt].cpu().numpy(), r_ids)
    np.testing.assert_array_equal(tl.tile_offsets[:tw * th].cpu().numpy(), O.isect_offsets(r_ids, 1, tw, th).reshape(-1))
    assert int(tl.tile_offsets[tw * th]) == n


@pytest.mark.parametrize("n,tw,th,kind,opts", [(30_000, 7, 6, "uniform", 0), (30_000, 7, 6, "clustered", 0), (9_500, 9, 5, "bimodal", 0),
                                               (20_000, 3, 2, "clustered", 4), (20_000, 3, 2, "outliers", 0), (70_000, 2, 1, "equal", 0)])
def test_giant_lists_are_spread_over_the_chip(ops, n, tw, th, kind, opts):
    """Lists over 8,192 entries take the cooperative path of csrc/tile_sort.hip (descriptor + pool, collect kernel, units of
    whole buckets in the long kernel).  Cases: 42 giant tiles of 28 k entries (more than the 32 descriptors and more than
    the 1 M-entry pool: the rest falls back to the long kernel's generic path), 45 giant tiles of which 32 get a descriptor,
    two depth clusters far apart (most buckets empty), the radix partition (debug knob 4: ungrouped lists), a
    sample that misses the outliers (the bucket function clamps) and 70 k identical depths (one bucket holds everything:
    the unit is the whole list, ordered by index).  Lists bit-identical to the stable sort on (tile, depth bits)."""
    from robosimgs_amd import _lib
    rng = np.random.default_rng(n + tw + len(kind))
    w, h = 16 * tw, 16 * th
    means2d = rng.uniform(0, 1, size=(n, 2)).astype(np.float32) * np.float32([w, h])
    radii = np.full(n, 16 * max(tw, th), np.int32)         # every Gaussian covers every tile
    radii[rng.random(n) < 0.05] = 0
    if kind == "uniform":
        depths = rng.uniform(0.5, 20.0, size=n).astype(np.float32)
    elif kind == "clustered":
        base = rng.choice(np.array([4.0, 4.0001, 4.0002, 6.5], np.float32), size=n)
        depths = (base + rng.integers(0, 300, size=n).astype(np.float32) * np.float32(4.76837158203125e-07)).astype(np.float32)
        depths[0], depths[1] = 0.011, 9.0e9
    elif kind == "bimodal":
        depths = np.where(rng.random(n) < 0.4, rng.normal(2.0, 0.01, n), rng.normal(9000.0, 3.0, n)).astype(np.float32)
    elif kind == "outliers":
        depths = rng.uniform(3.0, 3.5, size=n).astype(np.float32)
        depths[n - 200:] = rng.uniform(0.02, 4.0e8, size=200).astype(np.float32)      # none among a list's first entries
    else:
        depths = np.full(n, 3.25, np.float32)
    r_tpg, r_ids, r_flat = O.isect_tiles(means2d, radii, depths, 16, tw, th, dtype=np.float32)
    total = len(r_flat)
    assert total > 8192 * tw * th
    _dbg = _lib.use_debug_lib()
    lib = _dbg.__enter__()
    try:
        lib.mgs_debug_set_sort_opts(opts)
        for _ in range(2):                                  # (the second call finds the first one's descriptors in the workspace)
            tl = ops.isect_tiles_raw(_t(means2d), torch.from_numpy(radii).to(DEV), _t(depths), tw, th, total + 16, want_isect_ids=True)
            assert int(tl.n_isect.item()) == total and int(tl.status.item()) == 0
            np.testing.assert_array_equal(tl.flatten_ids[:total].cpu().numpy(), r_flat)
            np.testing.assert_array_equal(tl.isect_ids[:total].cpu().numpy(), r_ids)
            np.testing.assert_array_equal(tl.tile_ids[:total].cpu().numpy(), (r_ids >> 32).astype(np.int32))
            np.testing.assert_array_equal(tl.tile_offsets[:tw * th].cpu().numpy(), O.isect_offsets(r_ids, 1, tw, th).reshape(-1))
    finally:
        lib.mgs_debug_set_sort_opts(0)
        _dbg.__exit__(None, None, None)


def test_isect_tiles_empty_and_overflow(ops):
    from robosimgs_amd import _lib
    w = h = 64
    means2d = torch.zeros(8, 2, device=DEV)
    radii = torch.zeros(8, dtype=torch.int32, device=DEV)
    depths = torch.ones(8, device=DEV)
    tl = ops.isect_tiles_raw(means2d, radii, depths, 4, 4, 16)
    assert int(tl.n_isect.item()) == 0 and int(tl.status.item()) == 0
    assert torch.all(tl.tile_offsets == 0)
    # one Gaussian covering all 16 tiles, capacity 10 -> overflow flagged, n_isect reports 16
    means2d[0] = torch.tensor([32.0, 32.0])
    radii[0] = 100
    tl = ops.isect_tiles_raw(means2d, radii, depths, 4, 4, 10)
    assert int(tl.n_isect.item()) == 16
    assert int(tl.status.item()) & _lib.MGS_STATUS_ISECT_OVERFLOW
    tl = ops.isect_tiles_raw(means2d, radii, depths, 4, 4, 16)
    assert int(tl.status.item()) == 0
    np.testing.assert_array_equal(tl.tile_offsets.cpu().numpy(), np.arange(17))
    np.testing.assert_array_equal(tl.tile_ids.cpu().numpy(), np.arange(16))


@pytest.mark.parametrize("cap_frac,opts", [(0.45, 0), (0.8, 0), (0.45, 4)])
def test_deferred_lists_with_an_overflowed_capacity_stay_inside_their_buffers(ops, cap_frac, opts):
    """Lists over the LDS list (split into units: csrc/tile_sort.hip) while the capacity is too small for the pairs: the
    overflow is flagged, every offset lies inside the capacity, nothing is written past the buffers (canaries behind
    flatten_ids and tile_ids), and the lists that fit whole are the stable sort's."""
    from robosimgs_amd import _lib
    rng = np.random.default_rng(7)
    n, tw, th = 24_000, 3, 2
    means2d = (rng.uniform(0, 1, size=(n, 2)) * np.float32([16 * tw, 16 * th])).astype(np.float32)
    radii = np.full(n, 64, np.int32)                      # every Gaussian covers all six tiles: lists of 24 k entries
    depths = rng.uniform(0.5, 20.0, size=n).astype(np.float32)
    r_tpg, r_ids, r_flat = O.isect_tiles(means2d, radii, depths, 16, tw, th, dtype=np.float32)
    total = len(r_flat)
    cap = int(total * cap_frac)
    _dbg = _lib.use_debug_lib()
    lib = _dbg.__enter__()
    try:
        lib.mgs_debug_set_sort_opts(opts)
        for _ in range(2):
            tl = ops.isect_tiles_raw(_t(means2d), torch.from_numpy(radii).to(DEV), _t(depths), tw, th, cap)
            torch.cuda.synchronize()
            assert int(tl.n_isect.item()) == total and int(tl.status.item()) & _lib.MGS_STATUS_ISECT_OVERFLOW
            off = tl.tile_offsets.cpu().numpy()
            assert off.min() >= 0 and off.max() <= cap and np.all(np.diff(off) >= 0)
            # tiles whose lists fit whole inside the capacity hold the stable sort's entries
            ref_off = np.concatenate([O.isect_offsets(r_ids, 1, tw, th).reshape(-1), [total]])
            got = tl.flatten_ids.cpu().numpy()
            for t_ in range(tw * th):
                if ref_off[t_ + 1] <= cap and off[t_] == ref_off[t_] and off[t_ + 1] == ref_off[t_ + 1]:
                    np.testing.assert_array_equal(got[off[t_]:off[t_ + 1]], r_flat[ref_off[t_]:ref_off[t_ + 1]])
    finally:
        lib.mgs_debug_set_sort_opts(0)
        _dbg.__exit__(None, None, None)


def _raster_inputs(ops, g, cam, w, h, deg):
    t = g.to_torch(DEV, deg)
    vm, K = _t(cam.viewmat()), _t(cam.K)
    radii, means2d, depths, conics, opac, feats = ops.project_color_fwd_raw(
        t["means"], t["quats"], t["scales"], t["opacities"], deg, t["colors"], vm, K, w, h, 0.3,
        0.01, 1e10, 0.0, False, True)
    tw, th = -(-w // 16), -(-h // 16)
    cap = ops._upper_bound_isects(radii, tw, th) + 1
    tl = ops.isect_tiles_raw(means2d, radii, depths, tw, th, cap)
    return t, radii, means2d, depths, conics, feats, tl, tw, th


@pytest.mark.parametrize("n,mu,w,h,deg", [(10_000, 0.05, 256, 256, 0), (4_000, 0.15, 200, 120, 3),
                                           (20_000, 0.08, 96, 80, 1)])
def test_rasterize_matches_oracle(ops, n, mu, w, h, deg):
    """Blend stage on identical inputs: |diff| <= 1e-4 abs (north-star tolerance) on RGB,
    depth-sum and alpha at EVERY pixel, except those where the fp64 blend itself took a decision
    (alpha >= 1/255, T' <= 1e-4, sigma >= 0) within O.EPS_STAGE (2e-5 relative) of flipping; a pixel
    over tolerance that is not one of those fails the test (O.check_frame)."""
    g, cam = _scene(n, mu, deg, w, h)
    t, radii, means2d, depths, conics, feats, tl, tw, th = _raster_inputs(ops, g, cam, w, h, deg)
    bg = torch.tensor([0.1, 0.2, 0.3, 0.0], device=DEV)
    render, alphas, last = ops.rasterize_fwd_raw(means2d, conics, feats, t["opacities"], bg, w, h,
                                                 tw, th, tl.tile_offsets, tl.flatten_ids)
    n_isect = int(tl.n_isect.item())
    ref_img, ref_alpha, ref_last, stats = O.rasterize(
        means2d.cpu().numpy(), conics.cpu().numpy(), feats.cpu().numpy(),
        t["opacities"].cpu().numpy(), tl.flatten_ids[:n_isect].cpu().numpy(),
        tl.tile_offsets[:-1].cpu().numpy().reshape(th, tw), w, h, 16,
        background=bg.cpu().numpy(), margins=True, flip_eps=O.EPS_STAGE)
    st = O.check_frame(render.cpu().numpy(), alphas.cpu().numpy(), ref_img, ref_alpha, stats["margins"],
                       O.EPS_STAGE, what=f"blend stage n={n}", flip_weight=stats["flip_weight"],
                       feat_max=np.maximum(feats.abs().amax(0).cpu().numpy(), bg.abs().cpu().numpy()),
                       require_flip_bound=True)
    print(f"\nblend stage n={n} {w}x{h}: {st}")
    same_last = (last.cpu().numpy() == ref_last).mean()
    assert same_last >= 0.999, f"last_ids agree on {same_last:.5f} of pixels"
    assert stats["contribs"] > 0


@pytest.mark.parametrize("mode", ["RGB", "RGB+ED", "D", "RGB+D", "ED"])
def test_rasterization_end_to_end(mode):
    """configs[0]: 10k Gaussians, SH degree 0, 256x256 -- whole path vs whole oracle."""
    from robosimgs_amd import rasterization
    g, cam = _scene(10_000, 0.05, 0, 256, 256)
    t = g.to_torch(DEV, 0)
    colors, alphas, meta = rasterization(t["means"], t["quats"], t["scales"], t["opacities"],
                                         t["colors"], _t(cam.viewmat())[None], _t(cam.K)[None],
                                         256, 256, sh_degree=0, render_mode=mode, tile_bounds="classic")
    ref, ref_alpha, rmeta = O.render(g.means, g.quats, g.scales, g.opacities, g.sh_coeffs,
                                     _f32(cam.viewmat()), _f32(cam.K), 256, 256, sh_degree=0,
                                     render_mode=mode, margins=True, flip_eps=O.EPS_PATH)
    assert colors.shape == (1,) + ref.shape
    assert int(meta["radii"].gt(0).sum()) == rmeta["n_vis"] == 9849
    if "n_isects" in meta:
        assert int(meta["n_isects"][0]) == rmeta["n_isect"] == 37024
    # whole path vs whole fp64 oracle (fed the fp32-rounded camera the GPU gets): zero pixels over
    # 1e-4 that no threshold / knife edge within O.EPS_PATH explains
    st = O.check_frame(colors[0].cpu().numpy(), alphas[0].cpu().numpy(), ref, ref_alpha, rmeta["margins"],
                       O.EPS_PATH, rmeta["edge_mask"], expected_depth="E" in mode, what=f"configs[0] {mode}",
                       flip_weight=rmeta["flip_weight"], feat_max=rmeta["feat_max"], require_flip_bound=True)
    print(f"\nconfigs[0] {mode}: {st}")
    # default (tight) tile bounds: shorter lists, the same image bit for bit
    c2, a2, meta2 = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"],
                                  _t(cam.viewmat())[None], _t(cam.K)[None], 256, 256, sh_degree=0,
                                  render_mode=mode)
    assert torch.equal(c2, colors) and torch.equal(a2, alphas)
    if "n_isects" in meta2:
        assert int(meta2["n_isects"][0]) < int(meta["n_isects"][0])
        assert bool((meta2["tiles_per_gauss"] <= meta["tiles_per_gauss"]).all())


def test_rasterization_multi_camera_and_capacity():
    from robosimgs_amd import rasterization, check_isect_status, _lib
    g = synthetic_scene(5000, math.log(0.08), 2, 3)
    cams = camera_ring(3, 160, 96)
    t = g.to_torch(DEV, 2)
    vm = _t(np.stack([c.viewmat() for c in cams]))
    Ks = _t(np.stack([c.K for c in cams]))
    colors, alphas, meta = rasterization(t["means"], t["quats"], t["scales"], t["opacities"],
                                         t["colors"], vm, Ks, 160, 96, sh_degree=2,
                                         isect_capacity=200_000)
    check_isect_status(meta)
    for c, cam in enumerate(cams):
        ref, ref_alpha, rm = O.render(g.means, g.quats, g.scales, g.opacities, g.sh_coeffs,
                                      _f32(cam.viewmat()), _f32(cam.K), 160, 96, sh_degree=2, margins=True,
                                      flip_eps=O.EPS_PATH)
        O.check_frame(colors[c].cpu().numpy(), alphas[c].cpu().numpy(), ref, ref_alpha, rm["margins"],
                      O.EPS_PATH, rm["edge_mask"], what=f"camera {c}", flip_weight=rm["flip_weight"],
                      feat_max=rm["feat_max"], require_flip_bound=True)
    _, _, meta = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"],
                               vm, Ks, 160, 96, sh_degree=2, isect_capacity=100)
    with pytest.raises(_lib.MgsError):
        check_isect_status(meta)


def test_cpu_tensors_are_rejected():
    from robosimgs_amd import rasterization, _lib
    z = torch.zeros(4, 3)
    with pytest.raises(_lib.MgsError):
        rasterization(z, torch.zeros(4, 4), z, torch.zeros(4), z, torch.eye(4)[None],
                      torch.eye(3)[None], 16, 16)


def test_empty_and_fully_culled_scenes():
    """Edge cases: no Gaussians at all, and Gaussians that are all behind the camera."""
    from robosimgs_amd import rasterization
    vm, K = torch.eye(4, device=DEV)[None], torch.tensor([[[50.0, 0, 20], [0, 50.0, 12], [0, 0, 1]]], device=DEV)
    bg = torch.tensor([[0.25, 0.5, 0.75]], device=DEV)
    z3, z4 = torch.zeros(0, 3, device=DEV), torch.zeros(0, 4, device=DEV)
    c, a, meta = rasterization(z3, z4, z3, torch.zeros(0, device=DEV), torch.zeros(0, 1, 3, device=DEV),
                               vm, K, 40, 24, sh_degree=0, backgrounds=bg)
    assert c.shape == (1, 24, 40, 3) and float(a.abs().max()) == 0.0
    assert torch.allclose(c[0], bg[0].expand(24, 40, 3))
    n = 100
    means = torch.randn(n, 3, device=DEV) - torch.tensor([0.0, 0.0, 10.0], device=DEV)   # z < 0: behind
    quats = torch.randn(n, 4, device=DEV)
    c, a, meta = rasterization(means, quats, torch.full((n, 3), 0.1, device=DEV), torch.full((n,), 0.5, device=DEV),
                               torch.rand(n, 4, 3, device=DEV), vm, K, 40, 24, sh_degree=1, backgrounds=bg)
    assert int(meta["radii"].sum()) == 0 and int(meta["n_isects"][0]) == 0
    assert torch.allclose(c[0], bg[0].expand(24, 40, 3)) and float(a.abs().max()) == 0.0


def test_single_huge_gaussian_covers_every_tile():
    """A Gaussian whose rectangle is the whole tile grid (the per-Gaussian emit loop's extreme)."""
    from robosimgs_amd import rasterization
    vm = torch.eye(4, device=DEV)[None]
    K = torch.tensor([[[100.0, 0, 128], [0, 100.0, 72], [0, 0, 1]]], device=DEV)
    means = torch.tensor([[0.0, 0.0, 2.0], [0.3, 0.1, 1.5]], device=DEV)
    quats = torch.tensor([[1.0, 0, 0, 0], [1.0, 0, 0, 0]], device=DEV)
    scales = torch.tensor([[3.0, 3.0, 3.0], [0.05, 0.05, 0.05]], device=DEV)
    opac = torch.tensor([0.6, 0.9], device=DEV)
    cols = torch.tensor([[1.0, 0.5, 0.25], [0.0, 1.0, 0.0]], device=DEV)
    c, a, meta = rasterization(means, quats, scales, opac, cols, vm, K, 256, 144)
    assert int(meta["tiles_per_gauss"][0, 0]) == 16 * 9
    ref, ra, _ = O.render(means.cpu().numpy(), quats.cpu().numpy(), scales.cpu().numpy(), opac.cpu().numpy(),
                          cols.cpu().numpy(), np.eye(4), K[0].cpu().numpy(), 256, 144)
    np.testing.assert_allclose(c[0].cpu().numpy(), ref, atol=1e-4)
    np.testing.assert_allclose(a[0].cpu().numpy(), ra, atol=1e-4)


def test_degenerate_inputs_do_not_crash_or_leak():
    """NaN / Inf / absurd parameters: such Gaussians are culled or inert, the rest of the frame is
    identical to the frame without them, and nothing hangs or writes out of bounds."""
    from robosimgs_amd import rasterization, check_isect_status
    g, cam = _scene(4000, 0.1, 1, 128, 96)
    t = g.to_torch(DEV, 1)
    vm, K = _t(cam.viewmat())[None], _t(cam.K)[None]
    base, base_a, _ = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"],
                                    vm, K, 128, 96, sh_degree=1)
    bad = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in t.items()}
    n_bad = 6
    pad = lambda x, rows: torch.cat([x, rows.to(x)], 0)
    nan, inf = float("nan"), float("inf")
    bad["means"] = pad(t["means"], torch.tensor([[nan, 0, 0], [0, 0, 0], [0, 0, 0], [inf, 0, 0], [0, 0, 0], [0, 0, 0]]))
    bad["quats"] = pad(t["quats"], torch.tensor([[1.0, 0, 0, 0], [nan, 0, 0, 0], [0, 0, 0, 0], [1, 0, 0, 0], [1, 0, 0, 0], [1, 0, 0, 0]]))
    bad["scales"] = pad(t["scales"], torch.tensor([[.1, .1, .1], [.1, .1, .1], [.1, .1, .1], [.1, .1, .1], [nan, .1, .1], [.1, .1, .1]]))
    bad["opacities"] = pad(t["opacities"], torch.tensor([.5, .5, .5, .5, .5, nan]))
    bad["colors"] = pad(t["colors"], torch.zeros(n_bad, 4, 3))
    out, out_a, meta = rasterization(bad["means"], bad["quats"], bad["scales"], bad["opacities"], bad["colors"],
                                     vm, K, 128, 96, sh_degree=1, isect_capacity=200_000)
    check_isect_status(meta)
    assert torch.isfinite(out).all() and torch.isfinite(out_a).all()
    assert torch.equal(out, base) and torch.equal(out_a, base_a)
    # an absurdly large Gaussian covers every tile; a tiny capacity is flagged, never overrun
    huge = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in t.items()}
    huge["scales"][:50] = 1e6
    _, _, meta = rasterization(huge["means"], huge["quats"], huge["scales"], huge["opacities"], huge["colors"],
                               vm, K, 128, 96, sh_degree=1, isect_capacity=5_000)
    assert int(meta["isect_status"][0]) != 0
    torch.cuda.synchronize()


@pytest.mark.parametrize("deg,w,h", [(0, 256, 256), (3, 200, 120), (1, 97, 83)])
def test_raster_schedules_give_identical_bits(ops, deg, w, h):
    """MGS_RASTER_LATENCY (one wave per 8x8 block) and the default one-wave-per-tile kernel must agree
    on every bit of render, alpha and last_ids, with and without background / expected depth."""
    g, cam = _scene(15_000, 0.07, deg, w, h, seed=5)
    t, radii, means2d, depths, conics, feats, tl, tw, th = _raster_inputs(ops, g, cam, w, h, deg)
    bg = torch.tensor([0.1, 0.2, 0.3, 0.5], device=DEV)
    for track in (True, False):
        for kw in (dict(), dict(expected_last=True)):
            a = ops.rasterize_fwd_raw(means2d, conics, feats, t["opacities"], bg, w, h, tw, th, tl.tile_offsets,
                                      tl.flatten_ids, track_last=track, latency=False, **kw)
            b = ops.rasterize_fwd_raw(means2d, conics, feats, t["opacities"], bg, w, h, tw, th, tl.tile_offsets,
                                      tl.flatten_ids, track_last=track, latency=True, **kw)
            for x, y, name in zip(a, b, ("render", "alphas", "last_ids")):
                if x is not None:
                    assert torch.equal(x, y), f"{name} differs (track_last={track}, {kw})"
    assert float(a[1].max()) > 0.5


@pytest.mark.parametrize("aniso", [1.0, 30.0, 300.0])
def test_quadrant_cull_never_changes_a_pixel(ops, aniso):
    """The raster forward drops (Gaussian, quadrant) pairs that cannot reach alpha >= 1/255.  With
    the cull disabled every pair is evaluated; render, alpha and last_ids must be bit-identical --
    also for needle-like Gaussians (axis ratio up to 300) and sub-threshold opacities, where the
    quadratic form's rounding is worst."""
    from robosimgs_amd import _lib
    g, cam = _scene(20_000, 0.06, 1, 208, 144, seed=13)
    rng = np.random.default_rng(1)
    g.log_scales[:, 0] += math.log(aniso)                       # stretch one axis
    g.opacity_logits[::7] = rng.uniform(-6.5, -5.0, size=len(g.opacity_logits[::7]))   # around 1/255
    t, radii, means2d, depths, conics, feats, tl, tw, th = _raster_inputs(ops, g, cam, 208, 144, 1)
    for latency in (False, True):          # both raster kernels carry the cull
        # (the knob is process-global state: it exists in libmgs_debug.so only, the shipped library has none)
        with _lib.use_debug_lib() as dbg:
            try:
                dbg.mgs_debug_set_raster_cull(1)
                a = ops.rasterize_fwd_raw(means2d, conics, feats, t["opacities"], None, 208, 144, tw, th,
                                          tl.tile_offsets, tl.flatten_ids, latency=latency)
                dbg.mgs_debug_set_raster_cull(0)
                b = ops.rasterize_fwd_raw(means2d, conics, feats, t["opacities"], None, 208, 144, tw, th,
                                          tl.tile_offsets, tl.flatten_ids, latency=latency)
            finally:
                dbg.mgs_debug_set_raster_cull(1)
        # the shipped library renders the same bits as the debug build with the cull on
        c = ops.rasterize_fwd_raw(means2d, conics, feats, t["opacities"], None, 208, 144, tw, th,
                                  tl.tile_offsets, tl.flatten_ids, latency=latency)
        assert all(torch.equal(x, y) for x, y in zip(a, c))
        assert float(a[1].max()) > 0.5
        for x, y, name in zip(a, b, ("render", "alphas", "last_ids")):
            assert torch.equal(x, y), f"{name}: cull changed {int((x != y).sum())} values (latency={latency})"


def test_equal_depths_keep_gaussian_index_order():
    """Collisions of the sort key: Gaussians at exactly the same camera depth must be blended in
    Gaussian-index order (stable sort), in the lists and in the image."""
    from robosimgs_amd import rasterization
    n = 40
    rng = np.random.default_rng(2)
    vm = np.eye(4)
    K = np.array([[120.0, 0, 48], [0, 120.0, 40], [0, 0, 1]])
    means = np.column_stack([rng.uniform(-0.5, 0.5, n), rng.uniform(-0.4, 0.4, n), np.full(n, 2.0)])
    means[n // 2:, 2] = 3.0                                # two depth planes, 20 exact ties each
    quats = np.tile([1.0, 0, 0, 0], (n, 1))
    scales = np.full((n, 3), 0.15)
    opac = rng.uniform(0.3, 0.95, n)
    cols = rng.uniform(0, 1, (n, 3))
    perm = rng.permutation(n)                              # index order unrelated to position
    means, opac, cols = means[perm], opac[perm], cols[perm]
    c, a, meta = rasterization(_t(means), _t(quats), _t(scales), _t(opac), _t(cols), _t(vm)[None],
                               _t(K)[None], 96, 80, tile_bounds="classic")
    ref, ra, rmeta = O.render(means, quats, scales, opac, cols, vm, K, 96, 80)
    np.testing.assert_allclose(c[0].cpu().numpy(), ref, atol=1e-4)
    np.testing.assert_allclose(a[0].cpu().numpy(), ra, atol=1e-4)
    ids = meta["flatten_ids"].cpu().numpy()
    off = meta["isect_offsets"].cpu().numpy().reshape(-1)
    depth = means[:, 2]
    for t0, t1 in zip(off, list(off[1:]) + [len(ids)]):
        tile = ids[t0:t1]
        key = depth[tile] * 1000 + tile                    # depth-major, then Gaussian index
        assert np.all(np.diff(key) > 0)
    assert len(ids) == rmeta["n_isect"]


@pytest.mark.parametrize("seed", range(12))
def test_isect_tiles_random_sweep_bit_exact(ops, seed):
    """Binning on synthetic screen-space inputs (not derived from a projection): odd counts around
    the sort tile sizes, many zero radii, rectangles straddling every image edge, heavy depth ties,
    tile grids from 1x1 to 1023 wide.  Lists, keys, offsets and counts must equal the stable-sort
    formulation bit for bit."""
    rng = np.random.default_rng(100 + seed)
    n = int(rng.choice([1, 63, 64, 65, 255, 257, 1023, 1025, 4097, 30_011]))
    tw = int(rng.choice([1, 2, 7, 16, 61, 200, 1023]))
    th = int(rng.choice([1, 3, 9, 33]))
    w, h = tw * 16 - int(rng.integers(0, 16)), th * 16 - int(rng.integers(0, 16))
    means2d = np.column_stack([rng.uniform(-40, w + 40, n), rng.uniform(-40, h + 40, n)]).astype(np.float32)
    means2d[:: 7] = np.round(means2d[:: 7] / 16) * 16                       # exactly on tile edges
    radii = rng.choice([0, 0, 1, 3, 8, 17, 40, 300], size=n).astype(np.int32)
    if n * 50 > 3_000_000:
        radii = np.minimum(radii, 17)
    depths = rng.uniform(0.5, 30.0, n).astype(np.float32)
    depths[rng.integers(0, n, n // 2)] = np.float32(7.25)                   # half the scene at one depth
    tpg, isect_ids, flatten_ids = ops.isect_tiles(_t(means2d)[None], torch.from_numpy(radii).to(DEV)[None],
                                                  _t(depths)[None], 16, tw, th)
    r_tpg, r_ids, r_flat = O.isect_tiles(means2d, radii, depths, 16, tw, th, dtype=np.float32)
    np.testing.assert_array_equal(tpg[0].cpu().numpy(), r_tpg)
    np.testing.assert_array_equal(isect_ids.cpu().numpy(), r_ids)
    np.testing.assert_array_equal(flatten_ids.cpu().numpy(), r_flat)
    offs = ops.isect_offset_encode(isect_ids, 1, tw, th)
    np.testing.assert_array_equal(offs.cpu().numpy(), O.isect_offsets(r_ids, 1, tw, th))
    # the device-resident variant (count never read back) agrees, with exactly enough capacity
    total = int(r_tpg.sum())
    tl = ops.isect_tiles_raw(_t(means2d), torch.from_numpy(radii).to(DEV), _t(depths), tw, th, max(total, 1),
                             want_pair_info=True)
    assert int(tl.n_isect.item()) == total and int(tl.status.item()) == 0
    np.testing.assert_array_equal(tl.flatten_ids[:total].cpu().numpy(), r_flat)
    np.testing.assert_array_equal(tl.tile_offsets[:-1].cpu().numpy(), O.isect_offsets(r_ids, 1, tw, th).reshape(-1))
    assert int(tl.tile_offsets[-1]) == total


@pytest.mark.parametrize("seed", range(6))
def test_partition_variants_give_identical_lists(ops, seed):
    """The binning's two ways of bringing a tile's entries together -- the counting sort on tile groups with
    LDS counters (csrc/binning.hip direct_*_kernel, groups of 1 / 2 / 4 / 8 / 16 tiles) and the stable radix sort
    on the tile bits -- must hand over identical lists, offsets, tile ids, counts and record slots; with too
    small a capacity each must flag the overflow and keep every offset inside the buffers."""
    from robosimgs_amd import _lib
    rng = np.random.default_rng(500 + seed)
    n = int(rng.choice([300, 4097, 20_000, 70_001]))
    tw, th = int(rng.choice([3, 13, 30, 121])), int(rng.choice([1, 5, 18, 67]))
    w, h = tw * 16 - int(rng.integers(0, 16)), th * 16 - int(rng.integers(0, 16))
    means2d = np.column_stack([rng.uniform(-30, w + 30, n), rng.uniform(-30, h + 30, n)]).astype(np.float32)
    radii = rng.choice([0, 1, 3, 8, 17, 40, 90], size=n, p=[.2, .2, .2, .2, .1, .07, .03]).astype(np.int32)
    radii[:3] = 4000                                                        # whole-grid rectangles (walked by the wave)
    depths = rng.uniform(0.5, 30.0, n).astype(np.float32)
    depths[rng.integers(0, n, n // 3)] = np.float32(2.5)
    args = (_t(means2d), torch.from_numpy(radii).to(DEV), _t(depths), tw, th)
    r_tpg, r_ids, r_flat = O.isect_tiles(means2d, radii, depths, 16, tw, th, dtype=np.float32)
    total = len(r_flat)
    _dbg = _lib.use_debug_lib()
    lib = _dbg.__enter__()          # the partition knob lives in libmgs_debug.so only
    try:
        for opts in (4, 0x08, 0x18, 0x28, 0x38, 0x48, 0):
            lib.mgs_debug_set_sort_opts(opts)
            tl = ops.isect_tiles_raw(*args, total + 5, want_pair_info=True, want_isect_ids=True, want_tiles_per_gauss=True)
            assert int(tl.n_isect.item()) == total and int(tl.status.item()) == 0, hex(opts)
            np.testing.assert_array_equal(tl.flatten_ids[:total].cpu().numpy(), r_flat, err_msg=hex(opts))
            np.testing.assert_array_equal(tl.isect_ids[:total].cpu().numpy(), r_ids, err_msg=hex(opts))
            np.testing.assert_array_equal(tl.tile_ids[:total].cpu().numpy(), (r_ids >> 32).astype(np.int32), err_msg=hex(opts))
            np.testing.assert_array_equal(tl.tiles_per_gauss.cpu().numpy(), r_tpg, err_msg=hex(opts))
            np.testing.assert_array_equal(tl.tile_offsets[:-1].cpu().numpy(), O.isect_offsets(r_ids, 1, tw, th).reshape(-1), err_msg=hex(opts))
            assert int(tl.tile_offsets[-1]) == total
            if opts == 4:
                slots = tl.pair_info.cpu().numpy().copy()
            else:
                np.testing.assert_array_equal(tl.pair_info.cpu().numpy(), slots, err_msg=hex(opts))
            # overflow: a third of the room
            cap = max(total // 3, 1)
            tl = ops.isect_tiles_raw(*args, cap)
            assert int(tl.n_isect.item()) == total and int(tl.status.item()) & _lib.MGS_STATUS_ISECT_OVERFLOW, hex(opts)
            off = tl.tile_offsets.cpu().numpy()
            assert off.min() >= 0 and off.max() <= cap and np.all(np.diff(off) >= 0), hex(opts)
    finally:
        lib.mgs_debug_set_sort_opts(0)
        _dbg.__exit__(None, None, None)
    # ... and the shipped library (no knob) gives the same lists
    tl = ops.isect_tiles_raw(*args, total + 5, want_pair_info=True, want_isect_ids=True, want_tiles_per_gauss=True)
    np.testing.assert_array_equal(tl.flatten_ids[:total].cpu().numpy(), r_flat)
    np.testing.assert_array_equal(tl.pair_info.cpu().numpy(), slots)


@pytest.mark.parametrize("w,h", [(200, 120), (96, 80), (1000, 40)])
def test_tile_group_order_is_a_schedule_not_a_result(ops, w, h):
    """mgs_isect_tiles' tile_group_order: a permutation of the groups of four tiles, longest total list first
    (1024 length classes), identical in meaning on the direct and the radix partition; the raster kernels
    produce the same bits with it, without it, and on either schedule (also where the tile count is no multiple
    of four)."""
    from robosimgs_amd import _lib
    g, cam = _scene(6000, 0.1, 1, w, h)
    _dbg = _lib.use_debug_lib()
    lib = _dbg.__enter__()
    try:
        frames = []
        for opts in (0, 4):
            lib.mgs_debug_set_sort_opts(opts)
            t, radii, means2d, depths, conics, feats, tl, tw, th = _raster_inputs(ops, g, cam, w, h, 1)
            n_groups = (tw * th + 3) // 4
            order = tl.group_order.cpu().numpy()
            assert sorted(order.tolist()) == list(range(n_groups))
            off = tl.tile_offsets.cpu().numpy().astype(np.int64)
            totals = np.array([off[min(4 * k + 4, tw * th)] - off[4 * k] for k in range(n_groups)])
            cls = np.minimum(1023, (totals[order].astype(np.float32) * np.float32(1023.0 / max(totals.max(), 1))).astype(np.int64))
            assert np.all(np.diff(cls) <= 0), "length classes must fall along the order"
            for go in (None, tl.group_order):
                for lat in (False, True):
                    for track in (False, True):
                        frames.append(ops.rasterize_fwd_raw(means2d, conics, feats, t["opacities"], None, w, h, tw, th,
                                                            tl.tile_offsets, tl.flatten_ids, track_last=track,
                                                            latency=lat, group_order=go))
        for f in frames[1:]:
            assert torch.equal(f[0], frames[0][0]) and torch.equal(f[1], frames[0][1])
            if f[2] is not None:
                assert torch.equal(f[2], frames[1][2])
    finally:
        lib.mgs_debug_set_sort_opts(0)
        _dbg.__exit__(None, None, None)


@pytest.mark.parametrize("tw,th", [(1023, 58), (1023, 59), (1000, 131)])
def test_binning_at_the_edge_of_the_lds_histogram(ops, tw, th):
    """Tile grids around the largest the direct partition takes (15,000 groups of four tiles: a 60 KB LDS
    histogram) and beyond it (radix partition): lists equal the stable-sort formulation bit for bit."""
    rng = np.random.default_rng(tw + th)
    n = 5000
    w, h = tw * 16, th * 16
    means2d = np.column_stack([rng.uniform(0, w, n), rng.uniform(0, h, n)]).astype(np.float32)
    radii = rng.choice([0, 2, 9, 30, 200], size=n).astype(np.int32)
    depths = rng.uniform(0.5, 30.0, n).astype(np.float32)
    r_tpg, r_ids, r_flat = O.isect_tiles(means2d, radii, depths, 16, tw, th, dtype=np.float32)
    total = len(r_flat)
    tl = ops.isect_tiles_raw(_t(means2d), torch.from_numpy(radii).to(DEV), _t(depths), tw, th, total + 1, want_isect_ids=True)
    assert int(tl.n_isect.item()) == total and int(tl.status.item()) == 0
    np.testing.assert_array_equal(tl.flatten_ids[:total].cpu().numpy(), r_flat)
    np.testing.assert_array_equal(tl.isect_ids[:total].cpu().numpy(), r_ids)
    np.testing.assert_array_equal(tl.tile_offsets[:-1].cpu().numpy(), O.isect_offsets(r_ids, 1, tw, th).reshape(-1))
    order = tl.group_order.cpu().numpy()
    assert sorted(order.tolist()) == list(range((tw * th + 3) // 4))
