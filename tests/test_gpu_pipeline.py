"""FrameRenderer: graph-replayed frames with device-resident cameras, several in flight."""
import math

import numpy as np
import pytest
import torch

from robosimgs_amd import camera_ring, synthetic_scene

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)


def test_sequence_equals_direct_calls_and_overflow_is_reported():
    from robosimgs_amd import FrameRenderer, rasterization, _lib
    g = synthetic_scene(40_000, math.log(0.05), 2, 9)
    cams = camera_ring(7, 320, 192)
    t = g.to_torch(DEV, 2)
    for reorder in (None, "morton"):
        r = FrameRenderer(t, 320, 192, render_mode="RGB+ED", frames_in_flight=3,
                          sizing_camera=(cams[0].viewmat(), cams[0].K), reorder=reorder)
        got = {}
        r.render_sequence(cams, lambda i, f: got.__setitem__(i, (f["colors"].clone(), f["alphas"].clone())))
        assert sorted(got) == list(range(7))
        # the renderer's own scene (self.t: the caller's order, or its Morton-ordered copy) through the eager call
        # gives the very same bits; against the caller's order only pixels with a bit-exact depth tie may differ
        assert (r.order is None) == (reorder is None)
        if r.order is not None:
            assert torch.equal(r.t["means"], t["means"][r.order]) and sorted(r.order.tolist()) == list(range(len(g)))
        for i, cam in enumerate(cams):
            c, a, _ = rasterization(r.t["means"], r.t["quats"], r.t["scales"], r.t["opacities"], r.t["colors"],
                                    _t(cam.viewmat())[None], _t(cam.K)[None], 320, 192, sh_degree=2,
                                    render_mode="RGB+ED")
            assert torch.equal(got[i][0], c[0]) and torch.equal(got[i][1], a[0]), f"camera {i} reorder {reorder}"
            c0, a0, _ = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"],
                                      _t(cam.viewmat())[None], _t(cam.K)[None], 320, 192, sh_degree=2,
                                      render_mode="RGB+ED")
            differ = ((got[i][0][..., :3] - c0[0][..., :3]).abs().amax(-1) > 1e-6).float().mean().item()
            assert differ <= (0.0 if reorder is None else 5e-3), f"camera {i}: {differ:.2e} of the pixels differ"   # (40 k depths on ~1.3e7 fp32 values: ~60 tied pairs per camera, the few that overlap differ)
    # slot discipline
    tk = [r.submit(cams[0].viewmat(), cams[0].K) for _ in range(3)]
    with pytest.raises(RuntimeError):
        r.submit(cams[0].viewmat(), cams[0].K)
    for k in tk:
        r.fetch(k)
        with pytest.raises(RuntimeError):
            r.fetch(k)
        r.release(k)
    with pytest.raises(RuntimeError):
        r.release(tk[0])
    # a capacity that is too small is reported, not silently wrong
    small = FrameRenderer(t, 320, 192, frames_in_flight=1, isect_capacity=1000)
    with pytest.raises(_lib.MgsError, match="capacity"):
        small.render(cams[0].viewmat(), cams[0].K)


def test_render_sharded_single_rank_with_and_without_renderer():
    from robosimgs_amd import FrameRenderer
    from robosimgs_amd.distributed import render_sharded
    g = synthetic_scene(20_000, math.log(0.06), 1, 3)
    cams = camera_ring(5, 192, 128)
    t = g.to_torch(DEV, 1)
    vm, Ks = _t(np.stack([c.viewmat() for c in cams])), _t(np.stack([c.K for c in cams]))
    a, aa, mine = render_sharded(t, vm, Ks, 192, 128)
    assert list(mine) == [0, 1, 2, 3, 4] and a.shape == (5, 128, 192, 3)
    r = FrameRenderer(t, 192, 128, frames_in_flight=2, sizing_camera=(cams[0].viewmat(), cams[0].K))
    b, ba, _ = render_sharded(t, vm, Ks, 192, 128, renderer=r)
    assert torch.equal(a, b) and torch.equal(aa, ba)


def test_overflow_is_per_frame_and_packed_camera_submit():
    """The binning overwrites its status word every frame: a frame that overflows the slot's
    capacity raises, the next frame that fits renders normally.  A packed (viewmat | K) device
    tensor submits with a single copy and renders the same bits."""
    from robosimgs_amd import FrameRenderer, _lib
    g = synthetic_scene(30_000, math.log(0.05), 1, 4)
    t = g.to_torch(DEV, 1)
    near, far = camera_ring(2, 256, 160, radius=6.0)[0], camera_ring(2, 256, 160, radius=60.0)[0]
    r_far = FrameRenderer(t, 256, 160, frames_in_flight=1, sizing_camera=(far.viewmat(), far.K),
                          capacity_margin=1.05)
    ok = r_far.render(far.viewmat(), far.K)
    with pytest.raises(_lib.MgsError, match="capacity"):
        r_far.render(near.viewmat(), near.K)               # the close-up needs far more tile pairs
    again = r_far.render(far.viewmat(), far.K)
    assert torch.equal(again["colors"], ok["colors"]) and torch.equal(again["alphas"], ok["alphas"])
    packed = FrameRenderer.pack_camera(far.viewmat(), far.K, device=DEV)
    tk = r_far.submit(packed)
    f = r_far.fetch(tk)
    assert torch.equal(f["colors"], ok["colors"])
    r_far.release(tk)


@pytest.mark.parametrize("mode", ["D", "ED"])
def test_depth_only_modes_with_a_fixed_capacity_and_in_a_frame_renderer(mode):
    """"D" / "ED" of an SH-coloured scene with isect_capacity (no read-back: capturable) go through the fused
    path and must equal the depth channel of "RGB+D" / "RGB+ED"; FrameRenderer(render_mode=mode) captures."""
    from robosimgs_amd import FrameRenderer, rasterization
    g = synthetic_scene(4000, math.log(0.08), 2, 9)
    cam = camera_ring(1, 160, 96, thetas=[1.1])[0]
    t = g.to_torch(DEV, 2)
    vm = torch.from_numpy(cam.viewmat().astype(np.float32)).to(DEV)[None]
    K = torch.from_numpy(cam.K.astype(np.float32)).to(DEV)[None]
    args = (t["means"], t["quats"], t["scales"], t["opacities"], t["colors"], vm, K, 160, 96)
    full, fa, _ = rasterization(*args, sh_degree=2, render_mode="RGB+" + mode, isect_capacity=200_000)
    d, a, _ = rasterization(*args, sh_degree=2, render_mode=mode, isect_capacity=200_000)
    assert d.shape == (1, 96, 160, 1) and torch.equal(d, full[..., 3:4]) and torch.equal(a, fa)
    ref, _, _ = rasterization(*args, sh_degree=2, render_mode=mode)          # the operator path (reads n_isect back)
    torch.testing.assert_close(d, ref, rtol=1e-5, atol=1e-5)
    fr = FrameRenderer(t, 160, 96, render_mode=mode, frames_in_flight=2, isect_capacity=200_000)
    out = fr.render(cam.viewmat(), cam.K)
    assert torch.equal(out["colors"], d[0])


def test_lean_frames_drop_the_unread_arrays_and_keep_every_bit():
    """FrameRenderer's slots (lean_meta=True) pass NULL for radii / means2d / conics / feats / tiles_per_gauss /
    tile_ids at the C ABI: the image must be the one the full call gives, bit for bit, in every mode."""
    from robosimgs_amd import rasterization
    g = synthetic_scene(60_000, math.log(0.04), 3, 4)
    cam = camera_ring(3, 400, 304)[1]
    t = g.to_torch(DEV, 3)
    vm, K = _t(cam.viewmat())[None], _t(cam.K)[None]
    for mode in ("RGB", "RGB+ED", "RGB+D", "ED"):
        for aa in ("classic", "antialiased"):
            kw = dict(sh_degree=3, render_mode=mode, rasterize_mode=aa, isect_capacity=2_000_000)
            with torch.no_grad():
                c0, a0, m0 = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"], vm, K,
                                           400, 304, **kw)
                c1, a1, m1 = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"], vm, K,
                                           400, 304, lean_meta=True, **kw)
            assert torch.equal(c0, c1) and torch.equal(a0, a1), (mode, aa)
            assert "radii" in m0 and "radii" not in m1 and "means2d" not in m1 and "tiles_per_gauss" not in m1
            assert int(m1["n_isects"][0]) == int(m0["n_isects"][0]) > 0 and int(m1["isect_status"][0]) == 0
            assert "depths" not in m1 and "tile_lists" not in m1
            with pytest.raises(KeyError):
                m1["isect_ids"]
    # gradients asked for: lean is ignored, the full set of arrays is back
    p = {k: t[k].detach().clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "colors")}
    c2, a2, m2 = rasterization(p["means"], p["quats"], p["scales"], p["opacities"], p["colors"], vm, K, 400, 304,
                               sh_degree=3, render_mode="RGB+ED", isect_capacity=2_000_000, lean_meta=True)
    assert "radii" in m2
    c2.sum().backward()
    assert p["means"].grad is not None


def test_projection_refuses_null_outputs_without_their_substitutes():
    from robosimgs_amd import _lib
    L = _lib.lib()
    z = torch.zeros(64, device=DEV)
    # radii NULL but no splats / bin_info: an argument error, not a fault
    rc = L.mgs_project_color_fwd(4, z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), 0, 1, z.data_ptr(),
                                 z.data_ptr(), z.data_ptr(), 16, 16, 0.3, 0.01, 1e10, 0.0, None, None, z.data_ptr(),
                                 None, None, 3, None, None, 0, None, None, None, None)
    assert rc == -1 and b"may be NULL only" in L.mgs_last_error_string()
    # the per-axis radius rule writes two arrays: radii without radii_y is refused as well
    rc = L.mgs_project_color_fwd(4, z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), 0, 1, z.data_ptr(),
                                 z.data_ptr(), z.data_ptr(), 16, 16, 0.3, 0.01, 1e10, 0.0, z.data_ptr(), z.data_ptr(),
                                 z.data_ptr(), z.data_ptr(), None, 3, z.data_ptr(), None, 2, None, None, None, None)
    assert rc == -1 and b"radii_y" in L.mgs_last_error_string()


def test_batched_cameras_through_one_call_equal_the_per_camera_frames():
    """mgs_render_frames (rasterization(lean_meta=True) with C cameras): every frame is the frame the full per-camera
    call gives, bit for bit; backgrounds per camera; a capacity too small is reported per camera."""
    from robosimgs_amd import rasterization, check_isect_status, _lib
    g = synthetic_scene(50_000, math.log(0.04), 2, 6)
    cams = camera_ring(5, 336, 208)
    t = g.to_torch(DEV, 2)
    vm = _t(np.stack([c.viewmat() for c in cams]))
    Ks = _t(np.stack([c.K for c in cams]))
    bg = torch.rand(5, 4, device=DEV)
    for mode, aa, b in (("RGB+ED", "classic", None), ("RGB", "antialiased", bg[:, :3].contiguous()), ("RGB+D", "classic", bg)):
        kw = dict(sh_degree=2, render_mode=mode, rasterize_mode=aa, backgrounds=b, isect_capacity=1_500_000)
        with torch.no_grad():
            c0, a0, m0 = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"], vm, Ks, 336, 208, **kw)
            c1, a1, m1 = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"], vm, Ks, 336, 208,
                                       lean_meta=True, **kw)
        assert c1.shape == (5, 208, 336, c0.shape[-1]) and torch.equal(c0, c1) and torch.equal(a0, a1), (mode, aa)
        assert torch.equal(m0["n_isects"], m1["n_isects"]) and m1["isect_status"].shape == (5,)
        check_isect_status(m1)
        # tile_bounds="classic" reaches the batched call too (MGS_FRAMES_CLASSIC_BOUNDS): gsplat's counts, the same pixels
        with torch.no_grad():
            c2, a2, m2 = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"], vm, Ks, 336, 208,
                                       tile_bounds="classic", **kw)
            c3, a3, m3 = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"], vm, Ks, 336, 208,
                                       tile_bounds="classic", lean_meta=True, **kw)
        assert torch.equal(c3, c0) and torch.equal(a3, a0)
        assert torch.equal(m3["n_isects"], m2["n_isects"]) and bool((m3["n_isects"] > m1["n_isects"]).all())
    with torch.no_grad():
        _, _, m2 = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"], vm, Ks, 336, 208,
                                 sh_degree=2, isect_capacity=64, lean_meta=True)
    assert bool((m2["isect_status"] != 0).all()) and bool((m2["n_isects"] > 64).all())
    with pytest.raises(_lib.MgsError):
        check_isect_status(m2)
    L = _lib.lib()
    rc = L.mgs_render_frames(4, None, None, None, None, 0, 1, None, 1, None, None, 16, 16, 0.3, 0.01, 1e10, 0.0, 0, 7, 0, None,
                             100, None, None, None, None, None, None, 0, None, None, None, None)
    assert rc == -1 and b"channels" in L.mgs_last_error_string()


def test_batched_frames_ragged_sizes_empty_scene_and_everything_culled():
    """mgs_render_frames at a resolution that is no multiple of the tile size, with a camera that sees nothing, and with
    an empty scene: frames equal the per-camera path (or are empty) and nothing faults."""
    from robosimgs_amd import rasterization
    g = synthetic_scene(8_000, math.log(0.06), 1, 11)
    cams = camera_ring(2, 333, 207)
    t = g.to_torch(DEV, 1)
    away = np.array(cams[0].viewmat(), dtype=np.float64)
    away[2, 3] = -1e3                                             # everything behind the camera
    vm = _t(np.stack([cams[0].viewmat(), away, cams[1].viewmat()]))
    Ks = _t(np.stack([cams[0].K, cams[0].K, cams[1].K]))
    with torch.no_grad():
        c0, a0, m0 = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"], vm, Ks, 333, 207,
                                   sh_degree=1, render_mode="RGB+ED", isect_capacity=400_000)
        c1, a1, m1 = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"], vm, Ks, 333, 207,
                                   sh_degree=1, render_mode="RGB+ED", isect_capacity=400_000, lean_meta=True)
    assert torch.equal(c0, c1) and torch.equal(a0, a1)
    assert int(m1["n_isects"][1]) == 0 and float(a1[1].abs().max()) == 0.0 and float(c1[1].abs().max()) == 0.0
    assert int(m1["n_isects"][0]) > 0 and int(m1["n_isects"][2]) > 0
    e = {k: (v[:0].contiguous() if torch.is_tensor(v) else v) for k, v in t.items()}
    with torch.no_grad():
        c2, a2, m2 = rasterization(e["means"], e["quats"], e["scales"], e["opacities"], e["colors"], vm[:1], Ks[:1], 333, 207,
                                   sh_degree=1, render_mode="RGB", isect_capacity=1000, lean_meta=True)
    assert c2.shape == (1, 207, 333, 3) and float(c2.abs().max()) == 0.0 and int(m2["n_isects"][0]) == 0


def test_frame_renderer_meta_maps_back_to_the_callers_order():
    """reorder="morton" (the default) renders from the renderer's own Z-curve copy: per-Gaussian meta arrays of a
    lean_meta=False frame come in ITS order; to_caller_order() puts them in the order of the tensors it was given."""
    from robosimgs_amd import FrameRenderer, rasterization
    g = synthetic_scene(20_000, math.log(0.05), 1, 3)
    cam = camera_ring(1, 240, 160, thetas=[0.8])[0]
    t = g.to_torch(DEV, 1)
    fr = FrameRenderer(t, 240, 160, render_mode="RGB", frames_in_flight=1, isect_capacity=600_000, lean_meta=False)
    assert fr.order is not None
    f = fr.render(cam.viewmat(), cam.K)
    with torch.no_grad():
        _, _, meta = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"], _t(cam.viewmat())[None],
                                   _t(cam.K)[None], 240, 160, sh_degree=1, isect_capacity=600_000)
    for key in ("radii", "means2d", "depths", "conics"):
        got = fr.to_caller_order(f["meta"][key][0])
        assert torch.equal(got, meta[key][0]), key
    assert not torch.equal(f["meta"]["radii"][0], meta["radii"][0])          # (it really was another order)


def test_slot_streams_sit_on_distinct_hardware_queues_and_are_reused():
    """pipeline.independent_streams: the in-flight slots' streams are classed by hardware queue with spin-kernel pairs;
    the three slots of every renderer of the process get the same three streams, pairwise independent and independent
    of the consumer (current) stream -- a renderer built late in a process used to land on colliding queues and lose
    14 % (profiles/r4/00_experiments.md section 15)."""
    from robosimgs_amd import pipeline, FrameRenderer
    dev = torch.device(DEV, torch.cuda.current_device())
    a = pipeline.independent_streams(dev, 3)
    b = pipeline.independent_streams(dev, 3)
    assert [s.cuda_stream for s in a] == [s.cuda_stream for s in b]            # probed once, reused
    assert len({s.cuda_stream for s in a}) == 3
    cur = torch.cuda.current_stream(dev)
    assert all(s.cuda_stream != cur.cuda_stream for s in a)
    one = min(pipeline._spin_pair_ms(cur, cur, dev) for _ in range(5)) / 2.0
    for i, s in enumerate(a):                   # two spins on independent queues overlap: well under twice one spin (2.0 = one queue)
        assert min(pipeline._spin_pair_ms(cur, s, dev) for _ in range(5)) < 1.7 * one, i
        for r in a[i + 1:]:
            assert min(pipeline._spin_pair_ms(s, r, dev) for _ in range(5)) < 1.7 * one
    # more slots than hardware queues: the extra slots share among themselves, never with the consumer
    many = pipeline.independent_streams(dev, 6)
    assert len(many) == 6 and all(s.cuda_stream != cur.cuda_stream for s in many)
    g = synthetic_scene(2000, math.log(0.05), 1, 2)
    t = g.to_torch(DEV, 1)
    cam = camera_ring(1, 96, 64)[0]
    frs = [FrameRenderer(t, 96, 64, isect_capacity=100_000, frames_in_flight=3) for _ in range(3)]
    assert all([s["stream"].cuda_stream for s in f._slots] == [s.cuda_stream for s in a] for f in frs)
    ref = None
    for f in frs:                               # sharing streams between renderers changes no pixel
        tk = f.submit(cam.viewmat(), cam.K)
        out = f.fetch(tk)["colors"].clone()
        f.release(tk)
        assert ref is None or torch.equal(out, ref)
        ref = out


def test_a_frame_submitted_alone_takes_the_latency_schedule_and_the_same_pixels():
    """Every slot of a renderer with several frames in flight holds both raster schedules as graphs (one memory pool): a
    frame submitted while nothing else is in flight -- a synchronous render(), the first frame of a burst -- runs one wave
    per 8x8 block (the shortest launch), the frames behind it one wave per tile (the fewest instructions).  Same pixels bit
    for bit whichever graph ran; the overflow word is read over both graphs; naming a schedule pins it."""
    from robosimgs_amd import FrameRenderer, _lib
    g = synthetic_scene(60_000, math.log(0.05), 2, 11)
    cams = camera_ring(6, 352, 208)
    t = g.to_torch(DEV, 2)
    fr = FrameRenderer(t, 352, 208, render_mode="RGB+ED", frames_in_flight=3, sizing_camera=(cams[0].viewmat(), cams[0].K))
    assert all(set(s["variants"]) == {"throughput", "latency"} for s in fr._slots)
    alone = [fr.render(c.viewmat(), c.K) for c in cams]                     # one at a time: each has the GPU to itself
    assert all(s["variant"] == "latency" for s in fr._slots[:3])
    used, piped = [], {}
    tickets = [fr.submit(c.viewmat(), c.K) for c in cams[:3]]
    used = [fr._slots[tk]["variant"] for tk in tickets]
    assert used == ["latency", "throughput", "throughput"]                 # the first of the burst was alone
    for i, tk in enumerate(tickets):
        f = fr.fetch(tk)
        piped[i] = (f["colors"].clone(), f["alphas"].clone())
        fr.release(tk)
    for i in range(3):
        assert torch.equal(piped[i][0], alone[i]["colors"]) and torch.equal(piped[i][1], alone[i]["alphas"]), i
    assert fr.isect_status_max() == 0
    pinned = FrameRenderer(t, 352, 208, render_mode="RGB+ED", frames_in_flight=3, isect_capacity=fr.capacity, raster_schedule="throughput")
    assert all(set(s["variants"]) == {"throughput"} for s in pinned._slots)
    f = pinned.render(cams[4].viewmat(), cams[4].K)
    assert torch.equal(f["colors"], alone[4]["colors"])
    tiny = FrameRenderer(t, 352, 208, render_mode="RGB+ED", frames_in_flight=3, isect_capacity=1000)
    with pytest.raises(_lib.MgsError):
        tiny.render(cams[0].viewmat(), cams[0].K)
    assert tiny.isect_status_max() != 0
