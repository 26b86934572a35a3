"""mgs_frame_to_dataset on the MI355X against the arrays the reference's readers were given
(tests/golden/dataset_reference.npz), and the whole output path on a rendered frame."""
import math
import os

import numpy as np
import pytest
import torch

from robosimgs_amd import camera as C

pytestmark = pytest.mark.gpu
DEV = "cuda"
HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "dataset_reference.npz"))


@pytest.mark.parametrize("i", range(int(G["n_frames"])))
def test_kernel_writes_the_arrays_the_reference_read(i, tmp_path):
    from robosimgs_amd.dataset import DatasetWriter, frame_to_dataset
    colors = torch.from_numpy(G[f"colors_{i}"]).to(DEV)
    alpha = torch.from_numpy(G[f"alpha_{i}"]).to(DEV)
    for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        rgba, dist = frame_to_dataset(colors, alpha[..., None], G[f"K_{i}"], background=G["background"].tolist(),
                                      distance_dtype=dt)
        assert rgba.dtype == torch.uint8 and dist.dtype == dt and dist.shape == (*alpha.shape, 1)
        assert np.array_equal(rgba.cpu().numpy(), G[f"rgba_{i}"])                         # bit for bit
        assert np.array_equal(dist.cpu().numpy(), G[f"distance_{tag}_{i}"])               # bit for bit
        img, dep = DatasetWriter(str(tmp_path / tag)).write(i, rgba, dist)
        assert open(img, "rb").read() == G[f"png_bytes_{i}"].tobytes()                    # the very files
        assert open(dep, "rb").read() == G[f"npygz_bytes_{tag}_{i}"].tobytes()            # the reference opened
    # the light gather payload: the same distance rounded ONCE to IEEE half (what np.float16 of the f64 array gives)
    rgba16, d16 = frame_to_dataset(colors, alpha[..., None], G[f"K_{i}"], background=G["background"].tolist(),
                                   distance_dtype=torch.float16)
    assert d16.dtype == torch.float16 and np.array_equal(rgba16.cpu().numpy(), G[f"rgba_{i}"])
    assert np.array_equal(d16.cpu().numpy(), G[f"distance_f64_{i}"].astype(np.float16))
    rgba_only, none = frame_to_dataset(colors[..., :3].contiguous(), alpha)              # RGB frame: image only
    assert none is None and np.array_equal(rgba_only.cpu().numpy()[..., 3], G[f"rgba_{i}"][..., 3])
    with pytest.raises(ValueError):
        frame_to_dataset(colors[..., :3].contiguous(), alpha, G[f"K_{i}"])


def test_rendered_frame_through_the_writer_and_back(tmp_path):
    """Render -> mgs_frame_to_dataset -> files -> this repo's distance_to_depth (reference-pinned in
    test_camera_golden.py): the z-depth that comes back is the renderer's ED channel to the last bit, the
    stored mask is alpha > 0."""
    from robosimgs_amd import camera_ring, rasterization, synthetic_scene
    from robosimgs_amd.dataset import DatasetWriter, frame_to_dataset, read_dataset_frame
    g = synthetic_scene(1_500, math.log(0.02), 3, 5)          # sparse: some pixels see no splat at all
    cams = camera_ring(3, 320, 208)
    t = g.to_torch(DEV, 3)
    wr = DatasetWriter(str(tmp_path))
    for i, cam in enumerate(cams):
        vm = torch.from_numpy(cam.viewmat().astype(np.float32)).to(DEV)[None]
        K = torch.from_numpy(cam.K.astype(np.float32)).to(DEV)[None]
        with torch.no_grad():
            colors, alphas, _ = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"], vm, K,
                                              320, 208, sh_degree=3, render_mode="RGB+ED")
        rgba, dist = frame_to_dataset(colors[0], alphas[0], cam.K, distance_dtype=torch.float64)
        img, dep = wr.write(i, rgba, dist)
        back_rgba, back_dist = read_dataset_frame(img, dep)
        a = alphas[0, ..., 0].cpu().numpy()
        assert 0.01 < (a > 0).mean() < 1.0
        assert np.array_equal(back_rgba[..., 3] > 0, a > 0)
        z = C.distance_to_depth(back_dist, cam.K).astype(np.float32)
        assert np.array_equal(z, colors[0, ..., 3].cpu().numpy())
        rgba32, dist32 = frame_to_dataset(colors[0], alphas[0], cam.K)                   # fp32 files: one ulp
        z32 = C.distance_to_depth(dist32[..., 0].cpu().numpy().astype(np.float64), cam.K)
        ed = colors[0, ..., 3].cpu().numpy()
        assert np.all(np.abs(z32 - ed.astype(np.float64)) <= np.spacing(np.abs(ed)))
    assert sorted(os.listdir(wr.image_dir)) == [f"frame_{i:05d}.png" for i in range(3)]
    assert sorted(os.listdir(wr.depth_dir)) == [f"frame_{i:05d}.npy.gz" for i in range(3)]


@pytest.mark.parametrize("dt,sched,bg", [(torch.float16, "throughput", False), (torch.float32, "latency", True),
                                         (torch.float64, "throughput", True), (None, "latency", False)])
def test_dataset_frame_straight_out_of_the_raster_equals_the_conversion_kernel(dt, sched, bg):
    """mgs_rasterize_fwd's ds_* outputs (rasterization(dataset_out=...), through mgs_render_frames): RGBA8 + ray distance
    written by the raster's epilogue are, byte for byte, what mgs_frame_to_dataset makes of the float frame -- both raster
    schedules, every distance type, with a background, with and without the float frame beside them, several cameras."""
    from robosimgs_amd import camera_ring, rasterization, synthetic_scene
    from robosimgs_amd.dataset import frame_to_dataset
    W, H, C = 200, 136, 3                       # ragged: not multiples of the 16-pixel tile
    g = synthetic_scene(20_000, math.log(0.05), 2, 4)
    cams = camera_ring(C, W, H)
    t = g.to_torch(DEV, 2)
    vm = torch.from_numpy(np.stack([c.viewmat() for c in cams]).astype(np.float32)).to(DEV)
    Ks = torch.from_numpy(np.stack([c.K for c in cams]).astype(np.float32)).to(DEV)
    bgs = torch.rand(C, 4, device=DEV) * 0.5 if bg else None
    kw = dict(sh_degree=2, render_mode="RGB+ED", isect_capacity=600_000, lean_meta=True, raster_schedule=sched, backgrounds=bgs)
    with torch.no_grad():
        c0, a0, _ = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"], vm, Ks, W, H, **kw)
        for keep in (True, False):
            rgba = torch.zeros(C, H, W, 4, dtype=torch.uint8, device=DEV)
            dist = torch.zeros(C, H, W, 1, dtype=dt, device=DEV) if dt is not None else None
            c1, a1, _ = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"], vm, Ks, W, H,
                                      dataset_out=(rgba, dist, cams[0].K, keep), **kw)
            if keep:
                assert torch.equal(c1, c0) and torch.equal(a1, a0)              # the float frame beside it: the same bits
            for c in range(C):
                r_ref, d_ref = frame_to_dataset(c0[c], a0[c], cams[0].K if dt is not None else None, distance_dtype=dt)
                assert torch.equal(rgba[c], r_ref), (c, keep)
                if dt is not None:
                    assert torch.equal(dist[c], d_ref), (c, keep)
            assert int((rgba[..., 3] > 0).sum()) > 0.2 * C * W * H
    # what the output needs: "RGB+ED" inference frames through the one-call path
    rgba = torch.zeros(C, H, W, 4, dtype=torch.uint8, device=DEV)
    from robosimgs_amd import _lib
    with pytest.raises(ValueError):
        rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"], vm, Ks, W, H, sh_degree=2,
                      render_mode="RGB+ED", dataset_out=(rgba, None, None, True))          # no capacity: not the lean path
    with pytest.raises(_lib.MgsError):
        rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"], vm, Ks, W, H, sh_degree=2,
                      render_mode="RGB+D", isect_capacity=600_000, lean_meta=True, dataset_out=(rgba, None, None, True))


@pytest.mark.parametrize("dt", [torch.float16, torch.float32])
def test_frame_renderer_dataset_frames(dt):
    """FrameRenderer(dataset_output=, dataset_K=): the slots' graphs write the dataset frame from the raster; every frame
    of a sequence is, byte for byte, frame_to_dataset of the float frame a plain renderer gives for the same camera --
    as three views of one flat buffer (what the multi-GPU loop copies into its gather batch)."""
    from robosimgs_amd import FrameRenderer, camera_ring, synthetic_scene
    from robosimgs_amd.dataset import frame_to_dataset
    W, H = 208, 144
    g = synthetic_scene(20_000, math.log(0.05), 2, 4)
    cams = camera_ring(7, W, H)
    t = g.to_torch(DEV, 2)
    plain = FrameRenderer(t, W, H, render_mode="RGB+ED", frames_in_flight=3, isect_capacity=600_000)
    ref = {}
    plain.render_sequence(cams, lambda i, f: ref.__setitem__(i, frame_to_dataset(f["colors"], f["alphas"], cams[0].K,
                                                                                 distance_dtype=dt)))
    seen = []
    for keep in (False, True):
        fr = FrameRenderer(t, W, H, render_mode="RGB+ED", frames_in_flight=3, isect_capacity=600_000, dataset_output=dt,
                           dataset_K=cams[0].K, dataset_keep_float=keep)

        def consume(i, f):
            rgba, dist = ref[i]
            assert torch.equal(f["rgba"], rgba) and torch.equal(f["distance"], dist)
            n = W * H * 4
            assert f["dataset"].dtype == torch.uint8 and f["dataset"].numel() == n + W * H * dist.element_size()
            assert torch.equal(f["dataset"][:n].view(H, W, 4), rgba)
            assert torch.equal(f["dataset"][n:].view(dt).view(H, W, 1), dist)
            assert (f["colors"] is None and f["alphas"] is None) if not keep else f["colors"].shape == (H, W, 4)
            seen.append(i)
        fr.render_sequence(cams, consume)
    assert seen == list(range(7)) * 2
    with pytest.raises(ValueError):
        FrameRenderer(t, W, H, render_mode="RGB", isect_capacity=600_000, dataset_output=dt, dataset_K=cams[0].K)
    with pytest.raises(ValueError):
        FrameRenderer(t, W, H, render_mode="RGB+ED", isect_capacity=600_000, dataset_output=dt)


def test_dataset_only_renderer_through_the_convenience_paths_and_its_argument_checks():
    """A renderer built with dataset_output= (no float frame) through FrameRenderer.render() and
    render_sharded(renderer=): both return the dataset frames; float64 distances on an odd pixel count (the distance
    plane is then not 8-byte aligned inside a packed frame); a submitted K that is not dataset_K and output buffers of
    the wrong size or device are refused before any kernel writes through them."""
    from robosimgs_amd import FrameRenderer, camera_ring, rasterization, synthetic_scene
    from robosimgs_amd.dataset import frame_to_dataset
    from robosimgs_amd.distributed import render_sharded
    W, H = 203, 131                                   # odd x odd: W * H * 4 is 4 mod 8
    g = synthetic_scene(20_000, math.log(0.05), 2, 4)
    cams = camera_ring(4, W, H)
    t = g.to_torch(DEV, 2)
    plain = FrameRenderer(t, W, H, render_mode="RGB+ED", frames_in_flight=2, isect_capacity=600_000)
    for dt in (torch.float64, torch.float16):
        fr = FrameRenderer(t, W, H, render_mode="RGB+ED", frames_in_flight=2, isect_capacity=600_000, dataset_output=dt,
                           dataset_K=cams[0].K)
        refs = []
        for cam in cams:
            f = plain.render(cam.viewmat(), cam.K)
            assert f["colors"].shape == (H, W, 4) and "rgba" not in f
            refs.append(frame_to_dataset(f["colors"], f["alphas"], cam.K, distance_dtype=dt))
        out = fr.render(cams[1].viewmat(), cams[1].K)
        assert out["colors"] is None and out["alphas"] is None and "dataset" not in out
        assert torch.equal(out["rgba"], refs[1][0]) and torch.equal(out["distance"], refs[1][1])
        vms = torch.from_numpy(np.stack([c.viewmat() for c in cams]).astype(np.float32)).to(DEV)
        Ks = torch.from_numpy(np.stack([c.K for c in cams]).astype(np.float32)).to(DEV)
        rgba, dist, mine = render_sharded(t, vms, Ks, W, H, gather=False, renderer=fr, render_mode="RGB+ED")
        assert list(mine) == [0, 1, 2, 3] and rgba.dtype == torch.uint8 and dist.dtype == dt
        for i in range(4):
            assert torch.equal(rgba[i], refs[i][0]) and torch.equal(dist[i], refs[i][1]), i
        with pytest.raises(ValueError):
            render_sharded(t, vms, Ks, W, H, gather=False, renderer=fr, render_mode="RGB+ED", as_u8=True)
        K_other = cams[0].K.copy()
        K_other[0, 0] *= 1.01
        with pytest.raises(ValueError):
            fr.submit(cams[0].viewmat(), K_other)
        with pytest.raises(ValueError):
            fr.submit(FrameRenderer.pack_camera(cams[0].viewmat(), K_other))          # packed on the host: checked too
        tk = fr.submit(cams[0].viewmat(), cams[0].K)                                  # (the refusals took no slot)
        fr.fetch(tk)
        fr.release(tk)
    vm, K1 = vms[:2], Ks[:2]
    kw = dict(sh_degree=2, render_mode="RGB+ED", isect_capacity=600_000, lean_meta=True)
    good_rgba = torch.zeros(2, H, W, 4, dtype=torch.uint8, device=DEV)
    good_dist = torch.zeros(2, H, W, 1, dtype=torch.float32, device=DEV)
    for rgba_, dist_ in ((torch.zeros(H, W, 4, dtype=torch.uint8, device=DEV), good_dist),          # one frame for two cameras
                         (good_rgba, torch.zeros(H, W, 1, dtype=torch.float32, device=DEV)),
                         (good_rgba, torch.zeros(2, H, W - 1, 1, dtype=torch.float32, device=DEV))):
        with pytest.raises(ValueError):
            rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"], vm, K1, W, H,
                          dataset_out=(rgba_, dist_, cams[0].K, False), **kw)
    from robosimgs_amd import _lib
    with pytest.raises(_lib.MgsError):
        rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"], vm, K1, W, H,
                      dataset_out=(good_rgba.cpu(), good_dist, cams[0].K, False), **kw)
    with pytest.raises(ValueError):
        rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"], vm, K1, W, H,
                      dataset_out=(good_rgba, good_dist, None, False), **kw)
