"""Row (f2): depth-tested compositing of the splat background with an opaque foreground layer."""
import math

import numpy as np
import pytest
import torch

from robosimgs_amd import camera_ring, synthetic_scene

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rule(bg, a, zb, fg, zf, mask, backdrop):
    """NumPy statement of the rule in include/mgs.h."""
    has = mask.astype(bool) if mask is not None else ((zf > 0) & np.isfinite(zf))
    front = has & (~(a > 0) | (zf <= zb))
    src = np.where(has[..., None], fg, np.asarray(backdrop, dtype=np.float64)[None, None])
    rgb = np.where(front[..., None], src, bg + (1 - a)[..., None] * src)
    depth = np.where(front, zf, np.where(a > 0, zb, np.where(has, zf, np.inf)))
    return rgb, depth


@pytest.mark.parametrize("use_mask", [False, True])
def test_composite_matches_rule_on_a_real_render(use_mask):
    from robosimgs_amd import composite_over, rasterization
    g = synthetic_scene(8000, math.log(0.08), 1, 2)
    cam = camera_ring(1, 200, 120, thetas=[0.5])[0]
    t = g.to_torch(DEV, 1)
    f = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(DEV)
    c, a, _ = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"],
                            f(cam.viewmat())[None], f(cam.K)[None], 200, 120, sh_degree=1, render_mode="RGB+ED")
    rng = np.random.default_rng(0)
    fg = rng.random((120, 200, 3)).astype(np.float32)
    zf = rng.uniform(3.0, 11.0, size=(120, 200)).astype(np.float32)
    present = rng.random((120, 200)) < 0.6
    mask = present.astype(np.uint8) if use_mask else None
    if not use_mask:
        zf = np.where(present, zf, np.inf).astype(np.float32)
    rgb, depth = composite_over(c[0, ..., :3], a[0], c[0, ..., 3:], f(fg), f(zf),
                                torch.from_numpy(mask).to(DEV) if use_mask else None, backdrop=(0.2, 0.3, 0.4))
    r_rgb, r_depth = _rule(c[0, ..., :3].cpu().numpy().astype(np.float64), a[0, ..., 0].cpu().numpy().astype(np.float64),
                           c[0, ..., 3].cpu().numpy().astype(np.float64), fg.astype(np.float64),
                           zf.astype(np.float64), mask, (0.2, 0.3, 0.4))
    np.testing.assert_allclose(rgb.cpu().numpy(), r_rgb, atol=1e-6)
    np.testing.assert_array_equal(depth.cpu().numpy(), r_depth.astype(np.float32))
    # both occlusion orders actually occur in this scene
    av, zb = a[0, ..., 0].cpu().numpy(), c[0, ..., 3].cpu().numpy()
    assert (present & (av > 0) & (zf <= zb)).any() and (present & (av > 0) & (zf > zb)).any()


@pytest.mark.parametrize("mode,bg", [("RGB", None), ("RGB+ED", (0.2, 0.4, 0.9))])
def test_frame_to_u8_is_splatfacto_postprocessing_quantised(mode, bg):
    from robosimgs_amd import frame_to_u8, rasterization
    g = synthetic_scene(8000, math.log(0.08), 1, 2)
    cam = camera_ring(1, 203, 117, thetas=[0.5])[0]
    t = g.to_torch(DEV, 1)
    f = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(DEV)
    c, a, _ = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"],
                            f(cam.viewmat())[None], f(cam.K)[None], 203, 117, sh_degree=1, render_mode=mode)
    u8 = frame_to_u8(c[0], a[0], bg)
    assert u8.dtype == torch.uint8 and u8.shape == (117, 203, 3)
    back = torch.tensor(bg if bg is not None else (0.0, 0.0, 0.0), device=DEV)
    ref = (c[0, ..., :3] + (1 - a[0]) * back).clamp(0, 1)
    diff = (u8.float() - ref * 255).abs()
    assert float(diff.max()) <= 0.5 + 1e-3                     # round to nearest
    assert int(u8.max()) > 100 and int((u8 == 0).sum()) >= 0
