"""Row (f1): nerfstudio-export ingestion end to end -- .ply + dataparser_transforms.json +
transforms.json on disk -> Gaussians / Cameras -> render() on the GPU -> oracle."""
import json
import math

import numpy as np
import pytest

from oracle import gs_oracle_np as O
from robosimgs_amd import (camera_ring, cameras_from_transforms_json, load_dataparser_transforms,
                           load_ply, save_ply, synthetic_scene)

pytestmark = pytest.mark.gpu


def test_ply_and_json_to_pixels(tmp_path):
    from robosimgs_amd import render
    world = synthetic_scene(6000, math.log(0.07), 2, 21)
    # what nerfstudio does on export: the scene lives in a normalised frame x_ns = s * (T x_world)
    th = 0.4
    R = np.array([[math.cos(th), 0, math.sin(th)], [0, 1, 0], [-math.sin(th), 0, math.cos(th)]])
    T = np.hstack([R, np.array([[0.3], [-0.2], [0.1]])])
    s = 0.5
    ns = world.transformed(R, s * T[:, 3], s)
    save_ply(str(tmp_path / "splat.ply"), ns)
    (tmp_path / "dataparser_transforms.json").write_text(json.dumps({"transform": T.tolist(), "scale": s}))
    cams = camera_ring(3, 160, 96, radius=6.0)
    frames = [{"file_path": f"f{i}.png", "transform_matrix": c.c2w.tolist()} for i, c in enumerate(cams)]
    (tmp_path / "transforms.json").write_text(json.dumps(
        {"fl_x": cams[0].fx, "fl_y": cams[0].fy, "cx": cams[0].cx, "cy": cams[0].cy, "w": 160, "h": 96,
         "frames": frames}))

    g = load_ply(str(tmp_path / "splat.ply"))
    Tl, sl = load_dataparser_transforms(str(tmp_path / "dataparser_transforms.json"))
    g = g.undo_dataparser_transform(Tl, sl)
    np.testing.assert_allclose(g.means, world.means, atol=2e-5)
    loaded = cameras_from_transforms_json(str(tmp_path / "transforms.json"))
    out = render(g, loaded, render_mode="RGB+ED", background=(0.1, 0.1, 0.1))
    assert out["rgb"].shape == (3, 96, 160, 3) and out["depth"].shape == (3, 96, 160, 1)
    f32 = lambda m: np.asarray(m, dtype=np.float32).astype(np.float64)
    for i, cam in enumerate(cams):
        # (1) the render of what was LOADED (the Gaussians the GPU was given, the camera read back from
        # transforms.json) against the fp64 oracle on those very inputs: the zero-unexplained-pixels gate
        lc = loaded[i]
        ref, ra, rm = O.render(g.means, g.quats, g.scales, g.opacities, g.sh_coeffs, f32(lc.viewmat()), f32(lc.K),
                               160, 96, sh_degree=2, render_mode="RGB+ED", margins=True, flip_eps=O.EPS_PATH)
        rgb_ref = np.clip(ref[..., :3] + (1 - ra) * 0.1, 0, 1)
        got = np.concatenate([out["rgb"][i].cpu().numpy(), out["depth"][i].cpu().numpy()], -1)
        want = np.concatenate([rgb_ref, np.where(ra > 0, ref[..., 3:], ref[..., 3].max())], -1)
        lit = ra[..., 0] > 0                    # splatfacto's post-processing replaces depth where alpha == 0
        got[~lit, 3] = want[~lit, 3]
        O.check_frame(got, out["alpha"][i].cpu().numpy(), want, ra, rm["margins"], O.EPS_PATH, rm["edge_mask"],
                      expected_depth=True, what=f"loaded scene, camera {i}", flip_weight=rm["flip_weight"],
                      feat_max=np.maximum(rm["feat_max"], [0.1, 0.1, 0.1, 0.0]), require_flip_bound=True)
        # (2) the export / import round trip (positions, orientations AND the SH colour field rotated there and
        # back, fp32 on disk) reproduces the ORIGINAL scene's image to the precision of that round trip
        ref0, ra0, _ = O.render(world.means, world.quats, world.scales, world.opacities, world.sh_coeffs,
                                cam.viewmat(), cam.K, 160, 96, sh_degree=2, render_mode="RGB+ED")
        rgb0 = np.clip(ref0[..., :3] + (1 - ra0) * 0.1, 0, 1)
        d = np.abs(out["rgb"][i].cpu().numpy() - rgb0).max(-1)
        assert (d > 5e-4).mean() < 1e-3, f"camera {i}: {int((d > 5e-4).sum())} px off (max {d.max():.2e})"
        np.testing.assert_allclose(out["alpha"][i, ..., 0].cpu().numpy(), ra0[..., 0], atol=2e-4)
