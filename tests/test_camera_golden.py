"""Camera / nerfstudio-format conventions against vectors generated from the REFERENCE's own
helpers (tests/golden/make_camera_golden.py imports
/root/reference/Articulation/utils/nerf2physic_utils.py) and against the reference's committed
segmentation output (part-mesh centroids vs SAM masks)."""
import json
import os

import numpy as np
import pytest

from oracle import gs_oracle_np as O
from robosimgs_amd import (Camera, Gaussians, cameras_from_transforms_json, depth_to_distance,
                           distance_to_depth, load_dataparser_transforms, unproject_point)

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "camera_reference.npz"))


def _cams():
    return [Camera.from_c2w_opengl(G["cam_c2w"][i], G["cam_K"][i], *G["cam_res"][i])
            for i in range(len(G["cam_names"]))]


def test_project_matches_reference_project_3d_to_2d():
    for i, cam in enumerate(_cams()):
        uv, dist = cam.project(G["pts"], return_dists=True)
        np.testing.assert_allclose(uv, G["proj_uv"][i], rtol=1e-10, atol=1e-9)
        np.testing.assert_allclose(dist, G["proj_dist"][i], rtol=1e-12)


def test_viewmat_is_the_oracles_and_round_trips():
    for cam in _cams():
        np.testing.assert_allclose(cam.viewmat(), O.viewmat_from_c2w_opengl(cam.c2w), atol=1e-13)
        back = Camera.from_w2c_opencv(cam.viewmat(), cam.K, cam.width, cam.height)
        np.testing.assert_allclose(back.c2w, cam.c2w, atol=1e-12)
        np.testing.assert_allclose(O.campos_from_viewmat(cam.viewmat()), cam.position, atol=1e-12)


def test_depth_distance_and_unproject():
    np.testing.assert_allclose(depth_to_distance(G["dd_depth"], G["dd_K"]), G["dd_distance"], rtol=1e-12)
    np.testing.assert_allclose(distance_to_depth(G["dd_distance"], G["dd_K"]), G["dd_roundtrip"], rtol=1e-12)
    for (u, v), xyz in zip(G["unproj_px"], G["unproj_xyz"]):
        np.testing.assert_allclose(unproject_point((int(u), int(v)), G["dd_depth"], G["unproj_c2w"],
                                                   G["dd_K"]), xyz, rtol=1e-12, atol=1e-12)


def test_transforms_json_and_dataparser(tmp_path):
    pt, pd = tmp_path / "transforms.json", tmp_path / "dataparser_transforms.json"
    pt.write_text(str(G["transforms_json"]))
    pd.write_text(str(G["dataparser_json"]))
    cams = cameras_from_transforms_json(str(pt))
    assert len(cams) == 3
    for i, cam in enumerate(cams):                       # per-frame intrinsics win, as in the reference
        np.testing.assert_allclose(cam.c2w, G["tj_c2ws"][i])
        np.testing.assert_allclose(cam.K, G["tj_K_frames"][i])
        np.testing.assert_allclose(np.linalg.inv(cam.c2w), G["tj_w2cs"][i], atol=1e-12)
    tj = json.loads(str(G["transforms_json"]))
    for fr in tj["frames"]:
        for k in ("fl_x", "fl_y", "cx", "cy"):
            fr.pop(k)
    pt.write_text(json.dumps(tj))
    for cam in cameras_from_transforms_json(str(pt)):    # global intrinsics
        np.testing.assert_allclose(cam.K, G["tj_K_global"])
        assert (cam.width, cam.height) == (800, 600)
    T, s = load_dataparser_transforms(str(pd))
    np.testing.assert_allclose(T, G["dp_transform"])
    assert s == float(G["dp_scale"])
    n = len(G["dp_points_ns"])
    g = Gaussians(G["dp_points_ns"], np.zeros((n, 3)), np.tile([1.0, 0, 0, 0], (n, 1)), np.zeros(n),
                  np.zeros((n, 3)), np.zeros((n, 0, 3)))
    world = g.undo_dataparser_transform(T, s)
    np.testing.assert_allclose(world.means, G["dp_points_world"], rtol=2e-6, atol=2e-6)
    # extents scale by 1/scale (similarity), orientation by the inverse rotation
    np.testing.assert_allclose(np.exp(world.log_scales), 1.0 / s, rtol=1e-5)


def test_real_data_centroids_land_in_their_masks():
    """25k face centroids of the reference's exported parts project into the SAM masks that
    produced them (subsampled 1/8): pins OpenGL c2w + the image-Y flip the reference applies
    (interactive_segmenter.py:1436-1460) against our OpenCV viewmat (no flip needed)."""
    names = [str(n) for n in G["cam_names"]]
    cam = _cams()[names.index(str(G["mask_camera"]))]
    masks = {c: np.unpackbits(G[f"mask_{c}_bits"])[: np.prod(G[f"mask_{c}_shape"])]
             .reshape(G[f"mask_{c}_shape"]).astype(bool) for c in ("RED", "GREEN")}
    assert not (masks["RED"] & masks["GREEN"]).any()
    for pts, own, other in ((G["lid_centroids"], "RED", "GREEN"), (G["body_centroids"], "GREEN", "RED")):
        uv = cam.project(pts.astype(np.float64))
        u, v = np.floor(uv[:, 0]).astype(int), np.floor(uv[:, 1]).astype(int)
        inside = (u >= 0) & (u < cam.width) & (v >= 0) & (v < cam.height)
        assert inside.all()
        assert masks[own][v, u].mean() > 0.995, masks[own][v, u].mean()
        assert masks[other][v, u].mean() < 0.005


def test_look_at_matches_reference_recipe():
    # reference recipe (interactive_segmenter.py:279-313): fov 50, 800 px -> f = 400/tan(25 deg)
    cam = Camera.look_at((3.0, -2.0, 1.0), (0, 0, 0), (0, 1, 0), 800, 800, 50.0)
    assert abs(cam.fx - float(G["cam_K"][0][0, 0])) < 1e-9
    uv = cam.project(np.zeros((1, 3)))
    np.testing.assert_allclose(uv[0], [400.0, 400.0], atol=1e-9)      # target at the principal point
    R = cam.c2w[:3, :3]
    np.testing.assert_allclose(R.T @ R, np.eye(3), atol=1e-12)
    assert np.linalg.det(R) > 0


def test_frame_renderer_pack_camera_layout():
    """(viewmat | K) as one 25-float tensor: the layout FrameRenderer.submit uploads in one copy."""
    import torch
    from robosimgs_amd import camera_ring
    from robosimgs_amd.pipeline import FrameRenderer
    cam = camera_ring(1, 320, 200, thetas=[0.7])[0]
    p = FrameRenderer.pack_camera(cam.viewmat(), cam.K)
    assert p.dtype == torch.float32 and p.shape == (25,)
    np.testing.assert_allclose(p[:16].reshape(4, 4).numpy(), cam.viewmat().astype(np.float32))
    np.testing.assert_allclose(p[16:].reshape(3, 3).numpy(), cam.K.astype(np.float32))
    q = FrameRenderer.pack_camera(torch.from_numpy(cam.viewmat()), torch.from_numpy(cam.K))
    assert torch.equal(p, q)
