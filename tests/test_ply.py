""".ply layout of nerfstudio's `ns-export gaussian-splat`."""
import numpy as np
import pytest

from robosimgs_amd import Gaussians, load_ply, save_ply, synthetic_scene


def test_round_trip_and_channel_major_rest(tmp_path):
    g = synthetic_scene(257, -2.0, 3, seed=3)
    p = tmp_path / "a.ply"
    save_ply(str(p), g)
    h = load_ply(str(p))
    for k in ("means", "log_scales", "quats", "opacity_logits", "sh_dc", "sh_rest"):
        np.testing.assert_array_equal(getattr(g, k), getattr(h, k))
    assert h.sh_degree == 3 and h.sh_coeffs.shape == (257, 16, 3)
    # f_rest is channel-major on disk: property f_rest_0 is coefficient 1 of RED, f_rest_15 is
    # coefficient 1 of GREEN
    raw = open(p, "rb").read()
    header_end = raw.index(b"end_header\n") + len(b"end_header\n")
    names = [l.split()[-1] for l in raw[:header_end].decode().splitlines() if l.startswith("property")]
    row0 = np.frombuffer(raw[header_end:header_end + 4 * len(names)], dtype="<f4")
    assert row0[names.index("f_rest_0")] == g.sh_rest[0, 0, 0]
    assert row0[names.index("f_rest_15")] == g.sh_rest[0, 0, 1]
    assert row0[names.index("f_rest_1")] == g.sh_rest[0, 1, 0]
    assert row0[names.index("opacity")] == g.opacity_logits[0]


def test_header_by_name_extra_props_uint8_colors_and_nan_rows(tmp_path):
    n = 5
    names = ["opacity", "junk", "x", "y", "z", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1",
             "rot_2", "rot_3"]
    rec = np.dtype([(k, "<f4") for k in names] + [("red", "u1"), ("green", "u1"), ("blue", "u1")])
    a = np.zeros(n, rec)
    for i, k in enumerate(names):
        a[k] = np.arange(n) + 10 * i
    a["red"], a["green"], a["blue"] = 255, 0, 128
    a["x"][3] = np.nan                                     # exporter-style bad row: must be dropped
    header = "ply\nformat binary_little_endian 1.0\ncomment made by a test\nelement vertex %d\n" % n
    header += "".join(f"property float {k}\n" for k in names)
    header += "property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n"
    p = tmp_path / "b.ply"
    p.write_bytes(header.encode() + a.tobytes())
    g = load_ply(str(p))
    assert len(g) == 4 and g.sh_degree == 0
    np.testing.assert_array_equal(g.means[:, 0], [20, 21, 22, 24])
    np.testing.assert_array_equal(g.opacity_logits, [0, 1, 2, 4])
    np.testing.assert_allclose(0.5 + 0.2820947917738781 * g.sh_dc[0], [1.0, 0.0, 128 / 255], atol=1e-6)


def test_ascii_and_errors(tmp_path):
    g = synthetic_scene(3, -2.0, 0, seed=1)
    names = ["x", "y", "z", "f_dc_0", "f_dc_1", "f_dc_2", "opacity", "scale_0", "scale_1", "scale_2",
             "rot_0", "rot_1", "rot_2", "rot_3"]
    rows = np.concatenate([g.means, g.sh_dc, g.opacity_logits[:, None], g.log_scales, g.quats], 1)
    txt = "ply\nformat ascii 1.0\nelement vertex 3\n" + "".join(f"property float {k}\n" for k in names)
    txt += "end_header\n" + "\n".join(" ".join(repr(float(v)) for v in r) for r in rows) + "\n"
    p = tmp_path / "c.ply"
    p.write_text(txt)
    h = load_ply(str(p))
    np.testing.assert_allclose(h.means, g.means, rtol=1e-6)
    np.testing.assert_allclose(h.quats, g.quats, rtol=1e-6)
    (tmp_path / "bad.ply").write_text("plx\n")
    with pytest.raises(ValueError):
        load_ply(str(tmp_path / "bad.ply"))
    (tmp_path / "miss.ply").write_text("ply\nformat ascii 1.0\nelement vertex 1\nproperty float x\nend_header\n0\n")
    with pytest.raises(ValueError, match="missing"):
        load_ply(str(tmp_path / "miss.ply"))
    with pytest.raises(ValueError):
        Gaussians(np.zeros((2, 3)), np.zeros((2, 3)), np.zeros((2, 4)), np.zeros(2), np.zeros((2, 3)),
                  np.zeros((2, 5, 3)))                      # K = 6 is not a square


def test_activations_and_transform():
    g = synthetic_scene(100, -3.0, 1, seed=2)
    np.testing.assert_allclose(g.scales, np.exp(g.log_scales))
    np.testing.assert_allclose(g.opacities, 1 / (1 + np.exp(-g.opacity_logits)), rtol=1e-6)
    th = 0.7
    R = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]])
    h = g.transformed(R, np.array([1.0, 2.0, 3.0]), 2.0)
    np.testing.assert_allclose(h.means, 2 * g.means @ R.T + [1, 2, 3], rtol=1e-5, atol=1e-5)
    # covariance transforms as s^2 R Sigma R^T
    from oracle import gs_oracle_np as O
    S0, S1 = O.covar_world(g.quats, g.scales), O.covar_world(h.quats, h.scales)
    np.testing.assert_allclose(S1, 4 * R @ S0 @ R.T, rtol=2e-5, atol=1e-9)


def test_rotating_the_scene_rotates_the_sh_colour_field():
    """Rigidly moving scene AND camera must not change the image: checks means / quats / extents
    and the per-degree SH rotation together, through the oracle (CPU)."""
    import math
    from oracle import gs_oracle_np as O
    from robosimgs_amd import Camera, camera_ring
    from robosimgs_amd.gaussians import sh_rotation_matrices, _sh_basis_np
    g = synthetic_scene(1500, math.log(0.12), 3, seed=4)
    g.sh_rest[:] *= 6.0                                     # make view dependence large
    cam = camera_ring(1, 96, 64, thetas=[0.8], radius=6.0)[0]
    rng = np.random.default_rng(0)
    Q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    R = Q * np.sign(np.linalg.det(Q))                        # proper rotation
    t, s = np.array([0.4, -1.0, 0.3]), 1.7
    # M_l are orthogonal and reproduce the rotated basis
    d = rng.normal(size=(10, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    for l, M in enumerate(sh_rotation_matrices(R, 3)):
        np.testing.assert_allclose(M @ M.T, np.eye(2 * l + 1), atol=1e-10)
        np.testing.assert_allclose(_sh_basis_np(3, d)[:, l * l:(l + 1) ** 2] @ M,
                                   _sh_basis_np(3, d @ R)[:, l * l:(l + 1) ** 2], atol=1e-10)
    h = g.transformed(R, t, s)
    S = np.eye(4)
    S[:3, :3], S[:3, 3] = s * R, t
    c2w = S @ cam.c2w
    c2w[:3, :3] /= s                                         # camera keeps unit axes
    cam2 = Camera(c2w, cam.fx, cam.fy, cam.cx, cam.cy, 96, 64)
    a, aa, _ = O.render(g.means, g.quats, g.scales, g.opacities, g.sh_coeffs, cam.viewmat(), cam.K, 96, 64, sh_degree=3)
    # the moved scene is s times larger and s times farther: same pixels
    b, ba, _ = O.render(h.means, h.quats, h.scales, h.opacities, h.sh_coeffs, cam2.viewmat(), cam2.K, 96, 64,
                        sh_degree=3, near_plane=0.01 * s)
    assert np.abs(a).max() > 0.5
    np.testing.assert_allclose(b, a, atol=5e-4)
    np.testing.assert_allclose(ba, aa, atol=5e-4)


def test_pack_transforms_layout_and_validation():
    """Host packing for mgs_transform_gaussians: {sR | t | q_R | s} and the per-degree SH matrices."""
    import pytest
    from robosimgs_amd.transform import pack_transforms
    from robosimgs_amd.gaussians import sh_rotation_matrices
    c, s = np.cos(0.4), np.sin(0.4)
    R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])
    x, rot = pack_transforms([np.eye(3), R], [[0, 0, 0], [1, 2, 3]], [1.0, 2.0], sh_degree=3)
    assert x.shape == (2, 20) and rot.shape == (2, 84) and x.dtype == np.float32
    np.testing.assert_allclose(x[1, :9].reshape(3, 3), 2.0 * R, atol=1e-6)
    np.testing.assert_allclose(x[1, 9:12], [1, 2, 3])
    np.testing.assert_allclose(x[1, 12:16], [np.cos(0.2), 0, 0, np.sin(0.2)], atol=1e-6)   # wxyz about z
    assert x[1, 16] == 2.0
    Ms = sh_rotation_matrices(R, 3)
    np.testing.assert_allclose(rot[1, :9].reshape(3, 3), Ms[1], atol=1e-6)
    np.testing.assert_allclose(rot[1, 9:34].reshape(5, 5), Ms[2], atol=1e-6)
    np.testing.assert_allclose(rot[1, 34:83].reshape(7, 7), Ms[3], atol=1e-6)
    np.testing.assert_allclose(rot[0, :9].reshape(3, 3), np.eye(3), atol=1e-6)
    with pytest.raises(ValueError):
        pack_transforms([2.0 * np.eye(3)], [[0, 0, 0]])          # scale belongs in `scales`
