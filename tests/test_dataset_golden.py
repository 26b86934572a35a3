"""The OUTPUT boundary of the render path, pinned by the reference's own readers.

tests/golden/dataset_reference.npz holds what load_images / load_depths of
/root/reference/Articulation/utils/nerf2physic_utils.py (:84-118) returned for files this repo's DatasetWriter
wrote (tests/golden/make_dataset_golden.py).  Here, without the reference and without a GPU: the host half of the
writer still produces those exact files, the oracle restatement of the device kernel still produces those
exact arrays, and the reference's answers say what they must (mask = alpha > 0, z-depth back to the bit)."""
import os

import numpy as np
import pytest

from oracle import dataset_np as D
from robosimgs_amd import camera as C
from robosimgs_amd.dataset import (DatasetWriter, decode_png_rgba, encode_npy_gz, encode_png_rgba,
                                   read_dataset_frame, unnormalize_points)

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "dataset_reference.npz"))
N = int(G["n_frames"])


@pytest.mark.parametrize("i", range(N))
def test_restatement_and_writer_reproduce_the_files_the_reference_read(i, tmp_path):
    for f64, tag in ((False, "f32"), (True, "f64")):
        rgba, dist = D.frame_to_dataset(G[f"colors_{i}"], G[f"alpha_{i}"], G[f"K_{i}"], G["background"], distance_f64=f64)
        assert np.array_equal(rgba, G[f"rgba_{i}"])
        assert dist.dtype == (np.float64 if f64 else np.float32) and np.array_equal(dist, G[f"distance_{tag}_{i}"])
        img, dep = DatasetWriter(str(tmp_path / tag)).write(i, rgba, dist)
        assert open(img, "rb").read() == G[f"png_bytes_{i}"].tobytes()
        assert open(dep, "rb").read() == G[f"npygz_bytes_{tag}_{i}"].tobytes()
        back_rgba, back_dist = read_dataset_frame(img, dep)
        assert np.array_equal(back_rgba, rgba) and np.array_equal(back_dist, dist[..., 0])


@pytest.mark.parametrize("i", range(N))
def test_what_the_reference_readers_returned(i):
    alpha, z, rgba = G[f"alpha_{i}"], G[f"colors_{i}"][..., 3], G[f"rgba_{i}"]
    assert np.array_equal(G[f"ref_mask_{i}"], alpha > 0)                        # load_images' mask
    assert np.array_equal(G[f"ref_image_raw_{i}"], rgba[..., :3])
    painted = rgba[..., :3].copy()
    painted[~(alpha > 0)] = 255                                                  # bg_change = 255
    assert np.array_equal(G[f"ref_image_bg255_{i}"], painted)
    # load_depths -> distance_to_depth: the ED channel comes back to the last fp32 bit from the f64 file,
    # to one ulp from the f32 file
    assert np.array_equal(G[f"ref_depth_f64_{i}"].astype(np.float32), z)
    assert np.all(np.abs(G[f"ref_depth_f32_{i}"] - z.astype(np.float64)) <= np.spacing(np.abs(z)))
    # this repo's own distance_to_depth (pinned to the reference in test_camera_golden.py) agrees with it
    for tag in ("f32", "f64"):
        mine = C.distance_to_depth(G[f"ref_distance_{tag}_{i}"], G[f"K_{i}"])
        assert np.allclose(mine, G[f"ref_depth_{tag}_{i}"], rtol=1e-14, atol=0)


def test_png_codec_round_trip_and_independent_decoder():
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, size=(37, 53, 4), dtype=np.uint8)
    data = encode_png_rgba(a)
    assert np.array_equal(decode_png_rgba(data), a)
    try:
        from PIL import Image
    except ImportError:
        pytest.skip("PIL not installed: the reference's reader (PIL) was run at golden-generation time")
    import io
    assert np.array_equal(np.array(Image.open(io.BytesIO(data))), a)
    assert encode_npy_gz(a) == encode_npy_gz(a.copy())                           # no time stamp inside


def test_unnormalize_points_is_load_ns_point_cloud_algebra():
    cam = np.load(os.path.join(HERE, "golden", "camera_reference.npz"))
    got = unnormalize_points(cam["dp_points_ns"], cam["dp_transform"], float(cam["dp_scale"]))
    assert np.allclose(got, cam["dp_points_world"], rtol=1e-13, atol=1e-13)


def test_writer_with_a_thread_pool_writes_the_same_files(tmp_path):
    """DatasetWriter(workers=4): encoding and writing on a pool changes when the files appear, not a byte of them."""
    from robosimgs_amd.dataset import DatasetWriter
    rng = np.random.default_rng(3)
    frames = [(rng.integers(0, 256, size=(48, 64, 4), dtype=np.uint8), rng.random((48, 64, 1)).astype(np.float32) * 9) for _ in range(9)]
    a = DatasetWriter(str(tmp_path / "sync"))
    with DatasetWriter(str(tmp_path / "pool"), workers=4, max_pending=3) as b:
        for i, (rgba, dist) in enumerate(frames):
            pa = a.write(i, rgba, dist)
            buf = rgba.copy()
            pb = b.write(i, buf, dist)
            buf[:] = 0                                   # the caller's buffer is its own again at once
    for i in range(len(frames)):
        for x, y in zip(a.paths(i), b.paths(i)):
            assert open(x, "rb").read() == open(y, "rb").read()
    fast = DatasetWriter(str(tmp_path / "fast"), png_level=1, gz_level=1)
    p_img, p_dep = fast.write(0, *frames[0])
    from robosimgs_amd.dataset import read_dataset_frame
    rgba, dist = read_dataset_frame(p_img, p_dep)
    assert np.array_equal(rgba, frames[0][0]) and np.array_equal(dist, frames[0][1][:, :, 0])
