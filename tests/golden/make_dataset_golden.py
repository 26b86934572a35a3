"""Generate tests/golden/dataset_reference.npz by running the REFERENCE'S OWN READERS
(/root/reference/Articulation/utils/nerf2physic_utils.py: load_images :84-101, load_depths :104-118, and through
it distance_to_depth :135-146) on files this repo wrote for the golden scene.

    python tests/golden/make_dataset_golden.py        (needs /root/reference and PIL; the tests need neither)

The frame is the golden scene's "RGB+ED" render (tests/golden/render_small.npz, rounded to fp32 as a renderer
delivers it) plus a second, synthetic frame that exercises what the golden render does not (alpha just above
zero, colours outside [0, 1], pixels without any splat, a non-trivial K with skew).  Files are written by the
product's DatasetWriter from the arrays mgs_frame_to_dataset must produce (oracle/dataset_np.py restates the
kernel; the -m gpu test checks the kernel against these arrays bit for bit), then read back by the reference.
The .npz holds data only: the inputs, the arrays written, the file bytes and what the reference returned.
"""
import importlib.util
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import dataset_np as D                       # noqa: E402
from robosimgs_amd.dataset import DatasetWriter          # noqa: E402  (host half only: numpy)

spec = importlib.util.spec_from_file_location(
    "ref_n2p", "/root/reference/Articulation/utils/nerf2physic_utils.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

g = np.load(os.path.join(HERE, "render_small.npz"))
frames = []
# frame 0: the golden scene
frames.append(dict(colors=g["RGB_ED_image"].astype(np.float32), alpha=g["RGB_ED_alpha"][..., 0].astype(np.float32),
                   K=g["K"].astype(np.float64)))
# frame 1: synthetic edge cases, same resolution (one directory = one resolution for the reference's loop)
rng = np.random.default_rng(11)
H, W = frames[0]["alpha"].shape
alpha = rng.uniform(0, 1, size=(H, W)).astype(np.float32)
alpha[:6] = 0.0                                            # no splat at all
alpha[6, :40] = np.float32(1e-6)                           # alpha > 0 that rounds to 0 in 8 bits
alpha[6, 40:] = np.float32(0.0019)                         # rounds to 0 as well (0.0019 * 255 < 0.5)
alpha[7] = 1.0
colors = rng.uniform(-0.2, 1.3, size=(H, W, 4)).astype(np.float32)
colors[..., 3] = rng.uniform(0.3, 40.0, size=(H, W)).astype(np.float32)      # z-depth
colors[:6, :, 3] = 0.0
K1 = np.array([[93.7, 0.31, 41.25], [0.0, 91.2, 22.5], [0.0, 0.0, 1.0]])
frames.append(dict(colors=colors, alpha=alpha, K=K1))
background = np.array([0.2, 0.4, 0.6], dtype=np.float32)

out = {"background": background, "n_frames": np.int64(len(frames))}
for f64 in (False, True):
    tag = "f64" if f64 else "f32"
    with tempfile.TemporaryDirectory() as d:
        wr = DatasetWriter(d)
        for i, fr in enumerate(frames):
            rgba, dist = D.frame_to_dataset(fr["colors"], fr["alpha"], fr["K"], background, distance_f64=f64)
            img_path, dep_path = wr.write(i, rgba, dist)
            out[f"rgba_{i}"] = rgba
            out[f"distance_{tag}_{i}"] = dist
            out[f"png_bytes_{i}"] = np.frombuffer(open(img_path, "rb").read(), np.uint8)
            out[f"npygz_bytes_{tag}_{i}"] = np.frombuffer(open(dep_path, "rb").read(), np.uint8)
        # ---- the reference reads the directories -----------------------------------------------------
        imgs, masks = ref.load_images(wr.image_dir, bg_change=255, return_masks=True)
        raw = ref.load_images(wr.image_dir, bg_change=None)
        depths = ref.load_depths(wr.depth_dir, [fr["K"] for fr in frames])
        dists = ref.load_depths(wr.depth_dir, None)
    for i, fr in enumerate(frames):
        out[f"ref_image_bg255_{i}"], out[f"ref_mask_{i}"], out[f"ref_image_raw_{i}"] = imgs[i], masks[i], raw[i]
        out[f"ref_depth_{tag}_{i}"] = depths[i]
        out[f"ref_distance_{tag}_{i}"] = dists[i]
        # what the pin is about -- checked here, at generation time, with the reference's arrays:
        assert np.array_equal(masks[i], fr["alpha"] > 0), "load_images' mask is not alpha > 0"
        z = fr["colors"][..., 3]
        if f64:
            assert np.array_equal(depths[i].astype(np.float32), z), "load_depths does not give the ED channel back"
        else:
            ulp = np.spacing(np.abs(z))
            assert np.all(np.abs(depths[i] - z.astype(np.float64)) <= ulp), "more than one ulp off"
        assert np.array_equal(raw[i], out[f"rgba_{i}"][..., :3])
for i, fr in enumerate(frames):
    out[f"colors_{i}"], out[f"alpha_{i}"], out[f"K_{i}"] = fr["colors"], fr["alpha"], fr["K"]
np.savez_compressed(os.path.join(HERE, "dataset_reference.npz"), **out)
print("wrote dataset_reference.npz:", {k: v.shape for k, v in out.items() if k.startswith("ref_")})
