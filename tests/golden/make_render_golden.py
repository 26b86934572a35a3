"""Generate tests/golden/render_small.npz: a small seeded scene with the oracle's forward
outputs (fp64 NumPy, oracle/gs_oracle_np.py) and backward (autograd of oracle/gs_oracle_torch.py).

These vectors are produced by this repo's own oracle, not by the reference: the reference has
no renderer (parity unpinned, see the oracle headers).  They freeze the oracle's behaviour so
that the NumPy, torch, C++ and HIP implementations are all compared to ONE stored answer.
    python tests/golden/make_render_golden.py
"""
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import gs_oracle_np as O            # noqa: E402
from oracle import gs_oracle_torch as OT        # noqa: E402
from robosimgs_amd import camera_ring, synthetic_scene  # noqa: E402

W, H, N, DEG = 80, 48, 600, 2
g = synthetic_scene(N, math.log(0.12), DEG, seed=7)
cam = camera_ring(1, W, H, thetas=[0.9], radius=6.0)[0]
out = dict(width=W, height=H, sh_degree=DEG, means=g.means, quats=g.quats, scales=g.scales,
           opacities=g.opacities, sh_coeffs=g.sh_coeffs, viewmat=cam.viewmat(), K=cam.K)
bg = np.array([0.2, 0.4, 0.6, 0.0])
for mode in ("RGB", "RGB+ED"):
    ch = 3 if mode == "RGB" else 4
    tag = mode.replace("+", "_")
    img, alpha, meta = O.render(g.means, g.quats, g.scales, g.opacities, g.sh_coeffs, cam.viewmat(),
                                cam.K, W, H, sh_degree=DEG, render_mode=mode)
    out[f"{tag}_image"], out[f"{tag}_alpha"] = img, alpha
    img_bg, _, _ = O.render(g.means, g.quats, g.scales, g.opacities, g.sh_coeffs, cam.viewmat(),
                            cam.K, W, H, sh_degree=DEG, render_mode=mode, background=bg[:ch])
    out[f"{tag}_bg_image"] = img_bg
out["background"] = bg
out.update(radii=meta["radii"], means2d=meta["means2d"], depths=meta["depths"], conics=meta["conics"],
           compensations=meta["compensations"], colors=meta["colors"],
           tiles_per_gauss=meta["tiles_per_gauss"], isect_ids=meta["isect_ids"],
           flatten_ids=meta["flatten_ids"], isect_offsets=meta["isect_offsets"],
           last_ids=meta["last_ids"], n_isect=meta["n_isect"], n_vis=meta["n_vis"],
           pair_evals=meta["pair_evals"])

# backward: L = <w_img, image> + <w_alpha, alpha>, RGB mode, no background
rng = np.random.default_rng(8)
w_img, w_alpha = rng.normal(size=(H, W, 3)), rng.normal(size=(H, W))
t = {k: torch.tensor(np.asarray(v, dtype=np.float64), requires_grad=True)
     for k, v in (("means", g.means), ("quats", g.quats), ("scales", g.scales),
                  ("opacities", g.opacities), ("sh_coeffs", g.sh_coeffs))}
img, alpha, _ = OT.render(t["means"], t["quats"], t["scales"], t["opacities"], t["sh_coeffs"],
                          torch.tensor(cam.viewmat()), torch.tensor(cam.K), W, H, sh_degree=DEG)
((img * torch.tensor(w_img)).sum() + (alpha[..., 0] * torch.tensor(w_alpha)).sum()).backward()
out.update(w_img=w_img, w_alpha=w_alpha, torch_image=img.detach().numpy())
for k, v in t.items():
    out["grad_" + k] = v.grad.numpy()
np.savez_compressed(os.path.join(HERE, "render_small.npz"), **out)
print("wrote render_small.npz  n_vis", meta["n_vis"], "n_isect", meta["n_isect"],
      "max|np - torch| image", float(np.abs(out["RGB_image"][..., :3] - (out["torch_image"] + 0)).max()))
