"""Data-generation loop for a scene with an articulated part (the use RoboSimGS is built for):
a static 3DGS background, one group of Gaussians that follows a hinge, an opaque simulator layer
composited by depth, 8-bit frames out.  Synthetic inputs, so it runs anywhere an MI355X is visible:

    python examples/articulated_scene.py [n_frames]
"""
import math
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from robosimgs_amd import (FrameRenderer, camera_ring, composite_over, frame_to_u8,  # noqa: E402
                           synthetic_scene)


def hinge(angle, axis_point):
    """Rotation by `angle` about the vertical line through `axis_point` (a door / lid hinge)."""
    c, s = math.cos(angle), math.sin(angle)
    R = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
    p = np.asarray(axis_point, dtype=np.float64)
    return R, p - R @ p


def main():
    n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    W, H = 1280, 720
    scene = synthetic_scene(300_000, math.log(0.02), 3, seed=0)
    tensors = scene.to_torch("cuda", 3)
    # the Gaussians inside a box are "the door": group 0; everything else is static (-1)
    m = tensors["means"]
    door = (m[:, 0] > 0.5) & (m[:, 0] < 2.0) & (m[:, 1].abs() < 0.3) & (m[:, 2].abs() < 1.0)
    group_ids = torch.where(door, 0, -1).to(torch.int32)
    cams = camera_ring(n_frames, W, H, radius=7.0)
    r = FrameRenderer(tensors, W, H, render_mode="RGB+ED", frames_in_flight=3,
                      sizing_camera=(cams[0].viewmat(), cams[0].K), capacity_margin=2.0,
                      group_ids=group_ids, n_groups=1)
    # a stand-in for the simulator's render of the robot: an opaque disc at 6 m depth
    yy, xx = torch.meshgrid(torch.arange(H, device="cuda"), torch.arange(W, device="cuda"), indexing="ij")
    disc = ((xx - W / 2) ** 2 + (yy - H / 2) ** 2) < 90 ** 2
    fg_rgb = torch.tensor([0.9, 0.3, 0.1], device="cuda").expand(H, W, 3).contiguous()
    fg_depth = torch.where(disc, 6.0, 0.0)

    frames = []

    def consume(i, f):
        rgb, depth = composite_over(f["colors"][..., :3], f["alphas"], f["colors"][..., 3], fg_rgb, fg_depth,
                                    backdrop=(0.05, 0.05, 0.08))
        frames.append(frame_to_u8(rgb, torch.ones(H, W, device="cuda")).cpu())     # already composited: alpha 1

    tickets, nxt = [], 0
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for i in range(n_frames):
        while nxt < n_frames and len(tickets) < r.n_slots:
            R, t = hinge(0.9 * math.sin(2 * math.pi * nxt / n_frames), (0.5, 0.0, 0.0))
            tickets.append(r.submit(cams[nxt].viewmat(), cams[nxt].K, rotations=[R], translations=[t]))
            nxt += 1
        tk = tickets.pop(0)
        consume(i, r.fetch(tk))
        r.release(tk)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{n_frames} frames of {W}x{H}, {int(door.sum())} of {len(scene)} Gaussians on the hinge: "
          f"{n_frames / dt:.0f} frames/s including compositing, 8-bit conversion and download")
    print("mean pixel value of the first / last frame:", float(frames[0].float().mean()), float(frames[-1].float().mean()))


if __name__ == "__main__":
    main()
