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
    r = FrameRenderer(t, 320, 192, render_mode="RGB+ED", frames_in_flight=3,
                      sizing_camera=(cams[0].viewmat(), cams[0].K))
    got = {}
    r.render_sequence(cams, lambda i, f: got.__setitem__(i, (f["colors"].clone(), f["alphas"].clone())))
    assert sorted(got) == list(range(7))
    for i, cam in enumerate(cams):
        c, a, _ = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"],
                                _t(cam.viewmat())[None], _t(cam.K)[None], 320, 192, sh_degree=2,
                                render_mode="RGB+ED")
        assert torch.equal(got[i][0], c[0]) and torch.equal(got[i][1], a[0]), f"camera {i}"
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
