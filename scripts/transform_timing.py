import math, sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
from robosimgs_amd import synthetic_scene, transform_gaussians
from robosimgs_amd.transform import pack_transforms
g = synthetic_scene(1_000_000, math.log(0.012), 3, 0); t = g.to_torch("cuda", 3)
rng = np.random.default_rng(0)
def rot():
    q,_ = np.linalg.qr(rng.normal(size=(3,3))); 
    if np.linalg.det(q) < 0: q[:,0] = -q[:,0]
    return q
Rs = [rot() for _ in range(8)]; ts = [rng.normal(size=3) for _ in range(8)]
gid_all = torch.from_numpy(rng.integers(0, 8, size=1_000_000).astype(np.int32)).cuda()
gid_10 = torch.where(torch.rand(1_000_000, device="cuda") < 0.1, gid_all, torch.full_like(gid_all, -1))
xp, rp = pack_transforms(Rs, ts, None, 3)
packed = (torch.from_numpy(xp).cuda(), torch.from_numpy(rp).cuda())
for name, gid, sh in (("all moving, SH rotated", gid_all, True), ("all moving, no SH", gid_all, False), ("10 % moving in place, SH rotated", gid_10, True)):
    out = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in t.items()}
    for _ in range(3): transform_gaussians(t, group_ids=gid, rotate_sh=sh, out=out, packed=packed)
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): transform_gaussians(t if "in place" not in name else out, group_ids=gid, rotate_sh=sh, out=out, packed=packed)
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1)/50*1e3:.1f} us per call (1 M Gaussians, 8 groups; pre-packed transforms)")
