"""The kernel sequence of ONE training step (forward RGB+ED, fused L1, backward) as the stream runs it: name, duration, and the gap
to the previous kernel's end -- which launches are the step's small change.
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/ps -o s -- python scripts/dbg/step_sequence.py run
    python scripts/dbg/step_sequence.py /tmp/ps/s_kernel_trace.csv"""
import csv, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if sys.argv[1] == "run":
    import numpy as np, torch
    from robosimgs_amd import synthetic_scene, camera_ring, rasterization, l1_loss
    g = synthetic_scene(1_000_000, math.log(0.012), 3, 0)
    cam = camera_ring(1, 1920, 1080, thetas=[0.3])[0]
    t = g.to_torch("cuda", 3)
    vm = torch.from_numpy(cam.viewmat().astype(np.float32)).cuda()[None]; K = torch.from_numpy(cam.K.astype(np.float32)).cuda()[None]
    p = {k: t[k].detach().clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "colors")}
    target = torch.rand(1, 1080, 1920, 4, device="cuda")
    for _ in range(4):
        for v in p.values():
            v.grad = None
        c, a, m = rasterization(p["means"], p["quats"], p["scales"], p["opacities"], p["colors"], vm, K, 1920, 1080, sh_degree=3, render_mode="RGB+ED", isect_capacity=4_700_000)
        torch.cuda.synchronize()
        l1_loss(c, target).backward()
        torch.cuda.synchronize()
    sys.exit(0)
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
# the last step: after the last gap of more than 1 ms that is followed by a projection kernel
starts = [i for i, r in enumerate(rows) if "project_color_fwd" in r["Kernel_Name"]]
rows = rows[starts[-1]:]
prev = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void ", "").replace("mgs::(anonymous namespace)::", "")[:70]
    print(f"{name:70s} {(e - s) / 1e3:8.1f} us   gap {((s - prev) / 1e3 if prev else 0):7.1f} us")
    prev = e
