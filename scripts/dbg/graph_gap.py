"""What a dependent kernel node costs inside a HIP graph: chains of tiny kernels, and of 20-us kernels, replayed."""
import time, torch
dev = "cuda"
x = torch.zeros(64, device=dev)
big = torch.zeros(64 << 20, device=dev)      # 256 MB: one pass ~ 70 us
def chain(n, t):
    for _ in range(n): t.add_(1.0)
for name, t in (("tiny", x), ("256MB add_", big)):
    for n in (1, 10, 50):
        s = torch.cuda.Stream(dev)
        with torch.cuda.stream(s):
            chain(n, t); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                chain(n, t)
        torch.cuda.synchronize()
        for _ in range(5): g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50): g.replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 50
        print(f"{name}: {n} nodes: {dt * 1e6:.1f} us per replay, {dt * 1e6 / n:.2f} us per node")
