"""Dev tool: which torch streams share a hardware queue?  Two streams that share one serialise a pair of spin kernels."""
import sys
import time

import torch

dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
streams = [torch.cuda.Stream(dev) for _ in range(N)]
print([hex(s.cuda_stream) for s in streams])
CYC = 400_000      # ~0.2 ms


def pair(a, b):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(a):
        torch.cuda._sleep(CYC)
    with torch.cuda.stream(b):
        torch.cuda._sleep(CYC)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


for s in streams:           # first use
    with torch.cuda.stream(s):
        torch.cuda._sleep(1000)
torch.cuda.synchronize()
one = min(pair(streams[0], streams[0]) for _ in range(3)) / 2
print(f"one spin {one * 1e3:.3f} ms")
cur = torch.cuda.current_stream()
allst = [cur] + streams
for i, a in enumerate(allst):
    row = []
    for j, b in enumerate(allst):
        if j == i:
            row.append(" .")
            continue
        r = min(pair(a, b) for _ in range(2)) / one
        row.append(" S" if r > 1.6 else " -")
    print(f"{i - 1:3d}" + "".join(row))
