"""Per unit of tile_sort_units_kernel (csrc/tile_sort.hip: lists over the LDS list are split into units): start and end on the 100 MHz clock.
Needs a build with MGS_TILE_SORT_FLAGS=-DMGS_TSORT_TIMING (the kernel logs one record per item, mgs_debug_tsort_log reads them):
    MGS_TILE_SORT_FLAGS=-DMGS_TSORT_TIMING python robosimgs_amd/csrc/build.py --force   (here)
    gpurun -- 'SCENE=heavy python scripts/dbg/long_sort_timeline.py'"""
import collections, ctypes, os, runpy, sys
import numpy as np
sys.argv = ["run_stage.py", "binning", "1"]
here = os.path.dirname(os.path.abspath(__file__))
runpy.run_path(os.path.join(here, "..", "run_stage.py"), run_name="__main__")
from robosimgs_amd import _lib
L = _lib.lib()
buf = np.zeros((16384, 6), np.uint64)
L.mgs_debug_tsort_log.restype = ctypes.c_uint
n = L.mgs_debug_tsort_log(ctypes.c_void_p(buf.ctypes.data), 16384)
rows = [tuple(int(x) for x in r) for r in buf[:n] if int(r[3]) != 0xffff]      # (0xffff: the main kernel's tiles, main_sort_timeline.py)
pops = {(r[2], r[3]): r[0] >> 32 for r in rows}
rows = [(r[0] & 0xffffffff,) + r[1:] for r in rows]
if not rows:
    sys.exit("no items logged")
# the last launch only (launches are far more than 300 us apart)
rows.sort(key=lambda r: r[4])
cut = 0
for k in range(1, len(rows)):
    if rows[k][4] - rows[k - 1][4] > 30_000:
        cut = k
rows = rows[cut:]
t_last = max(r[5] for r in rows)
t0 = min(r[4] for r in rows)
print(f"{len(rows)} items, span {(t_last - t0) / 100:.1f} us")
d = sorted((r[5] - r[4]) / 100 for r in rows)
m = sorted(r[1] for r in rows)
print(f"units: {len(rows)} of {len(set(r[2] for r in rows))} lists; entries per unit: median {m[len(m) // 2]}, p90 {m[int(len(m) * .9)]}, max {m[-1]}")
print(f"duration us: median {d[len(d) // 2]:.1f}, p90 {d[int(len(d) * .9)]:.1f}, max {d[-1]:.1f}")
print("the twelve units that end last:")
for r in sorted(rows, key=lambda r: -r[5])[:12]:
    print(f"  wg {r[0]:4d} tile {r[2]:5d} unit {r[3]:3d} ({r[1]:5d} entries): {(r[4] - t0) / 100:7.1f} -> {(r[5] - t0) / 100:7.1f} us")
print("the ten longest units:")
for r in sorted(rows, key=lambda r: r[4] - r[5])[:10]:
    print(f"  wg {r[0]:4d} list at {r[2]:8d} unit {r[3]:3d} ({r[1]:5d} entries): {(r[5] - r[4]) / 100:7.1f} us, buckets popped {pops[(r[2], r[3])] & 0xffff} (over the LDS list: {pops[(r[2], r[3])] >> 16})")
per_wg = collections.Counter(r[0] for r in rows)
print("units per workgroup: max", max(per_wg.values()))
first = {}
for r in rows:
    first[r[0]] = min(first.get(r[0], 1 << 62), r[4])
fs = sorted((v - t0) / 100 for v in first.values())
print("first unit of a workgroup starts at (us): min %.1f, p10 %.1f, median %.1f, p90 %.1f, max %.1f (of %d workgroups)" % (
    fs[0], fs[len(fs) // 10], fs[len(fs) // 2], fs[int(len(fs) * .9)], fs[-1], len(fs)))
