"""Per tile of tile_depth_sort_kernel (csrc/tile_sort.hip, the per-tile sort's main kernel): start and end on the 100 MHz clock, against the
tile's list length.  Needs a build with MGS_TILE_SORT_FLAGS=-DMGS_TSORT_TIMING:
    MGS_TILE_SORT_FLAGS=-DMGS_TSORT_TIMING python robosimgs_amd/csrc/build.py --force   (here)
    gpurun -- 'SCENE=heavy python scripts/dbg/main_sort_timeline.py'"""
import ctypes, os, runpy, sys
import numpy as np
sys.argv = ["run_stage.py", "binning", "1"]
here = os.path.dirname(os.path.abspath(__file__))
g = runpy.run_path(os.path.join(here, "..", "run_stage.py"), run_name="__main__")
from robosimgs_amd import _lib
L = _lib.lib()
buf = np.zeros((16384, 6), np.uint64)
L.mgs_debug_tsort_log.restype = ctypes.c_uint
L.mgs_debug_tsort_log(ctypes.c_void_p(buf.ctypes.data), 16384)          # (drops what the warm-up launches logged)
seed = g["project"](lean=True)[-1]
g["ops"].isect_tiles_raw(None, None, g["dep"], g["tw"], g["th"], g["CAP"], want_tiles_per_gauss=False, seed=seed, want_tile_ids=False)
n = L.mgs_debug_tsort_log(ctypes.c_void_p(buf.ctypes.data), 16384)
rows = [tuple(int(x) for x in r) for r in buf[:n]]
main = [r for r in rows if r[3] == 0xffff]
units = [r for r in rows if r[3] != 0xffff]
off = g["tl"].tile_offsets.cpu().numpy().astype(np.int64)
ln = off[1:] - off[:-1]
t0 = min(r[4] for r in main)
end = max(r[5] for r in main)
print(f"main kernel: {len(main)} tiles logged, span {(end - t0) / 100:.1f} us; units' kernel: {len(units)} units" + (f", {(min(r[4] for r in units) - t0) / 100:.1f} -> {(max(r[5] for r in units) - t0) / 100:.1f} us" if units else ""))
dur = np.array([(r[5] - r[4]) / 100 for r in main]); st = np.array([(r[4] - t0) / 100 for r in main]); en = np.array([(r[5] - t0) / 100 for r in main])
tl_ = np.array([ln[r[2]] for r in main])
print("workgroup-microseconds in all: %.0f (= %.1f us x 256 CUs x %.1f resident)" % (dur.sum(), (end - t0) / 100, dur.sum() / ((end - t0) / 100) / 256))
for lo, hi in ((0, 1), (1, 128), (128, 512), (512, 1024), (1024, 2049), (2049, 6144), (6144, 1 << 30)):
    m = (tl_ >= lo) & (tl_ < hi)
    if m.any():
        print(f"  lists of {lo:5d} .. {min(hi, int(tl_.max()) + 1) - 1:5d} entries: {int(m.sum()):5d} tiles, {int(tl_[m].sum()):8d} entries, duration median {np.median(dur[m]):6.1f} max {dur[m].max():6.1f} us, sum {dur[m].sum():8.0f}")
for q in (0.5, 0.9, 0.99, 1.0):
    print(f"  {q:.2f} of the tiles have ended by {np.quantile(en, q):6.1f} us, started by {np.quantile(st, q):6.1f} us")
tf = np.array([(r[1] - r[4]) / 100 for r in main])          # the group filter's share of the tile's time
seg = np.array([int(off[min(4 * (r[2] // 4) + 4, len(off) - 1)] - off[4 * (r[2] // 4)]) for r in main])
print("group filter: median %.1f us, p99 %.1f, max %.1f; sum %.0f of %.0f workgroup-microseconds" % (np.median(tf), np.quantile(tf, 0.99), tf.max(), tf.sum(), dur.sum()))
for lo, hi in ((0, 2048), (2048, 4096), (4096, 8192), (8192, 16384), (16384, 1 << 30)):
    m = (seg >= lo) & (seg < hi)
    if m.any():
        print(f"  segments of {lo:6d} .. {min(hi, int(seg.max()) + 1) - 1:6d} entries: {int(m.sum()):5d} tiles, filter median {np.median(tf[m]):6.1f} max {tf[m].max():6.1f} us, the rest median {np.median((dur - tf)[m]):6.1f} max {(dur - tf)[m].max():6.1f} us")
print("the twelve tiles that end last:")
for i in np.argsort(-en)[:12]:
    print(f"  wg {main[i][0]:5d} tile {main[i][2]:5d} ({tl_[i]:5d} entries, segment {seg[i]:6d}): {st[i]:6.1f} -> {en[i]:6.1f} us, filter {tf[i]:6.1f}")
# resident workgroups over time
ts = np.linspace(0, (end - t0) / 100, 13)[:-1]
print("resident workgroups at", " ".join(f"{t:5.0f}" for t in ts), "us")
print("                      ", " ".join(f"{int(((st <= t) & (en > t)).sum()):5d}" for t in ts))
