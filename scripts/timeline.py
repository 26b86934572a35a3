"""Timeline analysis of a rocprofv3 kernel trace of bench.py: how much of the steady state has a
raster kernel running, and what runs in the gaps.  usage: timeline.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    name = r["Kernel_Name"]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    kind = "raster" if "raster_fwd_kernel<3, false>" in name else ("proj" if "project_color_fwd" in name else ("bin" if "mgs::" in name else "other"))
    ev.append((s, e, kind, name))
ev.sort()
# steady state: the middle 60 % of the inference rasters
ras = [x for x in ev if x[2] == "raster" and x[1] - x[0] > 100_000]
lo, hi = ras[len(ras) // 5][0], ras[4 * len(ras) // 5][1]
pts = []
for s, e, k, _ in ev:
    if e < lo or s > hi: continue
    pts.append((max(s, lo), 1, k)); pts.append((min(e, hi), -1, k))
pts.sort()
active = collections.Counter(); t_prev = lo
hist = collections.Counter(); gap_kinds = collections.Counter()
for tt, d, k in pts:
    dt = tt - t_prev
    if dt > 0:
        hist[active["raster"]] += dt
        if active["raster"] == 0:
            key = ("proj" if active["proj"] else "") + ("+bin" if active["bin"] else "") or "idle"
            gap_kinds[key] += dt
    active[k] += d; t_prev = tt
tot = hi - lo
n_frames = sum(1 for x in ras if lo <= x[0] <= hi)
print(f"steady window {tot/1e3:.0f} us, {n_frames} frames, {tot/1e3/n_frames:.1f} us/frame")
for k in sorted(hist): print(f"  {k} raster kernels running: {100*hist[k]/tot:.1f} % of the time ({hist[k]/1e3/n_frames:.1f} us/frame)")
for k, v in gap_kinds.most_common(): print(f"  no raster, running [{k}]: {v/1e3/n_frames:.1f} us/frame")
