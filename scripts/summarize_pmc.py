"""profiles/pmc_traffic.json (scripts/pmc_traffic.sh) -> trimmed json + profiles/r6/06_pmc_counters.md"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
r = json.load(open(path))
keep = {}
for k, v in r["raw"].items():
    st, name = k.split(":", 1)
    if (st == "raster_inf" and "raster_fwd_kernel<4, false, false>" in name) or (st == "raster_inf_q" and "raster_fwd_q" in name) \
            or (st == "raster_bwd_split" and ("raster_bwd" in name or "reduce_rec" in name or "unit_table" in name)) or (st == "project" and "project_color" in name) \
            or (st == "binning" and "raster" not in name and "project" not in name):
        keep[k] = v
r["raw"] = keep
json.dump(r, open(path, "w"), indent=1)
lines = ["# PMC counters per launch at configs[1] (1 M Gaussians, SH 3, 1920x1080, tight lists, 4 channels)", "",
         "rocprofv3 --kernel-trace --pmc, one pass per counter group (FETCH_SIZE, WRITE_SIZE, SQ_* + GRBM_GUI_ACTIVE), scripts/pmc_traffic.sh; averages over launches 2..n of scripts/run_stage.py.",
         "HBM-side traffic = 2 x FETCH_SIZE (gfx950 tallies 128-byte requests at 64 B, MI355X_MICROARCH.md) + WRITE_SIZE, KiB -> bytes.  build stamp " + r["stamp"][:16], "",
         "| stage : kernel | FETCH KiB | WRITE KiB | traffic MB | VALU M | SALU M | LDS M | wave quad-cycles M | wait-inst M | GUI_ACTIVE k (sum of 8 XCDs) | SIMD-cycles per VALU |",
         "|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|"]
for k, v in keep.items():
    f, w = v["FETCH_SIZE_KiB"] or 0, v["WRITE_SIZE_KiB"] or 0
    g = lambda x: ("%.2f" % (v[x] / 1e6)) if v.get(x) is not None else "-"
    gui, valu = (v.get("GRBM_GUI_ACTIVE") or 0), (v.get("SQ_INSTS_VALU") or 0)
    cpi = ("%.2f" % (gui / 8 * 1024 / valu)) if valu else "-"
    lines.append("| %s | %.0f | %.0f | %.1f | %s | %s | %s | %s | %s | %.0f | %s |" % (
        k[:80].replace("mgs::(anonymous namespace)::", "").replace("void ", ""), f, w, (2 * f + w) * 1024 / 1e6, g("SQ_INSTS_VALU"),
        g("SQ_INSTS_SALU"), g("SQ_INSTS_LDS"), g("SQ_WAVE_CYCLES"), g("SQ_WAIT_INST_ANY"), gui / 1e3, cpi))
open(os.path.join(ROOT, "profiles", "r6", "06_pmc_counters.md"), "w").write("\n".join(lines) + "\n")
sys.path.insert(0, ROOT)
from robosimgs_amd.csrc import build
print("\n".join(l[:160] for l in lines[7:12]))
print("stamp matches the tree:", build.current_stamp() == r["stamp"])
