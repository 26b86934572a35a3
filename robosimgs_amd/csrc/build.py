"""Build libmgs.so (HIP, gfx950 only) in-tree with hipcc.  No cmake, no JIT cache: the .so
sits next to the sources so it travels with the repo snapshot to the GPU box."""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["api.hip", "projection.hip", "sort.hip", "binning.hip", "tile_sort.hip", "raster_fwd.hip",
           "raster_bwd.hip", "backward.hip", "composite.hip", "points.hip", "loss.hip", "transform.hip", "frame.hip"]
HEADERS = ["mgs_common.h", "mgs_math.h", "raster_common.h", "sh_staging.h", "tile_rect.h", "tile_order.h",
           "../../include/mgs.h"]
LIB = os.path.join(HERE, "libmgs.so")
# The same sources with -DMGS_DEBUG_HOOKS: the process-global test / measurement knobs (mgs_debug_set_*) exist in this
# build only; tests and A/B scripts load it (robosimgs_amd._lib.use_debug_lib), the product never does.
DEBUG_LIB = os.path.join(HERE, "libmgs_debug.so")
HOOK_SOURCES = ["raster_fwd.hip", "sort.hip"]          # the translation units MGS_DEBUG_HOOKS changes
OBJ_DIR = os.path.join(HERE, "build")
# -fno-slp-vectorize: hipcc's SLP pass packs adjacent f32 adds/multiplies into v_pk_*_f32, which on
# gfx950 buys no throughput and costs v_mov shuffles: raster fwd 299 -> 260 us, bwd 920 -> 725 us.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
         "-fno-slp-vectorize", "-Wall", "-Wno-unused-function"] + os.environ.get("MGS_EXTRA_FLAGS", "").split()


# per-source additions (A/B-measured, see profiles/r1/05); overridable for experiments
PER_SOURCE_FLAGS = {
    "raster_bwd.hip": os.environ.get("MGS_RASTER_BWD_FLAGS", "").split(),
    "raster_fwd.hip": os.environ.get("MGS_RASTER_FWD_FLAGS", "").split(),
    "tile_sort.hip": os.environ.get("MGS_TILE_SORT_FLAGS", "").split(),
    "binning.hip": os.environ.get("MGS_BINNING_FLAGS", "").split(),
}


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found; libmgs.so cannot be built")


def _stamp() -> str:
    h = hashlib.sha256((" ".join(FLAGS) + repr(sorted(PER_SOURCE_FLAGS.items()))).encode())
    for f in SOURCES + HEADERS:
        p = os.path.join(HERE, f)
        if os.path.exists(p):
            with open(p, "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()


def _code_only(text: str) -> str:
    """C / C++ source with comments removed and runs of whitespace collapsed (string and character
    literals are kept as they are): two sources that differ in comments or layout only give the same text."""
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if c in "\"'":                                   # literal: copy to the closing quote
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1])
            i = j + 1
        elif text.startswith("//", i):
            j = text.find("\n", i)
            i = n if j < 0 else j
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            i = n if j < 0 else j + 2
            out.append(" ")
        else:
            out.append(c)
            i += 1
    return " ".join("".join(out).split())


def current_stamp() -> str:
    """Hash of the CODE libmgs.so is built from -- sources and headers without comments and layout, plus the
    flags (bench.py keys measured PMC traffic to it, so a number taken on another build is never printed,
    while an edit of a comment does not orphan the measurement)."""
    h = hashlib.sha256((" ".join(FLAGS) + repr(sorted(PER_SOURCE_FLAGS.items()))).encode())
    for f in SOURCES + HEADERS:
        p = os.path.join(HERE, f)
        if os.path.exists(p):
            with open(p, "r", encoding="utf-8", errors="replace") as fh:
                h.update(_code_only(fh.read()).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = False) -> str:
    stamp_file = os.path.join(OBJ_DIR, "stamp")
    stamp = _stamp()
    if (not force and os.path.exists(LIB) and os.path.exists(DEBUG_LIB) and os.path.exists(stamp_file)
            and open(stamp_file).read() == stamp):
        return LIB
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = _hipcc()
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(HERE, s))]

    def compile_one(job) -> str:
        src, hooks = job
        obj = os.path.join(OBJ_DIR, src.replace(".hip", ".dbg.o" if hooks else ".o"))
        cmd = [hipcc, *FLAGS, *PER_SOURCE_FLAGS.get(src, []), *(["-DMGS_DEBUG_HOOKS"] if hooks else []), "-c",
               os.path.join(HERE, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    jobs = [(s_, False) for s_ in srcs] + [(s_, True) for s_ in srcs if s_ in HOOK_SOURCES]
    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = dict(zip(jobs, ex.map(compile_one, jobs)))
    for lib, hooks in ((LIB, False), (DEBUG_LIB, True)):
        use = [objs[(s_, hooks and s_ in HOOK_SOURCES)] for s_ in srcs]
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *use, "-o", lib]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
    with open(stamp_file, "w") as f:
        f.write(stamp)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
