"""Build + ctypes binding of oracle/gs_cpu.cpp (TEST INFRASTRUCTURE ONLY; see its header).
Used by tests/ as the large-scene checker and by bench.py's cpu_baseline leg."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "gs_cpu.cpp")
LIB = os.path.join(HERE, "_build", "libgs_cpu.so")


# The same source a second time for bench.py's cpu_baseline leg ONLY: AVX2 + FMA code (x86-64-v3 runs on every host a GPU box
# can have; -march=native would tie the .so, which is built in the dev container, to this container's CPU).  The CHECKER
# stays the plain build above it: its fp64 sums must not depend on what the compiler contracts into FMAs on which host.
LIB_V3 = os.path.join(HERE, "_build", "libgs_cpu_v3.so")
BASE_FLAGS = ["-O3", "-std=c++17", "-fopenmp", "-shared", "-fPIC"]
V3_FLAGS = BASE_FLAGS + ["-march=x86-64-v3"]


def build(force: bool = False) -> str:
    for lib_path, flags in ((LIB, BASE_FLAGS), (LIB_V3, V3_FLAGS)):
        if force or not os.path.exists(lib_path) or os.path.getmtime(lib_path) < os.path.getmtime(SRC):
            os.makedirs(os.path.dirname(lib_path), exist_ok=True)
            subprocess.run(["g++", *flags, SRC, "-o", lib_path], check=True)
    return LIB


_lib = None
_lib_v3 = None


def _bind(path):
    L = ctypes.CDLL(path)
    L.gs_cpu_render.restype = ctypes.c_longlong
    L.gs_cpu_render_f64.restype = ctypes.c_longlong
    L.gs_cpu_blend_f64.restype = ctypes.c_longlong
    L.gs_cpu_blend_f32.restype = ctypes.c_longlong
    L.gs_cpu_max_threads.restype = ctypes.c_int
    return L


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = _bind(LIB)
    return _lib


def host_has_v3() -> bool:
    """AVX2 + FMA + BMI2 on this host (what -march=x86-64-v3 code needs)."""
    try:
        with open("/proc/cpuinfo") as f:
            flags = next(l for l in f if l.startswith("flags")).split()
        return all(x in flags for x in ("avx2", "fma", "bmi2", "movbe", "f16c", "abm"))
    except Exception:
        return False


def baseline_lib():
    """(library, flags it was built with) for the timed CPU baseline: the x86-64-v3 build where the host can run it."""
    global _lib_v3
    if not host_has_v3():
        return lib(), BASE_FLAGS
    if _lib_v3 is None:
        build()
        _lib_v3 = _bind(LIB_V3)
    return _lib_v3, V3_FLAGS


def max_threads() -> int:
    return lib().gs_cpu_max_threads()


def render(means, quats, scales, opacities, sh_coeffs, viewmat, K, width, height, sh_degree,
           with_depth=False, background=None, eps2d=0.3, near_plane=0.01, far_plane=1e10,
           radius_clip=0.0, n_threads=0, library=None):
    """Forward frame on the host.  Returns (render[H,W,ch], alpha[H,W], info dict).  library: baseline_lib()[0] for the
    timed baseline (default: the checker's build)."""
    f = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    means, quats, scales, opacities, sh = f(means), f(quats), f(scales), f(opacities), f(sh_coeffs)
    vm, Km = f(viewmat), f(K)
    n, ch = means.shape[0], 4 if with_depth else 3
    out = np.empty((height, width, ch), np.float32)
    alpha = np.empty((height, width), np.float32)
    counters = np.zeros(2, np.int64)
    bg = f(background) if background is not None else None
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p) if a is not None else None
    cf = ctypes.c_float
    n_isect = (library or lib()).gs_cpu_render(n, p(means), p(quats), p(scales), p(opacities), int(sh_degree),
                                  sh.shape[1], p(sh), p(vm), p(Km), int(width), int(height),
                                  cf(eps2d), cf(near_plane), cf(far_plane), cf(radius_clip), ch,
                                  p(bg), int(n_threads), p(out), p(alpha), p(counters))
    return out, alpha, {"n_isect": int(n_isect), "n_vis": int(counters[0]),
                        "pair_evals": int(counters[1])}


def render_f64(means, quats, scales, opacities, sh_coeffs, viewmat, K, width, height, sh_degree,
               with_depth=False, background=None, eps2d=0.3, near_plane=0.01, far_plane=1e10,
               radius_clip=0.0, n_threads=0, margins=True, v_render=None, v_alpha=None,
               want_projected=False, flip_eps=None, want_touched=False, want_budget=False,
               thresholds=None, radius_rule="classic"):
    """The frame in fp64 arithmetic on the fp32 inputs the GPU gets (the full-size reference
    answer).  Returns (render[H,W,ch] f32, alpha[H,W] f32, info).  info carries
      margins [4,H,W], edge_mask [H,W] bool, n_edge_gaussians        (margins=True; feed
          oracle.gs_oracle_np.explained_pixels)
      flip_weight [H,W], feat_max [ch]   (margins=True and flip_eps = an EPS_* dict of gs_oracle_np: what the
          decisions within eps of flipping are worth; feed oracle.gs_oracle_np.check_frame)
      touched [N] bool   (want_touched, with margins and flip_eps): Gaussians that reach alpha >= 0.5/255 at a
          could-flip pixel -- the only rows of a gradient that may differ from the oracle's by more than rounding
      budget [N,4] f64   (want_budget, with want_touched and v_render): per Gaussian, how far its rows of the blend's
          gradient (means2d, conics, feats, opacity -- one bound per row, valid for each component) can move when the
          near-flip decisions at the could-flip pixels it reaches go the other way (gs_cpu.cpp: Extras::budget)
      g_means2d [N,2], g_conics [N,3], g_feats [N,ch], g_opacities [N]  f64: the blend's backward
          (A.2 step 10) for upstream v_render [H,W,ch] / v_alpha [H,W]
      means2d, conics, feats, radii as projected by the oracle          (want_projected=True).
    radius_rule: "classic" (A.2 step 5) or "opacity_aware" (per-axis extents; info["radii"] then holds
    the x extents -- positive exactly for the visible Gaussians).
    thresholds = (a, t): the blend tests alpha >= a / 255 and stops at T' <= t * 1e-4 (default 1, 1) -- a test moves
    them by less than the gate's eps to produce real flips and nothing else."""
    f = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    means, quats, scales, opacities, sh = f(means), f(quats), f(scales), f(opacities), f(sh_coeffs)
    vm, Km = f(viewmat), f(K)
    n, ch = means.shape[0], 4 if with_depth else 3
    out = np.empty((height, width, ch), np.float32)
    alpha = np.empty((height, width), np.float32)
    counters = np.zeros(2, np.int64)
    bg = f(background) if background is not None else None
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p) if a is not None else None
    cf = ctypes.c_float
    marg = np.empty((4, height, width), np.float32) if margins else None
    edge = np.empty((height, width), np.uint8) if margins else None
    n_edge = np.zeros(1, np.int64)
    fw = np.empty((height, width), np.float32) if (margins and flip_eps is not None) else None
    fe = (np.array([flip_eps["alpha"], flip_eps["T"], flip_eps["sigma"], flip_eps.get("depth", 0.0)], np.float32)
          if fw is not None else None)
    want_projected = want_projected or fw is not None        # feat_max comes from the projected features
    tch = np.zeros(n, np.uint8) if ((want_touched or want_budget) and fw is not None) else None
    if want_budget and (tch is None or v_render is None):
        raise ValueError("want_budget needs margins, flip_eps and v_render")
    bud = np.zeros((n, 4), np.float64) if want_budget else None
    bwd = v_render is not None
    vr = f(v_render).reshape(height, width, ch) if bwd else None
    va = f(v_alpha if v_alpha is not None else np.zeros((height, width))).reshape(height, width) if bwd else None
    gm, gc, gf, go = ((np.empty((n, 2)), np.empty((n, 3)), np.empty((n, ch)), np.empty(n)) if bwd
                      else (None, None, None, None))
    om, oc, of_, orad = ((np.empty((n, 2)), np.empty((n, 3)), np.empty((n, ch)), np.empty(n, np.int32))
                         if want_projected else (None, None, None, None))
    n_isect = lib().gs_cpu_render_f64(
        n, p(means), p(quats), p(scales), p(opacities), int(sh_degree), sh.shape[1], p(sh), p(vm),
        p(Km), int(width), int(height), cf(eps2d), cf(near_plane), cf(far_plane), cf(radius_clip),
        ch, p(bg), int(n_threads), p(out), p(alpha), p(counters), p(marg), p(edge), p(n_edge),
        p(vr), p(va), p(gm), p(gc), p(gf), p(go), p(om), p(oc), p(of_), p(orad), p(fw), p(fe), p(tch), p(bud),
        p(np.array(thresholds, np.float32)) if thresholds is not None else None,
        {"classic": 0, "opacity_aware": 1}[radius_rule])
    info = {"n_isect": int(n_isect), "n_vis": int(counters[0]), "pair_evals": int(counters[1])}
    if margins:
        info.update(margins=marg, edge_mask=edge.astype(bool), n_edge_gaussians=int(n_edge[0]))
        if fw is not None:
            vis = orad > 0
            if tch is not None:
                info["touched"] = tch.astype(bool)
            if bud is not None:
                info["budget"] = bud
            info.update(flip_weight=fw, feat_max=(np.abs(of_[vis]).max(axis=0) if vis.any() else np.zeros(ch)))
    if bwd:
        info.update(g_means2d=gm, g_conics=gc, g_feats=gf, g_opacities=go)
    if want_projected:
        info.update(means2d=om, conics=oc, feats=of_, radii=orad)
    return out, alpha, info


def blend_f64(means2d, conics, opacities, feats, flatten_ids, tile_offsets, width, height, depths=None, background=None,
              n_threads=0, margins=True, flip_eps=None, v_render=None, v_alpha=None, want_budget=False, thresholds=None):
    """STAGE-ISOLATED blend: SURVEY.md A.2 steps 9-10 in fp64 on GIVEN fp32 projected quantities and depth-ordered tile
    lists -- e.g. the GPU's own means2d / conics / opacities / feats / flatten_ids / tile_offsets, read back -- so that a
    blend kernel can be held to zero unexplained pixels whatever rounding the projection did upstream.
    feats [N,ch] (ch = 3 or 4; the fourth channel is the plain depth sum, not divided), flatten_ids [n_isect] int32,
    tile_offsets [tiles + 1] int32 (tiles of 16 px, row-major).  depths [N]: only for the depth-tie margin (None: no such
    margin; with given lists the order is given).  Returns (render [H,W,ch] f32, alpha [H,W] f32, info) with info as
    render_f64's: margins [4,H,W], flip_weight [H,W] + feat_max (flip_eps), edge_mask (all False: membership is given),
    noise_weight [H,W] (what sigma's rounding alone can move the pixel by, as a blend weight: gs_cpu.cpp Extras),
    g_means2d / g_conics / g_feats / g_opacities (v_render), touched / budget (want_budget)."""
    f = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    m2d, con, opa, fe = f(means2d), f(conics), f(opacities), f(feats)
    n, ch = m2d.shape[0], fe.shape[1]
    ids = np.ascontiguousarray(flatten_ids, dtype=np.int32)
    offs = np.ascontiguousarray(tile_offsets, dtype=np.int32)
    tw, th = -(-width // 16), -(-height // 16)
    if offs.shape[0] != tw * th + 1 or int(offs[-1]) > ids.shape[0]:
        raise ValueError("tile_offsets must hold tiles + 1 entries and end inside flatten_ids")
    if ids.size and (ids[:int(offs[-1])].min() < 0 or ids[:int(offs[-1])].max() >= n):
        raise ValueError("flatten_ids outside 0..N-1")
    dep = f(depths) if depths is not None else None
    out = np.empty((height, width, ch), np.float32)
    alpha = np.empty((height, width), np.float32)
    counters = np.zeros(2, np.int64)
    bg = f(background) if background is not None else None
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p) if a is not None else None
    marg = np.empty((4, height, width), np.float32) if margins else None
    fw = np.empty((height, width), np.float32) if (margins and flip_eps is not None) else None
    nw = np.empty((height, width), np.float32) if margins else None
    fe_arr = (np.array([flip_eps["alpha"], flip_eps["T"], flip_eps["sigma"], flip_eps.get("depth", 0.0) if dep is not None else 0.0],
                       np.float32) if fw is not None else None)
    edge = np.zeros((height, width), np.uint8) if (want_budget and fw is not None) else None
    tch = np.zeros(n, np.uint8) if edge is not None else None
    bwd = v_render is not None
    if want_budget and (tch is None or not bwd):
        raise ValueError("want_budget needs margins, flip_eps and v_render")
    bud = np.zeros((n, 4), np.float64) if want_budget else None
    vr = f(v_render).reshape(height, width, ch) if bwd else None
    va = f(v_alpha if v_alpha is not None else np.zeros((height, width))).reshape(height, width) if bwd else None
    gm, gc, gf, go = ((np.empty((n, 2)), np.empty((n, 3)), np.empty((n, ch)), np.empty(n)) if bwd else (None, None, None, None))
    n_isect = lib().gs_cpu_blend_f64(
        n, p(m2d), p(con), p(opa), p(fe), p(dep), p(ids), p(offs), int(width), int(height), ch, p(bg), int(n_threads),
        p(out), p(alpha), p(counters), p(marg), p(vr), p(va), p(gm), p(gc), p(gf), p(go), p(fw), p(fe_arr), p(edge), p(tch), p(bud),
        p(np.array(thresholds, np.float32)) if thresholds is not None else None, p(nw))
    info = {"n_isect": int(n_isect), "pair_evals": int(counters[1])}
    if margins:
        info.update(margins=marg, edge_mask=np.zeros((height, width), bool), noise_weight=nw)
        if fw is not None:
            listed = np.zeros(n, bool)
            listed[ids[:int(offs[-1])]] = True
            info.update(flip_weight=fw, feat_max=(np.abs(fe[listed]).max(axis=0) if listed.any() else np.zeros(ch)))
            if tch is not None:
                info["touched"] = tch.astype(bool)
            if bud is not None:
                info["budget"] = bud
    if bwd:
        info.update(g_means2d=gm, g_conics=gc, g_feats=gf, g_opacities=go)
    return out, alpha, info


def blend_f32(means2d, conics, opacities, feats, flatten_ids, tile_offsets, width, height, background=None, n_threads=0):
    """blend_f64's inputs through the port's FLOAT instantiation (the reference's formulas in plain fp32): (render, alpha)."""
    f = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    m2d, con, opa, fe = f(means2d), f(conics), f(opacities), f(feats)
    ids = np.ascontiguousarray(flatten_ids, dtype=np.int32)
    offs = np.ascontiguousarray(tile_offsets, dtype=np.int32)
    ch = fe.shape[1]
    out = np.empty((height, width, ch), np.float32)
    alpha = np.empty((height, width), np.float32)
    bg = f(background) if background is not None else None
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p) if a is not None else None
    lib().gs_cpu_blend_f32(m2d.shape[0], p(m2d), p(con), p(opa), p(fe), p(ids), p(offs), int(width), int(height), ch, p(bg),
                           int(n_threads), p(out), p(alpha))
    return out, alpha
