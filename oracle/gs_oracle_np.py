"""NumPy oracle for the 3DGS background-render hot path (TEST INFRASTRUCTURE ONLY).

PARITY UNPINNED.  The reference checkout (/root/reference, Maxwell-Zhao/RoboSimGS @
2025-10-17) contains no Gaussian-splatting renderer: README.md:75 delegates scene
reconstruction to Nerfstudio, README.md:29 / :84-85 say the stage that renders the
exported .ply is unreleased, and nerfstudio / gsplat are neither vendored nor pinned
(Articulation/requirements.txt:1-18).  This file therefore restates the *published*
algorithm of the third-party dependency the reference names (nerfstudio `splatfacto` ->
gsplat 1.x "classic" rasterisation) and is validated
independently of itself by closed-form cases in tests/test_oracle_closed_form.py.  The
only reference-pinned piece is the camera convention, `viewmat_from_c2w_opengl`, which
follows Articulation/utils/nerf2physic_utils.py:10-23 (`project_3d_to_2d`: negate cam-Y
and cam-Z of w2c) and is checked against vectors generated from that function
(tests/golden/make_camera_golden.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  The product package (robosimgs_amd/) never does.

All functions take a `dtype` (np.float64 for the reference answer, np.float32 to mimic
device rounding op-by-op without FMA contraction).
"""
from __future__ import annotations

import numpy as np

ALPHA_MIN = 1.0 / 255.0
ALPHA_MAX = 0.999
T_STOP = 1e-4

SH_C0 = 0.2820947917738781
SH_C1 = 0.48860251190292
SH_C2 = (0.5462742152960395, -1.092548430592079, 0.9461746957575601,
         0.3153915652525201)
SH_C3 = (-0.5900435899266435, 1.445305721320277, -2.285228997322329,
         0.4570457994644658, 1.865881662950577, 1.119528997770346)


# --------------------------------------------------------------------------------------
# camera convention (the only reference-pinned part)
# --------------------------------------------------------------------------------------
def viewmat_from_c2w_opengl(c2w):
    """OpenGL camera-to-world (nerfstudio `transform_matrix`) -> OpenCV world-to-camera.

    Follows /root/reference/Articulation/utils/nerf2physic_utils.py:14-16: apply
    inv(c2w) then negate camera Y and Z.  Equivalent to nerfstudio's get_viewmat:
    R' = R_c2w * diag(1,-1,-1); viewmat = [[R'^T, -R'^T t],[0, 1]].
    """
    c2w = np.asarray(c2w, dtype=np.float64)
    w2c = np.linalg.inv(c2w)
    flip = np.diag([1.0, -1.0, -1.0, 1.0])
    return flip @ w2c


# --------------------------------------------------------------------------------------
# A.2 step 1-5: projection
# --------------------------------------------------------------------------------------
def quat_to_rotmat(quats, dtype=np.float64):
    """wxyz quaternion (un-normalised) -> rotation matrix [N,3,3].  A.2 step 1."""
    q = np.asarray(quats, dtype=dtype)
    inv_norm = dtype(1.0) / np.sqrt(q[:, 0] * q[:, 0] + q[:, 1] * q[:, 1]
                                    + q[:, 2] * q[:, 2] + q[:, 3] * q[:, 3])
    w, x, y, z = (q[:, i] * inv_norm for i in range(4))
    x2, y2, z2 = x * x, y * y, z * z
    xy, xz, yz = x * y, x * z, y * z
    wx, wy, wz = w * x, w * y, w * z
    one, two = dtype(1.0), dtype(2.0)
    R = np.empty((q.shape[0], 3, 3), dtype=dtype)
    R[:, 0, 0] = one - two * (y2 + z2)
    R[:, 0, 1] = two * (xy - wz)
    R[:, 0, 2] = two * (xz + wy)
    R[:, 1, 0] = two * (xy + wz)
    R[:, 1, 1] = one - two * (x2 + z2)
    R[:, 1, 2] = two * (yz - wx)
    R[:, 2, 0] = two * (xz - wy)
    R[:, 2, 1] = two * (yz + wx)
    R[:, 2, 2] = one - two * (x2 + y2)
    return R


def covar_world(quats, scales, dtype=np.float64):
    """Sigma = (R S)(R S)^T.  A.2 step 1."""
    R = quat_to_rotmat(quats, dtype)
    M = R * np.asarray(scales, dtype=dtype)[:, None, :]
    return M @ np.swapaxes(M, 1, 2)


EXTENT_MAX = 3.33     # gsplat >= 1.5: the per-axis extent is capped at 3.33 sigma


def project(means, quats, scales, viewmat, K, width, height, eps2d=0.3,
            near_plane=0.01, far_plane=1e10, radius_clip=0.0, dtype=np.float64,
            radius_rule="classic", opacities=None, antialiased=False):
    """A.2 steps 1-5 for one camera.

    Returns dict(radii[N] int32, means2d[N,2], depths[N], conics[N,3],
    compensations[N]); culled Gaussians have radii 0 and zeros elsewhere.

    radius_rule "classic": A.2 step 5 (gsplat 1.4).  "opacity_aware": SURVEY.md A.4 (gsplat >= 1.5) -- radii[N,2],
    per-axis extents ceil(e sqrt(Sigma_xx)), ceil(e sqrt(Sigma_yy)) of the blurred 2-D covariance with
    e = min(3.33, sqrt(2 ln(255 o))), o = opacity (x compensation when antialiased; e = 3.33 without opacities);
    culled: o < 1/255, both extents <= radius_clip, the box mean +- extents wholly off screen.
    """
    means = np.asarray(means, dtype=dtype)
    viewmat = np.asarray(viewmat, dtype=dtype)
    K = np.asarray(K, dtype=dtype)
    N = means.shape[0]
    Rcw, tcw = viewmat[:3, :3], viewmat[:3, 3]
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    W, H = dtype(width), dtype(height)

    pc = means @ Rcw.T + tcw                                     # step 2
    x, y, z = pc[:, 0], pc[:, 1], pc[:, 2]
    valid = (z >= dtype(near_plane)) & (z <= dtype(far_plane))
    zs = np.where(valid, z, dtype(1.0))

    cov = covar_world(quats, scales, dtype)
    cov_c = Rcw[None] @ cov @ Rcw.T[None]

    tanx, tany = dtype(0.5) * W / fx, dtype(0.5) * H / fy        # step 3
    k03 = dtype(0.3)
    lim_xp, lim_xn = (W - cx) / fx + k03 * tanx, cx / fx + k03 * tanx
    lim_yp, lim_yn = (H - cy) / fy + k03 * tany, cy / fy + k03 * tany
    rz = dtype(1.0) / zs
    rz2 = rz * rz
    tx = zs * np.minimum(lim_xp, np.maximum(-lim_xn, x * rz))
    ty = zs * np.minimum(lim_yp, np.maximum(-lim_yn, y * rz))
    J = np.zeros((N, 2, 3), dtype=dtype)
    J[:, 0, 0] = fx * rz
    J[:, 0, 2] = -fx * tx * rz2
    J[:, 1, 1] = fy * rz
    J[:, 1, 2] = -fy * ty * rz2
    cov2 = J @ cov_c @ np.swapaxes(J, 1, 2)
    mu = np.stack([fx * x * rz + cx, fy * y * rz + cy], axis=-1)

    a, b, c = cov2[:, 0, 0], cov2[:, 0, 1], cov2[:, 1, 1]       # step 4
    det0 = a * c - b * b
    a = a + dtype(eps2d)
    c = c + dtype(eps2d)
    det = a * c - b * b
    valid &= det > 0
    dets = np.where(det > 0, det, dtype(1.0))
    comp = np.sqrt(np.maximum(dtype(0.0), det0 / dets))
    conic = np.stack([c / dets, -b / dets, a / dets], axis=-1)

    m = dtype(0.5) * (a + c)                                     # step 5
    lam = m + np.sqrt(np.maximum(dtype(0.01), m * m - dets))
    extra = {}
    if radius_rule == "classic":
        radius = np.ceil(dtype(3.0) * np.sqrt(lam))
        radius_y = radius
        valid &= radius > dtype(radius_clip)
        extra["extent_xy"] = np.stack([dtype(3.0) * np.sqrt(lam)] * 2, axis=-1)
    elif radius_rule == "opacity_aware":
        ext = np.full(N, dtype(EXTENT_MAX), dtype=dtype)
        op_ok = np.ones(N, dtype=bool)
        if opacities is not None:
            op = np.asarray(opacities, dtype=dtype)
            if antialiased:
                op = op * comp
            op_ok = op >= dtype(1.0) / dtype(255.0)
            with np.errstate(invalid="ignore", divide="ignore"):
                lim = np.sqrt(dtype(2.0) * np.log(np.where(op_ok, op, dtype(1.0)) * dtype(255.0)))
            ext = np.minimum(ext, lim)
            extra["op_rule"] = op
        valid &= op_ok
        ex, ey = ext * np.sqrt(np.maximum(a, 0)), ext * np.sqrt(np.maximum(c, 0))
        radius, radius_y = np.ceil(ex), np.ceil(ey)
        valid &= (radius > dtype(radius_clip)) | (radius_y > dtype(radius_clip))
        valid &= (radius > 0) & (radius_y > 0)
        extra["extent_xy"] = np.stack([ex, ey], axis=-1)
        extra["extent"] = ext
    else:
        raise ValueError(radius_rule)
    valid &= ~((mu[:, 0] + radius <= 0) | (mu[:, 0] - radius >= W)
               | (mu[:, 1] + radius_y <= 0) | (mu[:, 1] - radius_y >= H))
    if radius_rule == "classic":
        radii = np.where(valid, radius, 0).astype(np.int32)
    else:
        radii = np.where(valid[:, None], np.stack([radius, radius_y], axis=-1), 0).astype(np.int32)

    zok = (z >= dtype(near_plane)) & (z <= dtype(far_plane))
    out = {
        # un-masked intermediates (gaussian_edge_mask): lambda_1, raw means / depths / conics
        "lam": lam, "mu": mu, "z": z, "det_ok": det > 0, "conics_all": conic, "z_ok": zok, **extra,
        "radii": radii,
        "means2d": np.where(valid[:, None], mu, 0),
        "depths": np.where(valid, z, 0),
        "conics": np.where(valid[:, None], conic, 0),
        "compensations": np.where(valid, comp, 0),
    }
    return out


# --------------------------------------------------------------------------------------
# A.2 step 6: spherical harmonics
# --------------------------------------------------------------------------------------
def sh_basis(degree, dirs, dtype=np.float64):
    """Real SH basis Y_k(d) for unit dirs [N,3] -> [N,(degree+1)^2].  A.2 step 6."""
    d = np.asarray(dirs, dtype=dtype)
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    K = (degree + 1) ** 2
    Y = np.empty((d.shape[0], K), dtype=dtype)
    Y[:, 0] = dtype(SH_C0)
    if degree >= 1:
        Y[:, 1] = -dtype(SH_C1) * y
        Y[:, 2] = dtype(SH_C1) * z
        Y[:, 3] = -dtype(SH_C1) * x
    if degree >= 2:
        z2 = z * z
        fC1 = x * x - y * y
        fS1 = dtype(2.0) * x * y
        t = dtype(SH_C2[1]) * z
        Y[:, 4] = dtype(SH_C2[0]) * fS1
        Y[:, 5] = t * y
        Y[:, 6] = dtype(SH_C2[2]) * z2 - dtype(SH_C2[3])
        Y[:, 7] = t * x
        Y[:, 8] = dtype(SH_C2[0]) * fC1
    if degree >= 3:
        u = dtype(SH_C3[2]) * z2 + dtype(SH_C3[3])
        w = dtype(SH_C3[1]) * z
        fC2 = x * fC1 - y * fS1
        fS2 = x * fS1 + y * fC1
        Y[:, 9] = dtype(SH_C3[0]) * fS2
        Y[:, 10] = w * fS1
        Y[:, 11] = u * y
        Y[:, 12] = z * (dtype(SH_C3[4]) * z2 - dtype(SH_C3[5]))
        Y[:, 13] = u * x
        Y[:, 14] = w * fC1
        Y[:, 15] = dtype(SH_C3[0]) * fC2
    return Y


def sh_colors(degree, means, campos, coeffs, dtype=np.float64):
    """rgb = max(0, 0.5 + sum_k Y_k(normalize(mean - campos)) coeff_k).  coeffs [N,K,3]."""
    dirs = np.asarray(means, dtype=dtype) - np.asarray(campos, dtype=dtype)[None]
    n = np.sqrt((dirs * dirs).sum(-1, keepdims=True))
    dirs = dirs / np.where(n > 0, n, dtype(1.0))
    Y = sh_basis(degree, dirs, dtype)
    K = (degree + 1) ** 2
    col = np.einsum("nk,nkc->nc", Y, np.asarray(coeffs, dtype=dtype)[:, :K, :])
    return np.maximum(col + dtype(0.5), dtype(0.0))


def campos_from_viewmat(viewmat):
    """Camera centre in world space = (viewmat^-1)[:3,3]."""
    v = np.asarray(viewmat, dtype=np.float64)
    return -v[:3, :3].T @ v[:3, 3]


# --------------------------------------------------------------------------------------
# A.2 steps 7-8: tile intersection, sort, ranges
# --------------------------------------------------------------------------------------
def visible(radii):
    """bool [N]: radii [N] > 0, or the x extent of per-axis radii [N,2] > 0 (both are positive or both zero)."""
    r = np.asarray(radii)
    return (r > 0) if r.ndim == 1 else (r[:, 0] > 0)


def tile_rects(means2d, radii, tile_size, tile_w, tile_h, dtype=np.float64):
    """Per-Gaussian tile rectangle [x0,x1) x [y0,y1) (A.2 step 7); radii [N] or per-axis [N,2] (A.4)."""
    mu = np.asarray(means2d, dtype=dtype)
    r = np.asarray(radii).astype(dtype)
    ts = dtype(tile_size)
    trx, try_ = (r / ts, r / ts) if r.ndim == 1 else (r[:, 0] / ts, r[:, 1] / ts)
    txc, tyc = mu[:, 0] / ts, mu[:, 1] / ts
    x0 = np.clip(np.floor(txc - trx), 0, tile_w).astype(np.int64)
    x1 = np.clip(np.ceil(txc + trx), 0, tile_w).astype(np.int64)
    y0 = np.clip(np.floor(tyc - try_), 0, tile_h).astype(np.int64)
    y1 = np.clip(np.ceil(tyc + try_), 0, tile_h).astype(np.int64)
    vis = visible(radii)
    x0, x1, y0, y1 = (np.where(vis, v, 0) for v in (x0, x1, y0, y1))
    return x0, x1, y0, y1


def isect_tiles(means2d, radii, depths, tile_size, tile_w, tile_h, cam=0,
                n_cams=1, dtype=np.float64):
    """A.2 steps 7-8 for one camera: returns (tiles_per_gauss, isect_ids_sorted int64,
    flatten_ids_sorted int32).  Depth bits come from the float32 value of `depths`."""
    x0, x1, y0, y1 = tile_rects(means2d, radii, tile_size, tile_w, tile_h, dtype)
    tpg = ((x1 - x0) * (y1 - y0)).astype(np.int32)
    n_tiles = tile_w * tile_h
    tile_bits = int(np.floor(np.log2(n_tiles))) + 1 if n_tiles > 0 else 0
    dbits = np.asarray(depths, dtype=np.float32).view(np.uint32).astype(np.int64)
    # emit in Gaussian-index order, row-major (y outer, x inner) inside each rectangle
    cnt = tpg.astype(np.int64)
    total = int(cnt.sum())
    vals = np.repeat(np.arange(len(cnt), dtype=np.int64), cnt)
    start = np.cumsum(cnt) - cnt
    local = np.arange(total, dtype=np.int64) - np.repeat(start, cnt)
    wrect = np.maximum(x1 - x0, 1)
    tid = (y0[vals] + local // wrect[vals]) * tile_w + x0[vals] + local % wrect[vals]
    keys = (((np.int64(cam) << tile_bits) | tid) << 32) | dbits[vals]
    order = np.argsort(keys, kind="stable")
    return tpg, keys[order], vals[order].astype(np.int32)


def isect_offsets(isect_ids, n_cams, tile_w, tile_h):
    """First sorted index of each (cam, tile) -> int32 [C, th, tw].  A.2 step 8."""
    n_tiles = tile_w * tile_h
    tile_bits = int(np.floor(np.log2(n_tiles))) + 1
    tid = (np.asarray(isect_ids, dtype=np.int64) >> 32)
    cam = tid >> tile_bits
    flat = cam * n_tiles + (tid & ((1 << tile_bits) - 1))
    offs = np.searchsorted(flat, np.arange(n_cams * n_tiles), side="left")
    return offs.astype(np.int32).reshape(n_cams, tile_h, tile_w)


# --------------------------------------------------------------------------------------
# A.2 step 9: forward blend (literal sequential loop; small cases only)
# --------------------------------------------------------------------------------------
def rasterize(means2d, conics, colors, opacities, flatten_ids, offsets, width, height,
              tile_size=16, background=None, dtype=np.float64, exp=None, margins=False, depths=None,
              flip_eps=None):
    """Per-tile front-to-back alpha compositing for ONE camera.

    colors [N,D]; offsets [th,tw] int32 (first sorted index per tile); the range of
    tile t ends at offsets.flat[t+1] (or n_isect for the last tile).
    Returns (image[H,W,D], alpha[H,W], last_ids[H,W] int32, stats dict).
    `last_ids` is the sorted-list index of the last Gaussian that contributed (0-based,
    global index into flatten_ids); pixels with no contribution hold range_start - 1 ...
    they hold the value `range_start` is NOT touched: we use 0 like an all-zero init.

    margins=True adds stats["margins"], float64 [4,H,W]: per pixel, the smallest RELATIVE distance
    of any decision the blend took for it to the point where that decision flips (see
    `explained_pixels`):
      [0] alpha >= 1/255       |255 alpha - 1|                 over pairs evaluated on an open pixel
      [1] T (1 - alpha) <= 1e-4   |T' / 1e-4 - 1|              over pairs that passed the alpha test
      [2] sigma >= 0           sigma / S, S = 0.5 (|a| dx^2 + |c| dy^2) + |b dx dy|
                                                              (the magnitude sigma's rounding scales with)
      [3] depth order          (z_i - z_prev) / z_i            over consecutive CONTRIBUTORS of the pixel: two
                               Gaussians whose depths are within rounding of a tie may be sorted the other way
                               round by an fp32 projection (needs `depths`; inf without)
    An implementation whose arithmetic differs from this one by a relative eps can take a different
    branch only at a pixel whose margin is below ~eps; everywhere else it must agree to within the
    propagated rounding error.

    flip_eps (an EPS_* dict, with margins=True) adds stats["flip_weight"] [H,W] and stats["t_at_min"] [H,W]:
    what the decisions within flip_eps of flipping are WORTH at that pixel, as a blend weight --
      alpha / sigma toggle of a Gaussian  alpha T       T threshold (stop at this Gaussian or go on)  T
      depth-order swap of two neighbours  alpha_i alpha_j T
    (a toggle also moves every later T by the factor 1 - alpha, so later T thresholds are tested against
    eps_T + the toggled alphas so far).  A pixel that differs from this blend because such a decision went
    the other way can be off by at most that weight times the feature range: `check_frame` enforces it."""
    if exp is None:
        exp = np.exp
    mu = np.asarray(means2d, dtype=dtype)
    con = np.asarray(conics, dtype=dtype)
    col = np.asarray(colors, dtype=dtype)
    opa = np.asarray(opacities, dtype=dtype)
    D = col.shape[1]
    th, tw = offsets.shape
    n_isect = len(flatten_ids)
    flat_offs = np.concatenate([offsets.reshape(-1), [n_isect]]).astype(np.int64)
    img = np.zeros((height, width, D), dtype=dtype)
    alpha_img = np.zeros((height, width), dtype=dtype)
    last = np.zeros((height, width), dtype=np.int32)
    marg = np.full((4, height, width), np.inf) if margins else None
    want_fw = margins and flip_eps is not None
    fw_img = np.zeros((height, width)) if want_fw else None
    tmin_img = np.zeros((height, width)) if want_fw else None
    fe_a, fe_t, fe_s, fe_z = ((flip_eps["alpha"], flip_eps["T"], flip_eps["sigma"], flip_eps.get("depth", 0.0))
                              if want_fw else (0, 0, 0, 0))
    zs = np.asarray(depths, dtype=np.float64) if depths is not None else None
    n_eval = 0
    n_contrib = 0
    half, one = dtype(0.5), dtype(1.0)
    for ty in range(th):
        for tx in range(tw):
            s, e = flat_offs[ty * tw + tx], flat_offs[ty * tw + tx + 1]
            y_lo, y_hi = ty * tile_size, min((ty + 1) * tile_size, height)
            x_lo, x_hi = tx * tile_size, min((tx + 1) * tile_size, width)
            if y_lo >= y_hi or x_lo >= x_hi:
                continue
            py, px = np.meshgrid(np.arange(y_lo, y_hi, dtype=dtype) + half,
                                 np.arange(x_lo, x_hi, dtype=dtype) + half,
                                 indexing="ij")
            T = np.ones_like(px)
            C = np.zeros(px.shape + (D,), dtype=dtype)
            done = np.zeros(px.shape, dtype=bool)
            cur_last = np.zeros(px.shape, dtype=np.int32)
            tm = np.full((4,) + px.shape, np.inf) if margins else None
            z_prev = np.full(px.shape, -1.0)
            fw, loose, wt_prev, t_min, r_acc, sr_acc = (np.zeros(px.shape) for _ in range(6))
            for i in range(s, e):
                if done.all():
                    break
                g = flatten_ids[i]
                dx = mu[g, 0] - px
                dy = mu[g, 1] - py
                sigma = half * (con[g, 0] * dx * dx + con[g, 2] * dy * dy) \
                    + con[g, 1] * dx * dy
                a = np.minimum(dtype(ALPHA_MAX), opa[g] * exp(-sigma))
                ok = (~done) & (sigma >= 0) & (a >= dtype(ALPHA_MIN))
                n_eval += int((~done).sum())
                Tn = T * (one - a)
                if margins:
                    live = ~done
                    S = half * (abs(con[g, 0]) * dx * dx + abs(con[g, 2]) * dy * dy) \
                        + abs(con[g, 1] * dx * dy)
                    # CONDITIONED margins (round 6): how far alpha and T' are from their thresholds, less what ANY fp32
                    # evaluation of this pair can be off by -- sigma is a sum of terms of size S, so d alpha / alpha = d sigma
                    # >= SIGMA_ABS S however carefully it is evaluated (a needle seen 100 px from its mean: S ~ 6,000,
                    # sigma ~ 5), and T' = T (1 - alpha) carries alpha's relative error times r = alpha / (1 - alpha) (an
                    # opaque Gaussian at its centre: r ~ 100-1000).  With eps_alpha = eps_T = eps (both EPS sets):
                    #   |255 alpha - 1| < eps + c S             <=>  m_a = max(0, |255 alpha - 1| - c S) < eps
                    #   |T'/1e-4 - 1| < eps + (eps + c S) r     <=>  m_t = max(0, |T'/1e-4 - 1| - c S r) / (1 + r) < eps
                    #   (with the earlier contributors' r and c S r added to this one's: below)
                    # On a well-conditioned scene S ~ sigma <= 10 and r <= a few: the old margins to within 1e-5.
                    r_amp = np.where(opa[g] * exp(-sigma) >= dtype(ALPHA_MAX), 0, a / np.maximum(one - a, dtype(1e-3)))   # (a clamped alpha is exact)
                    cS = dtype(SIGMA_ABS) * S
                    m_a = np.maximum(np.abs(a * dtype(255.0) - one) - cS, 0)
                    # (T = prod (1 - alpha_j) carries every earlier contributor's error too: r_acc = sum r_j^2, sr_acc = sum c S_j r_j
                    #  over the pixel's contributors so far -- 77 of them deep in a list make the stop test 4-6 x as uncertain
                    #  as the last alpha alone: soak seed 59, profiles/r6/00_experiments.md section 2)
                    #  The eps shares are independent roundings and add in quadrature (r_acc = sum r_j^2: added plainly they flagged
                    #  5.9 % of configs[1] under EPS_PATH, whose eps already carries what accumulates on a well-conditioned scene);
                    #  the conditioned shares -- rare, large -- add plainly.
                    m_t = np.maximum(np.abs(Tn / dtype(T_STOP) - one) - (cS * r_amp + sr_acc), 0) / np.sqrt((one + r_amp) ** 2 + r_acc)
                    with np.errstate(divide="ignore", invalid="ignore"):
                        m_s = np.where(S > 0, np.abs(sigma) / S, np.inf)
                    could_count = a >= dtype(0.5 * ALPHA_MIN)     # a sigma flip only matters if alpha would count
                    if want_fw:
                        toggle = live & ((m_a < fe_a) | (could_count & (m_s < fe_s)))
                        fw += np.where(toggle, a * T, 0)
                        loose += np.where(toggle, a, 0)
                        closer = live & could_count & (m_t < tm[1])
                        t_min = np.where(closer, T, t_min)
                        fw += np.where(live & could_count & (m_t < fe_t + loose), T, 0)
                    tm[0] = np.where(live, np.minimum(tm[0], m_a), tm[0])
                    tm[1] = np.where(live & (a >= dtype(0.5 * ALPHA_MIN)), np.minimum(tm[1], m_t), tm[1])
                    tm[2] = np.where(live & could_count, np.minimum(tm[2], m_s), tm[2])
                    if zs is not None:
                        m_z = np.where(z_prev > 0, (zs[g] - z_prev) / zs[g], np.inf)
                        tm[3] = np.where(ok, np.minimum(tm[3], m_z), tm[3])
                        if want_fw:
                            fw += np.where(ok & (m_z < fe_z), wt_prev * a, 0)
                            wt_prev = np.where(ok, a * T, wt_prev)
                        z_prev = np.where(ok, zs[g], z_prev)
                stop = ok & (Tn <= dtype(T_STOP))
                done |= stop
                acc = ok & ~stop
                if margins:
                    r_acc = r_acc + np.where(acc, r_amp * r_amp, 0)
                    sr_acc = sr_acc + np.where(acc, cS * r_amp, 0)
                w = np.where(acc, a * T, 0)
                C += w[..., None] * col[g][None, None, :]
                T = np.where(acc, Tn, T)
                cur_last = np.where(acc, np.int32(i), cur_last)
                n_contrib += int(acc.sum())
            if background is not None:
                C = C + T[..., None] * np.asarray(background, dtype=dtype)[None, None]
            img[y_lo:y_hi, x_lo:x_hi] = C
            alpha_img[y_lo:y_hi, x_lo:x_hi] = one - T
            last[y_lo:y_hi, x_lo:x_hi] = cur_last
            if margins:
                marg[:, y_lo:y_hi, x_lo:x_hi] = tm
            if want_fw:
                fw_img[y_lo:y_hi, x_lo:x_hi] = fw
                tmin_img[y_lo:y_hi, x_lo:x_hi] = t_min
    stats = {"pair_evals": n_eval, "contribs": n_contrib}
    if margins:
        stats["margins"] = marg
    if want_fw:
        stats["flip_weight"], stats["t_at_min"] = fw_img, tmin_img
    return img, alpha_img, last, stats


# --------------------------------------------------------------------------------------
# classification of threshold flips (the parity gate of tests/ and smoke())
# --------------------------------------------------------------------------------------
# Relative slack granted to an fp32 implementation before a decision of the fp64 blend counts as
# "could have gone the other way".  Stage tests feed the blend IDENTICAL fp32 inputs: only the
# evaluation of sigma / exp / the running product differs (a few ulp, 1e-6 relative).  Whole-path tests
# also carry the fp32 projection (means2d to ~1e-4 px, conics to ~1e-5 relative), which moves sigma
# by up to ~1e-4 absolute, i.e. alpha by ~1e-4 relative, and depths by 2-3 ulps (2e-7 relative), which
# can swap the order of two Gaussians whose depths are within that of a tie.  Calibration at
# configs[1] (fp32 instantiation of the C++ port against its fp64 instantiation, 2 M pixels, 258 over
# 1e-4): alpha = T = 1e-4 and depth = 3e-7 already explain every one of them (largest error at a
# pixel nothing explains: 2.2e-5); the values below carry a factor ~2-3 on top and flag 1.6 % of
# the pixels of that frame.
# SIGMA_ABS: the absolute error of sigma per unit of S = 0.5 (|a| dx^2 + |c| dy^2) + |b dx dy| that the margins grant any
# fp32 evaluation: sixteen roundings at that magnitude (16 x 2^-24).  The dx, dy form makes three products and two sums of
# terms of size S; the HIP kernels' polynomial about the tile centre (raster_common.h) twice as many, of somewhat larger
# terms -- at 5e-7 the port's fp32 blend passed the full-size stage gate on the heavy-tailed scene and the HIP blend missed
# it at 2 pixels of 2 M by a third.  See the conditioned margins in `rasterize`.
SIGMA_ABS = 1e-6
EPS_STAGE = dict(alpha=2e-5, T=2e-5, sigma=2e-6, depth=0.0)
EPS_PATH = dict(alpha=3e-4, T=3e-4, sigma=2e-5, depth=5e-7)
# For GRADIENT rows (cpu_ref.render_f64(want_touched=True)): the relative error of T accumulates over the pixel's
# contributors (sum of d alpha / (1 - alpha): ~5e-4 after 50 of them with fp32-projected inputs), so the stop test
# T (1 - alpha) <= 1e-4 deep in a list can go the other way beyond EPS_PATH["T"].  In the image that moves the
# pixel by < 1e-4 |c| -- under the forward tolerance, which is why EPS_PATH need not cover it -- but a gradient
# row sees the whole term alpha T w ~ 1e-4 |w|, visible against rows of that size.
EPS_PATH_GRAD = dict(EPS_PATH, T=3e-3)


def explained_pixels(margins, eps, edge_mask=None):
    """bool [H,W]: pixels where some decision of the fp64 blend sits within `eps` (dict alpha / T /
    sigma / depth, relative) of flipping, or that a per-Gaussian knife edge of the projection reaches
    (`edge_mask`, see `gaussian_edge_mask`).  A test asserts that EVERY pixel over the 1e-4
    tolerance is one of these -- zero unexplained pixels -- and bounds how many there are."""
    m = (margins[0] < eps["alpha"]) | (margins[1] < eps["T"]) | (margins[2] < eps["sigma"])
    if len(margins) > 3:
        m = m | (margins[3] < eps.get("depth", 0.0))
    if edge_mask is not None:
        m = m | edge_mask
    return m


FLIP_SLACK = 1.5      # on the flip weight: the fp32 implementation's own alpha / T differ from these by ~eps


def check_frame(got, got_alpha, ref, ref_alpha, margins, eps, edge_mask=None, tol=1e-4,
                expected_depth=False, max_explained=0.05, what="frame", flip_weight=None, feat_max=None,
                require_flip_bound=False, noise_weight=None):
    """THE forward parity gate.  got / ref [H,W,D], alphas [H,W].  Asserts
      * every pixel whose colour, depth-sum or alpha differs by more than `tol` (1e-4 abs, the
        north-star tolerance) is a pixel where the fp64 blend took a decision within `eps` of
        flipping (explained_pixels) -- ZERO unexplained pixels;
      * such could-flip pixels are at most `max_explained` of the image (the gate is not vacuous; 5 % since the stop test's
        margin counts the error T has accumulated over the pixel's contributors: configs[4], lists of ~800 entries, reads
        3.7 % under EPS_PATH, configs[1] 2-3 %).
      * with flip_weight [H,W] (rasterize(flip_eps=eps) / cpu_ref.render_f64(flip_eps=eps); for whole-path
        frames already including the edge pixels' share) and feat_max [D] (largest |feature| per channel over
        the visible Gaussians): a could-flip pixel may be off by at most what the near-flip decisions are worth,
            |d colour_c| <= tol + FLIP_SLACK * flip_weight * 2 feat_max_c,   |d alpha| <= tol + FLIP_SLACK * flip_weight
        (toggling a Gaussian changes the pixel by alpha T (c - colour behind it)) -- an error larger than that
        at a could-flip pixel is as much a failure as an error at any other pixel.
      * noise_weight [H,W] (cpu_ref.blend_f64: stage-isolated blends only; needs feat_max): the tolerance of EVERY pixel grows
        by FLIP_SLACK * noise_weight * 2 feat_max_c (alpha: * 1) -- the smooth part of an fp32 blend's error where sigma is
        ill-conditioned (SIGMA_ABS); ~1e-6 elsewhere.
    expected_depth: the last channel is depth-sum / alpha ("ED"): ED = D / alpha, so
    |dED| <= tol (1 + |ED|) / alpha is the same 1e-4 bound on D and alpha propagated through the divide.
    Returns a dict of statistics (for printing)."""
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    ga, ra = np.asarray(got_alpha, dtype=np.float64), np.asarray(ref_alpha, dtype=np.float64)
    ga, ra = ga.reshape(ga.shape[:2]), ra.reshape(ra.shape[:2])
    d = np.abs(got - ref)
    lim = np.full(d.shape, tol)
    lim_a = np.full(ra.shape, tol)
    if expected_depth:
        lim[..., -1] = tol * (1.0 + np.abs(ref[..., -1])) / np.maximum(ra, 1e-10)
    if noise_weight is not None:
        assert not expected_depth, "noise_weight: plain channel sums only"
        nz = FLIP_SLACK * np.asarray(noise_weight, dtype=np.float64)
        lim = lim + nz[..., None] * 2.0 * np.asarray(feat_max, dtype=np.float64).reshape(-1)[None, None, :got.shape[-1]]
        lim_a = lim_a + nz
    bad = (d > lim).any(-1) | (np.abs(ga - ra) > lim_a)
    ex = explained_pixels(margins, eps, edge_mask)
    unexplained = bad & ~ex
    excess = np.where(ex[..., None], 0.0, d / lim)
    stats = {"over_tol": int(bad.sum()), "unexplained": int(unexplained.sum()),
             "could_flip_frac": float(ex.mean()),
             "max_err_nonflip": float(np.where(ex[..., None], 0.0, d[..., :got.shape[-1] - (1 if expected_depth else 0)]).max())
             if got.shape[-1] > (1 if expected_depth else 0) else 0.0,
             "max_err_over_tol_nonflip": float(excess.max()), "max_err": float(d.max())}
    assert stats["unexplained"] == 0, (
        f"{what}: {stats['unexplained']} pixels differ from the oracle by more than {tol:g} without any "
        f"threshold within eps of flipping (first at {tuple(np.argwhere(unexplained)[0])}); {stats}")
    assert stats["could_flip_frac"] <= max_explained, f"{what}: gate is vacuous: {stats}"
    assert flip_weight is not None or not require_flip_bound, f"{what}: no flip weight given"
    if flip_weight is not None:
        fw = FLIP_SLACK * np.asarray(flip_weight, dtype=np.float64)
        fm = np.asarray(feat_max, dtype=np.float64).reshape(-1)[:got.shape[-1]]
        lim_f = lim + fw[..., None] * 2.0 * fm[None, None, :]
        if expected_depth:      # d(D / alpha) <= (dD + |ED| d alpha) / alpha
            lim_f[..., -1] = (tol * (1.0 + np.abs(ref[..., -1])) + fw * (2.0 * fm[-1] + np.abs(ref[..., -1]))) \
                / np.maximum(np.minimum(ra, ga), 1e-10)
        over = ex & ((d > lim_f).any(-1) | (np.abs(ga - ra) > lim_a + fw))
        ratio = np.where(ex[..., None], d / lim_f, 0.0)
        stats.update(flip_over_bound=int(over.sum()), max_flip_err_over_bound=float(ratio.max()),
                     max_err_at_flip_pixels=float(np.where(ex[..., None], d[..., :3], 0.0).max()),
                     max_flip_weight=float(np.where(ex, fw / FLIP_SLACK, 0.0).max()))
        assert stats["flip_over_bound"] == 0, (
            f"{what}: {stats['flip_over_bound']} could-flip pixels are off by more than their near-flip decisions "
            f"are worth (first at {tuple(np.argwhere(over)[0])}: |d| = {d[tuple(np.argwhere(over)[0])]}, bound "
            f"{lim_f[tuple(np.argwhere(over)[0])]}); {stats}")
    return stats


REL_SLACK = 1.5       # check_frame_against_fp32_port: how much worse than the fp32 restatement a frame may be


def check_frame_against_fp32_port(got, got_alpha, ref, ref_alpha, fp32_render, fp32_alpha, margins, eps, edge_mask=None,
                                  tol=1e-4, expected_depth=False, what="frame", flip_weight=None, feat_max=None, counts=True):
    """The forward gate for scenes that are ILL-CONDITIONED FOR FP32 (needle-like Gaussians seen hundreds of pixels from
    their means, lists of thousands of entries): there a plain fp32 restatement of the reference's own formulas -- the C++
    port's float instantiation, fp32_render / fp32_alpha (depth SUM in the last channel with expected_depth) -- is itself
    thousands of pixels away from the fp64 answer (`ref`), because sigma of such a pair is uncertain by percents from fp32
    means2d / conics alone, and no two fp32 pipelines agree to 1e-4.  check_frame's zero-unexplained-pixels rule cannot
    hold for ANY fp32 implementation on such a scene; what can be demanded, and is here, is that the frame under test is in
    the SAME NOISE CLASS as that fp32 restatement against the fp64 answer: at most REL_SLACK times its pixels over `tol`
    (+ 8), its pixels unexplained by a near-flip decision (explained_pixels), its 99.9th / 99.99th percentile of the error
    (in units of the tolerance, the expected-depth channel through the divide); and every pixel that is farther off than twice
    its maximum (or 50 tolerances) must be a pixel where the fp64 blend took a decision within eps of flipping (explained_pixels)
    and, with flip_weight / feat_max (render_f64(flip_eps=)), be off by no more than that decision is worth.  (Round 5 bounded the
    maximum itself at 50 tolerances, "what one flipped decision is worth"; it is worth alpha T with T (1 - alpha) ~ 1e-4 when
    the decision is the stop test -- the closing Gaussian is not accumulated --, i.e. 1e-4 alpha / (1 - alpha): 125
    tolerances for the opaque Gaussian of soak seed 48, up to 1,000 at the 0.999 clamp; profiles/r6/00_experiments.md.)
    counts=False drops the two COUNT clauses (pixels over tolerance, unexplained pixels) and keeps the percentiles and the far
    outliers: on a small frame the counts are a lottery over a handful of needle-like Gaussians -- the 30 pixels along one
    needle whose conic one implementation happens to round 1e-4 worse are perfectly correlated (soak seed 67: 83 against 21
    pixels with projections of the same accuracy and a blend that passes its absolute stage gate) --, the percentiles are not.
    Returns the statistics of both."""
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    ga, ra = np.asarray(got_alpha, dtype=np.float64), np.asarray(ref_alpha, dtype=np.float64)
    ga, ra = ga.reshape(ga.shape[:2]), ra.reshape(ra.shape[:2])
    r32 = np.array(fp32_render, dtype=np.float64)
    a32 = np.asarray(fp32_alpha, dtype=np.float64).reshape(ra.shape)
    lim_px = np.full(ref.shape, tol)
    if expected_depth:
        r32[..., -1] /= np.maximum(a32, 1e-10)
        lim_px[..., -1] = tol * (1.0 + np.abs(ref[..., -1])) / np.maximum(ra, 1e-10)
    ex = explained_pixels(margins, eps, edge_mask)

    def stats_of(x, xa):
        e = np.maximum((np.abs(x - ref) / lim_px).max(-1), np.abs(xa - ra) / tol)      # error in units of the tolerance
        return {"over_tol": int((e > 1.0).sum()), "unexplained": int(((e > 1.0) & ~ex).sum()),
                "err_q999": float(np.quantile(e, 0.999)), "err_q9999": float(np.quantile(e, 0.9999)), "err_max": float(e.max())}
    st, st32 = stats_of(got, ga), stats_of(r32, a32)
    out = {"could_flip_frac": float(ex.mean())}
    out.update({k: v for k, v in st.items()})
    out.update({k + "_fp32_port": v for k, v in st32.items()})
    # two fp32 realisations of the same noise are compared, not a bound with its subject: over random heavy-tailed scenes the
    # HIP path is sometimes the closer one (315 pixels against 589 at 1 M / 1080p) and sometimes the other way round (583
    # against 480; scripts/soak_heavy.py) -- the same noise class.  The frame may be worse than the restatement by REL_SLACK
    # (+ a few pixels where the counts are tiny); the maximum is one pixel of the frame and gets a factor of two.
    for k in st:
        if k == "err_max" or (not counts and not k.startswith("err")):
            continue
        lim = max(REL_SLACK * st32[k], 1.0) if k.startswith("err") else REL_SLACK * st32[k] + 8
        assert st[k] <= lim, (
            f"{what}: farther from the fp64 answer than a plain fp32 restatement of the reference's formulas allows ({k}: {st[k]} > {lim}): {out}")
    # the far outliers: flipped decisions, each of them
    e = np.maximum((np.abs(got - ref) / lim_px).max(-1), np.abs(ga - ra) / tol)
    far = e > max(2.0 * st32["err_max"], 50.0)
    out["far_pixels"] = int(far.sum())
    assert not (far & ~ex).any(), (
        f"{what}: {int((far & ~ex).sum())} pixels are more than {max(2.0 * st32['err_max'], 50.0):.0f} tolerances off at no near-flip "
        f"decision of the fp64 blend (first at {tuple(np.argwhere(far & ~ex)[0])}): {out}")
    if flip_weight is not None and far.any():
        fw = FLIP_SLACK * np.asarray(flip_weight, dtype=np.float64)
        fm = np.asarray(feat_max, dtype=np.float64).reshape(-1)[:got.shape[-1]]
        lim_f = lim_px + fw[..., None] * 2.0 * fm[None, None, :]
        if expected_depth:
            lim_f[..., -1] = (tol * (1.0 + np.abs(ref[..., -1])) + fw * (2.0 * fm[-1] + np.abs(ref[..., -1]))) / np.maximum(np.minimum(ra, ga), 1e-10)
        over = far & ((np.abs(got - ref) > lim_f).any(-1) | (np.abs(ga - ra) > tol + fw))
        assert not over.any(), f"{what}: {int(over.sum())} far-off pixels exceed what their near-flip decisions are worth: {out}"
    return out


def gaussian_edge_mask(p, opacities, width, height, tile_size=16, eps_radius=3e-5, d_mu=1e-3,
                       eps_alpha=1e-3, near_plane=0.01, far_plane=1e10, return_weight=False):
    """bool [H,W]: pixels that a Gaussian reaches with alpha >= (1 - eps_alpha)/255 inside a tile
    whose membership in that Gaussian's tile rectangle depends on a knife edge of A.2 steps 2-5/7:
    3 sqrt(lambda) within eps_radius (relative) of an integer (the ceil), mean2d +- radius within
    d_mu pixels of a tile boundary or of the screen-cull limits, depth within 1e-5 relative of the
    near / far plane.  `p` is the dict `project` returned ("lam" included).  Under the per-axis radius rule (A.4)
    the two extents e sqrt(Sigma_ii) take the place of 3 sqrt(lambda) ("extent_xy"), and an opacity within 1e-5
    (relative) of 1/255 makes the Gaussian's very presence uncertain."""
    mask = np.zeros((height, width), dtype=bool)
    weight = np.zeros((height, width))           # sum of the alphas of the uncertain Gaussians reaching the pixel
    tw, th = -(-width // tile_size), -(-height // tile_size)
    v = p["extent_xy"] if "extent_xy" in p else np.stack([3.0 * np.sqrt(p["lam"])] * 2, axis=-1)     # [N,2]
    z, mu = p["z"], p["mu"]
    r = np.ceil(v)
    fr = v - np.floor(v)
    # (the opacity-aware extent goes through a logarithm of the opacity: its relative rounding is larger near o = 1/255)
    tol = eps_radius * np.maximum(v, 1.0) + 1e-6
    if "extent" in p:                                           # d(extent) = d(ln) / extent with d(ln) ~ 2e-7 in fp32
        tol = tol + v * (2e-7 / np.maximum(p["extent"] ** 2, 1e-12))[:, None]
    r_lo = np.where((fr > 0) & (fr < tol), r - 1, r)            # v barely above an integer: ceil may drop
    r_hi = np.where((1.0 - fr < tol) | (fr == 0), r + 1, r)     # v barely below one: ceil may rise
    zin = (z >= near_plane * (1 + 1e-5)) & (z <= far_plane * (1 - 1e-5)) & p["det_ok"]
    zout = (z >= near_plane * (1 - 1e-5)) & (z <= far_plane * (1 + 1e-5)) & p["det_ok"]
    if "op_rule" in p:                                          # A.4: culled below 1/255 -- uncertain within 1e-5 of it
        zin &= p["op_rule"] >= (1 + 1e-5) * ALPHA_MIN
        zout &= p["op_rule"] >= (1 - 1e-5) * ALPHA_MIN

    def rect(rr, grow):
        d = d_mu if grow else -d_mu
        rx, ry = rr[:, 0], rr[:, 1]
        x0 = np.clip(np.floor((mu[:, 0] - rx - d) / tile_size), 0, tw)
        x1 = np.clip(np.ceil((mu[:, 0] + rx + d) / tile_size), 0, tw)
        y0 = np.clip(np.floor((mu[:, 1] - ry - d) / tile_size), 0, th)
        y1 = np.clip(np.ceil((mu[:, 1] + ry + d) / tile_size), 0, th)
        on = ~((mu[:, 0] + rx + d <= 0) | (mu[:, 0] - rx - d >= width)
               | (mu[:, 1] + ry + d <= 0) | (mu[:, 1] - ry - d >= height)) & (rx > 0) & (ry > 0)
        return x0.astype(int), x1.astype(int), y0.astype(int), y1.astype(int), on
    ix0, ix1, iy0, iy1, ion = rect(r_lo, False)
    ox0, ox1, oy0, oy1, oon = rect(r_hi, True)
    ion &= zin
    oon &= zout
    ix0, ix1, iy0, iy1 = (np.where(ion, a, 0) for a in (ix0, ix1, iy0, iy1))
    unsure = oon & ((ox0 != ix0) | (ox1 != ix1) | (oy0 != iy0) | (oy1 != iy1) | ~ion)
    con, opa = p["conics_all"], np.asarray(opacities, dtype=np.float64)
    for g in np.nonzero(unsure)[0]:
        for ty in range(oy0[g], oy1[g]):
            for tx in range(ox0[g], ox1[g]):
                if ion[g] and ix0[g] <= tx < ix1[g] and iy0[g] <= ty < iy1[g]:
                    continue
                ys = np.arange(ty * tile_size, min((ty + 1) * tile_size, height)) + 0.5
                xs = np.arange(tx * tile_size, min((tx + 1) * tile_size, width)) + 0.5
                dy, dx = np.meshgrid(mu[g, 1] - ys, mu[g, 0] - xs, indexing="ij")
                sig = 0.5 * (con[g, 0] * dx * dx + con[g, 2] * dy * dy) + con[g, 1] * dx * dy
                a = np.minimum(ALPHA_MAX, opa[g] * np.exp(-sig))
                hit = a >= (1.0 - eps_alpha) * ALPHA_MIN
                mask[ty * tile_size:ty * tile_size + len(ys), tx * tile_size:tx * tile_size + len(xs)] |= hit
                weight[ty * tile_size:ty * tile_size + len(ys), tx * tile_size:tx * tile_size + len(xs)] += np.where(hit, a, 0.0)
    if return_weight:
        return mask, int(unsure.sum()), weight
    return mask, int(unsure.sum())


# --------------------------------------------------------------------------------------
# whole frame
# --------------------------------------------------------------------------------------
def render(means, quats, scales, opacities, sh_or_colors, viewmat, K, width, height,
           sh_degree=None, tile_size=16, render_mode="RGB", eps2d=0.3,
           near_plane=0.01, far_plane=1e10, radius_clip=0.0, background=None,
           rasterize_mode="classic", dtype=np.float64, margins=False, flip_eps=None, radius_rule="classic"):
    """Full single-camera frame following SURVEY.md A.1/A.2.  Inputs are post-activation
    (scales = exp(log_s), opacities = sigmoid(logit)).  Returns (colors[H,W,D],
    alpha[H,W,1], meta).  margins=True adds meta["margins"] (see `rasterize`) and
    meta["edge_mask"] / meta["n_edge_gaussians"] (see `gaussian_edge_mask`); flip_eps (an EPS_* dict) also
    meta["flip_weight"] [H,W] (edge pixels included) and meta["feat_max"] [D] for `check_frame`."""
    tile_w = -(-width // tile_size)
    tile_h = -(-height // tile_size)
    p = project(means, quats, scales, viewmat, K, width, height, eps2d, near_plane,
                far_plane, radius_clip, dtype, radius_rule=radius_rule, opacities=opacities,
                antialiased=rasterize_mode == "antialiased")
    opac = np.asarray(opacities, dtype=dtype)
    if rasterize_mode == "antialiased":
        opac = opac * p["compensations"]
    if sh_degree is None:
        rgb = np.asarray(sh_or_colors, dtype=dtype)
    else:
        rgb = sh_colors(sh_degree, means, campos_from_viewmat(viewmat), sh_or_colors,
                        dtype)
        rgb = np.where(visible(p["radii"])[:, None], rgb, 0)
    depth = p["depths"][:, None]
    if render_mode in ("RGB",):
        feats = rgb
    elif render_mode in ("D", "ED"):
        feats = depth
    elif render_mode in ("RGB+D", "RGB+ED"):
        feats = np.concatenate([rgb, depth], axis=-1)
    else:
        raise ValueError(render_mode)
    tpg, isect_ids, flatten_ids = isect_tiles(p["means2d"], p["radii"], p["depths"],
                                              tile_size, tile_w, tile_h, dtype=dtype)
    offs = isect_offsets(isect_ids, 1, tile_w, tile_h)[0]
    bg = None
    if background is not None:
        bg = np.asarray(background, dtype=dtype)
    img, alpha, last, stats = rasterize(p["means2d"], p["conics"], feats, opac,
                                        flatten_ids, offs, width, height, tile_size,
                                        bg, dtype, margins=margins, depths=p["depths"] if margins else None,
                                        flip_eps=flip_eps if margins else None)
    if render_mode in ("ED", "RGB+ED"):
        img = img.copy()
        img[..., -1] = img[..., -1] / np.maximum(alpha, dtype(1e-10))
    meta = dict(p)
    if margins:
        em, n_edge, ew = gaussian_edge_mask(p, opac, width, height, tile_size,
                                            near_plane=near_plane, far_plane=far_plane, return_weight=True)
        meta.update(edge_mask=em, n_edge_gaussians=n_edge)
        if flip_eps is not None:
            # an uncertain tile membership is worth the Gaussian's alpha (T <= 1) and moves the pixel's closest
            # T threshold by the factor (1 - alpha)
            fw = stats.pop("flip_weight") + ew
            fw = fw + np.where(em & (stats["margins"][1] < flip_eps["T"] + ew), stats.pop("t_at_min"), 0.0)
            vis = visible(p["radii"])
            meta.update(flip_weight=fw, feat_max=(np.abs(feats[vis]).max(axis=0) if vis.any()
                                                  else np.zeros(feats.shape[1])))
    meta.update(tiles_per_gauss=tpg, isect_ids=isect_ids, flatten_ids=flatten_ids,
                isect_offsets=offs, last_ids=last, colors=rgb, opacities=opac,
                tile_width=tile_w, tile_height=tile_h, n_isect=len(flatten_ids),
                n_vis=int(visible(p["radii"]).sum()), **stats)
    return img, alpha[..., None], meta
