"""PyTorch (CPU, float64) restatement of the render path whose AUTOGRAD is the backward
oracle (TEST INFRASTRUCTURE ONLY; PARITY UNPINNED -- see oracle/gs_oracle_np.py header: the
reference, Maxwell-Zhao/RoboSimGS, ships no renderer, so this follows the published gsplat
1.x algorithm).

The forward is written with differentiable torch ops only (no custom backward), vectorised
per tile with cumulative products instead of the sequential per-pixel loop; its forward
values are cross-checked against the literal NumPy loop in tests/test_oracle_*.py and its
gradients against central finite differences.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import torch

from . import gs_oracle_np as O

DT = torch.float64


def quat_to_rotmat(q):
    q = q / q.norm(dim=-1, keepdim=True)
    w, x, y, z = q.unbind(-1)
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=-1)
    return R.reshape(q.shape[:-1] + (3, 3))


def project(means, quats, scales, viewmat, K, width, height, eps2d=0.3, near_plane=0.01,
            far_plane=1e10, radius_clip=0.0, radius_rule="classic", opacities=None, antialiased=False):
    """A.2 steps 1-5; differentiable in means/quats/scales/viewmat.  Returns dict with
    radii (int, no grad), means2d, depths, conics, compensations (zeros where culled).
    radius_rule / opacities / antialiased: as gs_oracle_np.project (radii [N,2] under "opacity_aware")."""
    Rcw, tcw = viewmat[:3, :3], viewmat[:3, 3]
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    pc = means @ Rcw.T + tcw
    x, y, z = pc.unbind(-1)
    valid = (z >= near_plane) & (z <= far_plane)
    zs = torch.where(valid, z, torch.ones_like(z))
    M = quat_to_rotmat(quats) * scales[:, None, :]
    cov = M @ M.transpose(1, 2)
    cov_c = Rcw @ cov @ Rcw.T
    tanx, tany = 0.5 * width / fx, 0.5 * height / fy
    lim_xp, lim_xn = (width - cx) / fx + 0.3 * tanx, cx / fx + 0.3 * tanx
    lim_yp, lim_yn = (height - cy) / fy + 0.3 * tany, cy / fy + 0.3 * tany
    rz = 1.0 / zs
    tx = zs * torch.minimum(lim_xp, torch.maximum(-lim_xn, x * rz))
    ty = zs * torch.minimum(lim_yp, torch.maximum(-lim_yn, y * rz))
    zero = torch.zeros_like(rz)
    J = torch.stack([fx * rz, zero, -fx * tx * rz * rz,
                     zero, fy * rz, -fy * ty * rz * rz], dim=-1).reshape(-1, 2, 3)
    cov2 = J @ cov_c @ J.transpose(1, 2)
    mu = torch.stack([fx * x * rz + cx, fy * y * rz + cy], dim=-1)
    a, b, c = cov2[:, 0, 0], cov2[:, 0, 1], cov2[:, 1, 1]
    det0 = a * c - b * b
    a = a + eps2d
    c = c + eps2d
    det = a * c - b * b
    valid = valid & (det > 0)
    dets = torch.where(det > 0, det, torch.ones_like(det))
    comp = torch.sqrt(torch.clamp(det0 / dets, min=0.0))
    conic = torch.stack([c / dets, -b / dets, a / dets], dim=-1)
    m = 0.5 * (a + c)
    lam = m + torch.sqrt(torch.clamp(m * m - dets, min=0.01))
    if radius_rule == "classic":
        radius = torch.ceil(3.0 * torch.sqrt(lam)).detach()
        radius_y = radius
        valid = valid & (radius > radius_clip)
    else:                                   # SURVEY.md A.4 (gsplat >= 1.5): per-axis, opacity-aware, not differentiable
        ext = torch.full_like(lam, O.EXTENT_MAX)
        if opacities is not None:
            op = (opacities * comp if antialiased else opacities).detach()
            ok = op >= 1.0 / 255.0
            ext = torch.minimum(ext, torch.sqrt(2.0 * torch.log(torch.where(ok, op, torch.ones_like(op)) * 255.0)))
            valid = valid & ok
        radius = torch.ceil(ext * torch.sqrt(a)).detach()
        radius_y = torch.ceil(ext * torch.sqrt(c)).detach()
        valid = valid & ((radius > radius_clip) | (radius_y > radius_clip)) & (radius > 0) & (radius_y > 0)
    valid = valid & ~((mu[:, 0] + radius <= 0) | (mu[:, 0] - radius >= width)
                      | (mu[:, 1] + radius_y <= 0) | (mu[:, 1] - radius_y >= height))
    vf = valid.to(means.dtype)
    rr = radius if radius_rule == "classic" else torch.stack([radius, radius_y], dim=-1)
    vr = valid if radius_rule == "classic" else valid[:, None]
    return {"radii": torch.where(vr, rr, torch.zeros_like(rr)).to(torch.int32),
            "means2d": mu * vf[:, None], "depths": z * vf, "conics": conic * vf[:, None],
            "compensations": comp * vf}


def sh_basis(degree, dirs):
    x, y, z = dirs.unbind(-1)
    Y = [torch.full_like(x, O.SH_C0)]
    if degree >= 1:
        Y += [-O.SH_C1 * y, O.SH_C1 * z, -O.SH_C1 * x]
    if degree >= 2:
        z2, fC1, fS1 = z * z, x * x - y * y, 2 * x * y
        t = O.SH_C2[1] * z
        Y += [O.SH_C2[0] * fS1, t * y, O.SH_C2[2] * z2 - O.SH_C2[3], t * x, O.SH_C2[0] * fC1]
    if degree >= 3:
        u = O.SH_C3[2] * z2 + O.SH_C3[3]
        w = O.SH_C3[1] * z
        fC2, fS2 = x * fC1 - y * fS1, x * fS1 + y * fC1
        Y += [O.SH_C3[0] * fS2, w * fS1, u * y, z * (O.SH_C3[4] * z2 - O.SH_C3[5]), u * x,
              w * fC1, O.SH_C3[0] * fC2]
    return torch.stack(Y, dim=-1)


def spherical_harmonics(degree, dirs, coeffs):
    d = dirs / dirs.norm(dim=-1, keepdim=True)
    Y = sh_basis(degree, d)
    return torch.einsum("nk,nkc->nc", Y, coeffs[:, :(degree + 1) ** 2])


def rasterize(means2d, conics, colors, opacities, flatten_ids, offsets, width, height,
              tile_size=16, background=None):
    """A.2 step 9 vectorised per tile: alpha[P,G], exclusive cumprod for T, first index where
    T*(1-alpha) <= 1e-4 terminates the pixel.  Differentiable.  Returns image [H,W,D],
    alpha [H,W]."""
    D = colors.shape[1]
    th, tw = offsets.shape
    n_isect = len(flatten_ids)
    flat = torch.cat([torch.as_tensor(offsets).reshape(-1).long(), torch.tensor([n_isect])])
    ids = torch.as_tensor(flatten_ids).long()
    img = torch.zeros(height, width, D, dtype=means2d.dtype)
    alpha_img = torch.zeros(height, width, dtype=means2d.dtype)
    rows, cols = [], []
    for ty in range(th):
        row_img, row_alpha = [], []
        for tx in range(tw):
            s, e = int(flat[ty * tw + tx]), int(flat[ty * tw + tx + 1])
            y_lo, y_hi = ty * tile_size, min((ty + 1) * tile_size, height)
            x_lo, x_hi = tx * tile_size, min((tx + 1) * tile_size, width)
            hh, ww = y_hi - y_lo, x_hi - x_lo
            if hh <= 0 or ww <= 0:
                continue
            py, px = torch.meshgrid(torch.arange(y_lo, y_hi, dtype=means2d.dtype) + 0.5,
                                    torch.arange(x_lo, x_hi, dtype=means2d.dtype) + 0.5,
                                    indexing="ij")
            px, py = px.reshape(-1, 1), py.reshape(-1, 1)
            g = ids[s:e]
            if len(g) == 0:
                C = torch.zeros(hh * ww, D, dtype=means2d.dtype)
                T_end = torch.ones(hh * ww, dtype=means2d.dtype)
            else:
                dx = means2d[g, 0][None] - px
                dy = means2d[g, 1][None] - py
                sigma = 0.5 * (conics[g, 0][None] * dx * dx + conics[g, 2][None] * dy * dy) \
                    + conics[g, 1][None] * dx * dy
                a = torch.clamp(opacities[g][None] * torch.exp(-sigma), max=O.ALPHA_MAX)
                ok = (sigma >= 0) & (a >= O.ALPHA_MIN)
                a_eff = torch.where(ok, a, torch.zeros_like(a))
                # transmittance if nothing ever stopped
                T_after = torch.cumprod(1 - a_eff, dim=1)
                T_before = torch.cat([torch.ones_like(T_after[:, :1]), T_after[:, :-1]], dim=1)
                stop = ok & (T_after <= O.T_STOP)
                stopped = torch.cumsum(stop.to(torch.int64), dim=1) > 0    # inclusive
                acc = ok & ~stopped
                w = torch.where(acc, a * T_before, torch.zeros_like(a))
                C = w @ colors[g]
                T_end = torch.prod(torch.where(acc, 1 - a, torch.ones_like(a)), dim=1)
            if background is not None:
                C = C + T_end[:, None] * background[None]
            row_img.append(C.reshape(hh, ww, D))
            row_alpha.append((1 - T_end).reshape(hh, ww))
        if row_img:
            rows.append(torch.cat(row_img, dim=1))
            cols.append(torch.cat(row_alpha, dim=1))
    return torch.cat(rows, dim=0), torch.cat(cols, dim=0)


def render(means, quats, scales, opacities, sh_or_colors, viewmat, K, width, height,
           sh_degree=None, tile_size=16, render_mode="RGB", eps2d=0.3, near_plane=0.01,
           far_plane=1e10, radius_clip=0.0, background=None, rasterize_mode="classic", radius_rule="classic"):
    """Whole frame, differentiable w.r.t. means/quats/scales/opacities/colours/viewmat.
    The (integer) tile lists come from the NumPy oracle evaluated at the current values."""
    p = project(means, quats, scales, viewmat, K, width, height, eps2d, near_plane, far_plane,
                radius_clip, radius_rule=radius_rule, opacities=opacities, antialiased=rasterize_mode == "antialiased")
    opac = opacities * p["compensations"] if rasterize_mode == "antialiased" else opacities
    vis = p["radii"] > 0 if p["radii"].dim() == 1 else p["radii"][:, 0] > 0
    if sh_degree is None:
        rgb = sh_or_colors
    else:
        campos = -viewmat[:3, :3].T @ viewmat[:3, 3]
        rgb = torch.clamp(spherical_harmonics(sh_degree, means - campos, sh_or_colors) + 0.5,
                          min=0.0)
        rgb = rgb * vis.to(rgb.dtype)[:, None]
    depth = p["depths"][:, None]
    feats = {"RGB": rgb, "D": depth, "ED": depth}.get(render_mode)
    if feats is None:
        feats = torch.cat([rgb, depth], dim=-1)
    tile_w, tile_h = -(-width // tile_size), -(-height // tile_size)
    _, isect_ids, flatten_ids = O.isect_tiles(p["means2d"].detach().numpy(),
                                              p["radii"].numpy(), p["depths"].detach().numpy(),
                                              tile_size, tile_w, tile_h)
    offs = O.isect_offsets(isect_ids, 1, tile_w, tile_h)[0]
    img, alpha = rasterize(p["means2d"], p["conics"], feats, opac, flatten_ids, offs, width,
                           height, tile_size, background)
    if render_mode in ("ED", "RGB+ED"):
        img = torch.cat([img[..., :-1], img[..., -1:] / alpha.clamp(min=1e-10)[..., None]], -1)
    return img, alpha[..., None], p
