"""The gradient gate shared by the GPU backward tests and the oracle-only tests (a helper module, not a test file).

A gradient row of one Gaussian may differ from the fp64 oracle's
  (a) by rounding:  |d| <= HEADROOM * row_tol * (|row|_max + 1e-3 |tensor|_max), and
  (b) by what the near-flip decisions of the fp64 blend are worth at the could-flip pixels the Gaussian reaches:
      + FLIP_SLACK * budget   (oracle/gs_cpu.cpp Extras::budget, per row; chained through projection / SH with the
      absolute Jacobian by `chained_budget` below).
With a budget EVERY row is held to (a) + (b); without one (anti-aliased cases, stage tests on identical inputs) the
older rule applies: rows over HEADROOM * row_tol must be `touched`, and at most `bad_frac` of the rows may exceed row_tol.
"""
import numpy as np

UNTOUCHED_HEADROOM = 2.0
FLIP_SLACK = 1.5


def compare(name, got, ref, row_tol=2e-3, bad_frac=2e-3, cos_min=0.9999, touched=None, budget=None, verbose=True):
    """got: torch tensor or array; ref: array [rows, ...].  touched [rows] bool; budget [rows] or the shape of ref."""
    if hasattr(got, "detach"):
        got = got.detach().cpu().double().numpy()
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    ref = ref.reshape(ref.shape[0], -1)
    got = got.reshape(ref.shape)
    scale = np.abs(ref).max(axis=1, keepdims=True) + 1e-3 * np.abs(ref).max() + 1e-30
    diff = np.abs(got - ref)
    err = (diff / scale).max(axis=1)
    frac = (err > row_tol).mean()
    cos = (got * ref).sum() / (np.linalg.norm(got) * np.linalg.norm(ref) + 1e-30)
    assert np.isfinite(got).all(), f"{name}: non-finite gradient"
    assert frac <= bad_frac, f"{name}: {frac:.4%} rows over {row_tol} (max {err.max():.3e})"
    assert cos >= cos_min, f"{name}: cosine {cos:.7f}"
    stats = {"rows_over_tol": int((err > row_tol).sum()), "rows": len(err), "cosine": float(cos)}
    if budget is not None:
        b = np.asarray(budget, dtype=np.float64)
        b = b.reshape(ref.shape[0], -1) if b.size == ref.size else b.reshape(-1, 1)
        allow = UNTOUCHED_HEADROOM * row_tol * scale + FLIP_SLACK * b
        ratio = (diff / allow).max(axis=1)
        over = ratio > 1.0
        with_budget = (b > 0).any(axis=1)
        stats.update(rows_over_budget=int(over.sum()), worst_ratio=float(ratio.max()),
                     rows_with_budget=float(with_budget.mean()),
                     rows_over_rounding=int(((diff / (UNTOUCHED_HEADROOM * row_tol * scale)).max(axis=1) > 1.0).sum()))
        if verbose:
            print(f"\n{name}: {stats['rows_over_tol']} of {len(err)} rows over {row_tol:g}; {stats['rows_over_rounding']} over "
                  f"{UNTOUCHED_HEADROOM:g} x tol of which {stats['rows_over_budget']} also over their flip budget; worst "
                  f"|d| / (rounding + {FLIP_SLACK} budget) {ratio.max():.3f}; rows with a budget {with_budget.mean():.2%}, "
                  f"cosine {cos:.7f}")
        assert not over.any(), (
            f"{name}: {int(over.sum())} rows differ from the oracle by more than rounding + {FLIP_SLACK} x their flip budget "
            f"(first row {int(np.argmax(over))}: |d| / allowed {ratio[np.argmax(over)]:.3f}, scaled error {err[np.argmax(over)]:.3e})")
    elif touched is not None:
        touched = np.asarray(touched, dtype=bool).reshape(-1)
        unexplained = (err > UNTOUCHED_HEADROOM * row_tol) & ~touched
        clean = float(err[~touched].max()) if (~touched).any() else 0.0
        stats.update(unexplained=int(unexplained.sum()), clean_max=clean)
        if verbose:
            print(f"\n{name}: {stats['rows_over_tol']} of {len(err)} rows over {row_tol:g}, unexplained "
                  f"{int(unexplained.sum())}; largest scaled error on rows no could-flip pixel touches {clean:.3e}; "
                  f"touched rows {touched.mean():.2%}, cosine {cos:.7f}")
        assert not unexplained.any(), (
            f"{name}: {int(unexplained.sum())} rows over {UNTOUCHED_HEADROOM * row_tol:g} belong to Gaussians that touch no could-flip "
            f"pixel (first row {int(np.argmax(unexplained))}, scaled error {err[np.argmax(unexplained)]:.3e})")
    return stats


def chained_budget(params, outputs, budgets):
    """Budgets of the blend's gradient rows -> budgets of the parameters' gradient rows.

    params   {name: fp64 leaf tensor [N, ...]} of the torch oracle's projection / SH graph
    outputs  {name: tensor [N, k]}: the projected quantities the blend consumes (means2d, conics, feats ...); every
             row depends on its own Gaussian's parameters only, so ONE backward pass with ones in component j gives
             every Gaussian's d out_j / d params
    budgets  {name: array [N]}: bound on each component of d loss / d out[name] (oracle/gs_cpu.cpp Extras::budget)
    Returns {name: array like params[name]}: sum_j |d out_j / d param| * budget -- what the parameter's gradient can
    move by when the blend's rows move by their budgets."""
    import torch
    names = list(params)
    out = {k: np.zeros(tuple(params[k].shape), np.float64) for k in names}
    for oname, o in outputs.items():
        b = np.asarray(budgets[oname], np.float64)
        for j in range(o.shape[1]):
            cot = torch.zeros_like(o)
            cot[:, j] = 1.0
            grads = torch.autograd.grad(o, [params[k] for k in names], cot, retain_graph=True, allow_unused=True)
            for k, g_ in zip(names, grads):
                if g_ is not None:
                    out[k] += np.abs(g_.numpy()) * b.reshape((-1,) + (1,) * (g_.dim() - 1))
    return out


def blend_cotangents(mode, w_render, w_alpha, ref_render, ref_alpha):
    """Cotangents of the UN-normalised blend (what oracle/gs_cpu.cpp's backward takes) from those of the frame.
    "ED" modes: the last channel left as D / max(alpha, 1e-10), so d/dD = v / a and d/dalpha -= v * ED / a."""
    v_r = np.array(w_render, dtype=np.float64, copy=True)
    v_a = np.array(w_alpha, dtype=np.float64, copy=True)
    if mode in ("ED", "RGB+ED"):
        a = np.maximum(np.asarray(ref_alpha, np.float64), 1e-10)
        ed = np.asarray(ref_render, np.float64)[..., -1] / a          # the port returns the depth SUM
        v_a = v_a - np.where(np.asarray(ref_alpha) > 1e-10, v_r[..., -1] * ed / a, 0.0)
        v_r[..., -1] = v_r[..., -1] / a
    return v_r.astype(np.float32), v_a.astype(np.float32)


def oracle_budgets(g, viewmat_f32, K_f32, W, H, deg, mode, w_render, w_alpha, flip_eps, n_threads=0, radius_rule="classic"):
    """fp64 port: blend gradients, touched mask and the per-row flip budgets for loss = <w_render, frame> + <w_alpha, alpha>.
    Returns the port's info dict (g_means2d, g_conics, g_feats, g_opacities, budget [N,4], touched, radii, ...)."""
    from oracle import cpu_ref
    with_depth = mode != "RGB"
    ref, ra, _ = cpu_ref.render_f64(g.means, g.quats, g.scales, g.opacities, g.sh_coeffs, viewmat_f32, K_f32, W, H, deg,
                                    with_depth=with_depth, margins=False, n_threads=n_threads, radius_rule=radius_rule)
    v_r, v_a = blend_cotangents(mode, w_render, w_alpha, ref, ra)
    _, _, info = cpu_ref.render_f64(g.means, g.quats, g.scales, g.opacities, g.sh_coeffs, viewmat_f32, K_f32, W, H, deg,
                                    with_depth=with_depth, margins=True, v_render=v_r, v_alpha=v_a, want_projected=True,
                                    flip_eps=flip_eps, want_touched=True, want_budget=True, n_threads=n_threads,
                                    radius_rule=radius_rule)
    return info


def parameter_budgets(g, viewmat_f32, K_f32, W, H, deg, with_depth, budget, radius_rule="classic"):
    """Row budgets of d loss / d {means, quats, scales, colors} from the blend's (budget [N,4]: means2d, conics, feats,
    opacity) through the fp64 torch oracle's projection and SH colour (absolute Jacobian, chained_budget)."""
    import torch
    from oracle import gs_oracle_torch as OT
    d = lambda x, grad=False: torch.tensor(np.asarray(x, dtype=np.float64), requires_grad=grad)
    P = {"means": d(g.means, True), "quats": d(g.quats, True), "scales": d(g.scales, True),
         "colors": d(g.sh_coeffs[:, :(deg + 1) ** 2], True)}
    vm, K = d(viewmat_f32), d(K_f32)
    pr = OT.project(P["means"], P["quats"], P["scales"], vm, K, W, H, radius_rule=radius_rule, opacities=d(g.opacities))
    vis = (pr["radii"].reshape(len(g), -1)[:, 0] > 0).to(torch.float64)[:, None]
    campos = -vm[:3, :3].T @ vm[:3, 3]
    rgb = torch.clamp(OT.spherical_harmonics(deg, P["means"] - campos, P["colors"]) + 0.5, min=0.0) * vis
    feats = torch.cat([rgb, pr["depths"][:, None]], dim=-1) if with_depth else rgb
    return chained_budget(P, {"means2d": pr["means2d"], "conics": pr["conics"], "feats": feats},
                          {"means2d": budget[:, 0], "conics": budget[:, 1], "feats": budget[:, 2]})
