"""Gaussian scene container, the nerfstudio/Inria `.ply` layout, and the world-frame
alignment the reference describes.

`.ply` layout (what `ns-export gaussian-splat` writes, README.md:75 of the reference):
binary little-endian, one `vertex` element, float properties `x y z nx ny nz f_dc_0..2
f_rest_0..(3(K-1)-1) opacity scale_0..2 rot_0..3`.  Stored values are pre-activation
(opacity logit, log scale, un-normalised wxyz quaternion); `f_rest` is channel-major on
disk ([N,3,K-1]).  SH-degree-0 exports may carry uint8 `red green blue` instead.  The
header is parsed by property *name* so extra / missing properties are tolerated.

Dataparser un-normalisation follows `load_ns_point_cloud`
(/root/reference/Articulation/utils/nerf2physic_utils.py:68-74): build the 4x4
[[transform],[0,0,0,1/scale]], invert, apply.
"""
from __future__ import annotations

import json
from dataclasses import dataclass
from typing import Optional

import numpy as np

SH_C0 = 0.2820947917738781


@dataclass
class Gaussians:
    """Pre-activation parameters, float32, host (numpy) side.

    means [N,3]; log_scales [N,3]; quats [N,4] wxyz un-normalised; opacity_logits [N];
    sh_dc [N,3]; sh_rest [N,K-1,3] (coefficient-major, channel-minor)."""

    means: np.ndarray
    log_scales: np.ndarray
    quats: np.ndarray
    opacity_logits: np.ndarray
    sh_dc: np.ndarray
    sh_rest: np.ndarray

    def __post_init__(self):
        f = lambda a: np.ascontiguousarray(a, dtype=np.float32)
        self.means, self.log_scales, self.quats = f(self.means), f(self.log_scales), f(self.quats)
        self.opacity_logits, self.sh_dc = f(self.opacity_logits).reshape(-1), f(self.sh_dc)
        self.sh_rest = f(self.sh_rest).reshape(self.means.shape[0], -1, 3)
        n = self.means.shape[0]
        for name, a, shape in (("log_scales", self.log_scales, (n, 3)),
                               ("quats", self.quats, (n, 4)),
                               ("opacity_logits", self.opacity_logits, (n,)),
                               ("sh_dc", self.sh_dc, (n, 3))):
            if a.shape != shape:
                raise ValueError(f"{name}: expected shape {shape}, got {a.shape}")
        k = self.sh_rest.shape[1] + 1
        if int(round(k ** 0.5)) ** 2 != k:
            raise ValueError(f"sh_rest holds {k - 1} coefficients; K={k} is not a square")

    def __len__(self) -> int:
        return self.means.shape[0]

    def permuted(self, order: np.ndarray) -> "Gaussians":
        """The same scene with its Gaussians in another index order (`order` = a permutation of range(N))."""
        o = np.asarray(order)
        return Gaussians(self.means[o], self.log_scales[o], self.quats[o], self.opacity_logits[o], self.sh_dc[o],
                         self.sh_rest[o])

    def sorted_by_locality(self, bits: int = 10) -> "Gaussians":
        """The same scene in Morton (Z-curve) order of the means: Gaussians that are neighbours in space become
        neighbours in memory, so a tile's list entries gather from a narrow index range, a binning workgroup's pairs
        fall into few tile groups, and waves of culled Gaussians are culled together.  A one-off at load time for a
        static scene; the render differs from the unsorted scene's only where two Gaussians tie in depth to the last
        bit (ties go by index)."""
        lo, hi = self.means.min(axis=0), self.means.max(axis=0)
        q = np.clip(((self.means - lo) / np.maximum(hi - lo, 1e-30) * ((1 << bits) - 1)).astype(np.uint64), 0, (1 << bits) - 1)
        code = np.zeros(len(self), dtype=np.uint64)
        for b in range(bits):
            for axis in range(3):
                code |= ((q[:, axis] >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b + axis)
        return self.permuted(np.argsort(code, kind="stable"))

    @property
    def sh_degree(self) -> int:
        return int(round((self.sh_rest.shape[1] + 1) ** 0.5)) - 1

    # -- activations (what splatfacto applies before calling the rasteriser) -----------
    @property
    def scales(self) -> np.ndarray:
        return np.exp(self.log_scales)

    @property
    def opacities(self) -> np.ndarray:
        return (1.0 / (1.0 + np.exp(-self.opacity_logits.astype(np.float64)))).astype(np.float32)

    @property
    def sh_coeffs(self) -> np.ndarray:
        """[N,K,3] = cat(f_dc[:,None], f_rest)."""
        return np.concatenate([self.sh_dc[:, None, :], self.sh_rest], axis=1)

    def to_torch(self, device="cuda", sh_degree: Optional[int] = None):
        """Post-activation tensors in `rasterization(...)` argument order."""
        import torch
        deg = self.sh_degree if sh_degree is None else sh_degree
        k = (deg + 1) ** 2
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)
        return dict(means=t(self.means), quats=t(self.quats), scales=t(self.scales),
                    opacities=t(self.opacities), colors=t(self.sh_coeffs[:, :k]),
                    sh_degree=deg)

    # -- world-frame alignment ---------------------------------------------------------
    def transformed(self, rotation: np.ndarray, translation: np.ndarray,
                    scale: float = 1.0) -> "Gaussians":
        """Apply the similarity x -> scale * R x + t to means, orientations and extents, and
        rotate the view-dependent colour with it: the SH coefficients of every degree l are
        multiplied by the (2l+1)x(2l+1) real-SH rotation matrix of R, so that the transformed
        scene seen from a transformed camera renders the same image."""
        R = np.asarray(rotation, dtype=np.float64)
        means = scale * (self.means.astype(np.float64) @ R.T) + np.asarray(translation)
        qR = _rotmat_to_quat(R)
        quats = _quat_mul(qR[None], self.quats.astype(np.float64))
        rest = self.sh_rest
        if rest.shape[1]:
            coeffs = self.sh_coeffs.astype(np.float64)
            for l, M in enumerate(sh_rotation_matrices(R, self.sh_degree)):
                lo, hi = l * l, (l + 1) * (l + 1)
                coeffs[:, lo:hi, :] = np.einsum("kj,njc->nkc", M, coeffs[:, lo:hi, :])
            rest = coeffs[:, 1:, :]
        return Gaussians(means, self.log_scales + np.float32(np.log(scale)), quats,
                         self.opacity_logits, self.sh_dc, rest)

    def undo_dataparser_transform(self, transform: np.ndarray, scale: float) -> "Gaussians":
        """nerfstudio-normalised space -> original world (nerf2physic_utils.py:68-74):
        x_ns = scale * (T x_world)  =>  x_world = T^-1 (x_ns / scale)."""
        T = np.concatenate([np.asarray(transform, dtype=np.float64).reshape(3, 4),
                            np.array([[0.0, 0.0, 0.0, 1.0 / scale]])], axis=0)
        Tinv = np.linalg.inv(T)
        Tinv = Tinv / Tinv[3, 3]
        A, t = Tinv[:3, :3], Tinv[:3, 3]
        s = float(np.cbrt(np.linalg.det(A)))
        return self.transformed(A / s, t, s)


def load_dataparser_transforms(path: str):
    """`dataparser_transforms.json` -> (transform 3x4, scale); same keys as
    `parse_dataparser_transforms_json` (nerf2physic_utils.py:55-61)."""
    with open(path, "r") as f:
        d = json.load(f)
    return np.asarray(d["transform"], dtype=np.float64), float(d["scale"])


# --------------------------------------------------------------------------------------
# .ply IO
# --------------------------------------------------------------------------------------
_PLY_TYPES = {"float": "<f4", "float32": "<f4", "double": "<f8", "float64": "<f8",
              "uchar": "u1", "uint8": "u1", "char": "i1", "int8": "i1",
              "short": "<i2", "int16": "<i2", "ushort": "<u2", "uint16": "<u2",
              "int": "<i4", "int32": "<i4", "uint": "<u4", "uint32": "<u4"}


def load_ply(path: str) -> Gaussians:
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, n, props, in_vertex = None, None, [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] == "comment":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    n = int(tok[2])
                elif n is None:
                    raise ValueError(f"{path}: element '{tok[1]}' precedes 'vertex'")
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError(f"{path}: list property in vertex element")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if n is None:
            raise ValueError(f"{path}: no vertex element")
        if fmt == "binary_little_endian":
            data = np.frombuffer(f.read(n * np.dtype(props).itemsize), dtype=np.dtype(props), count=n)
        elif fmt == "ascii":
            raw = np.loadtxt(f, max_rows=n, ndmin=2)
            data = np.zeros(n, dtype=np.dtype(props))
            for i, (name, _) in enumerate(props):
                data[name] = raw[:, i]
        else:
            raise ValueError(f"{path}: unsupported PLY format {fmt!r}")
    names = set(data.dtype.names)
    col = lambda k: data[k].astype(np.float32)
    need = ["x", "y", "z", "opacity", "scale_0", "scale_1", "scale_2",
            "rot_0", "rot_1", "rot_2", "rot_3"]
    missing = [k for k in need if k not in names]
    if missing:
        raise ValueError(f"{path}: missing properties {missing}")
    means = np.stack([col("x"), col("y"), col("z")], -1)
    if "f_dc_0" in names:
        dc = np.stack([col(f"f_dc_{i}") for i in range(3)], -1)
    elif {"red", "green", "blue"} <= names:       # degree-0 uint8 colour export
        rgb = np.stack([col("red"), col("green"), col("blue")], -1) / 255.0
        dc = (rgb - 0.5) / SH_C0
    else:
        raise ValueError(f"{path}: neither f_dc_* nor red/green/blue present")
    n_rest = sum(1 for k in names if k.startswith("f_rest_"))
    if n_rest % 3:
        raise ValueError(f"{path}: {n_rest} f_rest_* properties is not a multiple of 3")
    if n_rest:
        rest = np.stack([col(f"f_rest_{i}") for i in range(n_rest)], -1)
        rest = rest.reshape(n, 3, n_rest // 3).transpose(0, 2, 1)       # channel-major on disk
    else:
        rest = np.zeros((n, 0, 3), np.float32)
    g = Gaussians(means, np.stack([col(f"scale_{i}") for i in range(3)], -1),
                  np.stack([col(f"rot_{i}") for i in range(4)], -1), col("opacity"), dc, rest)
    keep = np.isfinite(np.concatenate([g.means, g.log_scales, g.quats, g.opacity_logits[:, None],
                                       g.sh_dc, g.sh_rest.reshape(n, -1)], axis=1)).all(axis=1)
    if not keep.all():                                                   # exporters drop NaN/Inf rows
        g = Gaussians(g.means[keep], g.log_scales[keep], g.quats[keep], g.opacity_logits[keep],
                      g.sh_dc[keep], g.sh_rest[keep])
    return g


def save_ply(path: str, g: Gaussians) -> None:
    n, kr = len(g), g.sh_rest.shape[1]
    names = (["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(3)]
             + [f"f_rest_{i}" for i in range(3 * kr)] + ["opacity"]
             + [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)])
    arr = np.zeros((n, len(names)), dtype="<f4")
    arr[:, 0:3] = g.means
    arr[:, 6:9] = g.sh_dc
    arr[:, 9:9 + 3 * kr] = g.sh_rest.transpose(0, 2, 1).reshape(n, 3 * kr)
    o = 9 + 3 * kr
    arr[:, o] = g.opacity_logits
    arr[:, o + 1:o + 4] = g.log_scales
    arr[:, o + 4:o + 8] = g.quats
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % n
    header += "".join(f"property float {k}\n" for k in names) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(arr.tobytes())


# --------------------------------------------------------------------------------------
# synthetic scenes (identical draw order everywhere)
# --------------------------------------------------------------------------------------
def synthetic_scene(n: int, log_scale_mean: float, sh_degree: int = 3, seed: int = 0,
                    extent: float = 3.0) -> Gaussians:
    """PCG64(seed) draws, in this order: means U(-extent,extent)^3; log_scales
    N(mu,0.4^2); quats N(0,1)^4 normalised; opacity logit N(0,1.5^2); f_dc N(0,1);
    f_rest N(0,0.1^2) in the on-disk channel-major order."""
    rng = np.random.default_rng(seed)
    means = rng.uniform(-extent, extent, size=(n, 3))
    log_scales = rng.normal(log_scale_mean, 0.4, size=(n, 3))
    quats = rng.normal(0.0, 1.0, size=(n, 4))
    quats /= np.linalg.norm(quats, axis=1, keepdims=True)
    logits = rng.normal(0.0, 1.5, size=n)
    dc = rng.normal(0.0, 1.0, size=(n, 3))
    kr = (sh_degree + 1) ** 2 - 1
    rest = rng.normal(0.0, 0.1, size=(n, 3, kr)).transpose(0, 2, 1)
    return Gaussians(means, log_scales, quats, logits, dc, rest)


def synthetic_scene_heavy_tailed(n: int, log_scale_mean: float = float(np.log(0.0045)), sh_degree: int = 3, seed: int = 0,
                                 extent: float = 3.0, n_clusters: int = 96, n_screen_filling: int = 6,
                                 n_needles: int = 4000) -> Gaussians:
    """A scene shaped like an EXPORT rather than like SURVEY.md 8(d)'s i.i.d. cloud (NOT a BASELINE.json config: a
    no-cliff check for the per-tile sort's refinement levels, the backward's segment tables and the list capacities):
      * clustered means: `n_clusters` centres U(-extent, extent)^3 with Zipf-like shares (the largest holds ~1/6 of the
        scene), each an anisotropic blob -- one axis flattened by up to 10x, spread log-normal around 0.25 -- plus 8 % of
        uniform floaters;
      * heavy-tailed extents: log scale = N(mu, 1.2^2) shared by the three axes + N(0, 0.35^2) per axis (the 99.9th
        percentile is ~40x the median; SURVEY 8(d) draws N(mu, 0.4^2));
      * `n_needles` needle-like Gaussians (the last rows but `n_screen_filling`): axis ratio 100-400 : 1, opaque;
      * `n_screen_filling` Gaussians (the last rows) of extent 1.5-4 around the origin, opacity 0.04-0.2: at a camera 7
        units away each covers (nearly) every tile of the frame.
    PCG64(seed), draws in the order of the code below; the remaining attributes as in synthetic_scene."""
    rng = np.random.default_rng(seed)
    n_special = n_needles + n_screen_filling
    if n <= n_special:
        raise ValueError(f"n = {n} must exceed the {n_special} needle-like and screen-filling Gaussians")
    nb = n - n_special
    centres = rng.uniform(-extent, extent, size=(n_clusters, 3))
    share = 1.0 / np.arange(1, n_clusters + 1) ** 0.9
    share /= share.sum()
    spread = np.exp(rng.normal(np.log(0.25), 0.6, size=n_clusters))
    flat_axis = rng.integers(0, 3, size=n_clusters)
    flat = rng.uniform(0.1, 1.0, size=n_clusters)
    rot = rng.normal(size=(n_clusters, 3, 3))
    rot = np.linalg.qr(rot)[0]                                          # a random frame per cluster
    which = rng.choice(n_clusters, size=nb, p=share)
    local = rng.normal(size=(nb, 3)) * spread[which, None]
    local[np.arange(nb), flat_axis[which]] *= flat[which]
    means = centres[which] + np.einsum("nij,nj->ni", rot[which], local)
    floaters = rng.random(nb) < 0.08
    means[floaters] = rng.uniform(-extent, extent, size=(int(floaters.sum()), 3))
    log_scales = rng.normal(log_scale_mean, 1.2, size=(nb, 1)) + rng.normal(0.0, 0.35, size=(nb, 3))
    logits = rng.normal(0.0, 1.5, size=nb)
    # needle-like: one long axis, two thin ones, opaque
    nd_means = rng.uniform(-0.8 * extent, 0.8 * extent, size=(n_needles, 3))
    long_axis = np.exp(rng.uniform(np.log(0.15), np.log(0.6), size=n_needles))
    ratio = rng.uniform(100.0, 400.0, size=n_needles)
    nd_scales = np.stack([long_axis, long_axis / ratio, long_axis / ratio], axis=1)
    nd_logits = rng.normal(3.0, 1.0, size=n_needles)
    # screen-filling: large, faint, around the origin
    sf_means = rng.normal(0.0, 0.3, size=(n_screen_filling, 3))
    sf_scales = rng.uniform(1.5, 4.0, size=(n_screen_filling, 3))
    sf_opac = rng.uniform(0.04, 0.2, size=n_screen_filling)
    means = np.concatenate([means, nd_means, sf_means])
    log_scales = np.concatenate([log_scales, np.log(nd_scales), np.log(sf_scales)])
    logits = np.concatenate([logits, nd_logits, np.log(sf_opac / (1.0 - sf_opac))])
    quats = rng.normal(0.0, 1.0, size=(n, 4))
    quats /= np.linalg.norm(quats, axis=1, keepdims=True)
    dc = rng.normal(0.0, 1.0, size=(n, 3))
    kr = (sh_degree + 1) ** 2 - 1
    rest = rng.normal(0.0, 0.1, size=(n, 3, kr)).transpose(0, 2, 1)
    return Gaussians(means, log_scales, quats, logits, dc, rest)


# --------------------------------------------------------------------------------------
# rotating spherical harmonics (host side; used by the world-frame alignment)
# --------------------------------------------------------------------------------------
def _sh_basis_np(degree: int, dirs: np.ndarray) -> np.ndarray:
    """Real SH basis in the renderer's ordering and sign convention,
    float64, for unit `dirs` [M,3] -> [M,(degree+1)^2]."""
    x, y, z = dirs[:, 0], dirs[:, 1], dirs[:, 2]
    Y = [np.full_like(x, SH_C0)]
    if degree >= 1:
        c1 = 0.48860251190292
        Y += [-c1 * y, c1 * z, -c1 * x]
    if degree >= 2:
        z2, fC1, fS1 = z * z, x * x - y * y, 2 * x * y
        t = -1.092548430592079 * z
        Y += [0.5462742152960395 * fS1, t * y, 0.9461746957575601 * z2 - 0.3153915652525201, t * x,
              0.5462742152960395 * fC1]
    if degree >= 3:
        u = -2.285228997322329 * z2 + 0.4570457994644658
        w = 1.445305721320277 * z
        fC2, fS2 = x * fC1 - y * fS1, x * fS1 + y * fC1
        Y += [-0.5900435899266435 * fS2, w * fS1, u * y, z * (1.865881662950577 * z2 - 1.119528997770346),
              u * x, w * fC1, -0.5900435899266435 * fC2]
    return np.stack(Y, axis=-1)


_SH_FIT_CACHE: dict = {}


def _sh_fit_basis(degree: int):
    """Fixed sample directions and, per band, the pseudo-inverse of the basis evaluated there
    (they do not depend on the rotation, so a fit is two small matrix products)."""
    hit = _SH_FIT_CACHE.get(degree)
    if hit is None:
        rng = np.random.default_rng(12345)
        d = rng.normal(size=(64, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        Y_new = _sh_basis_np(degree, d)               # basis at the new-frame directions
        pinv = [np.linalg.pinv(Y_new[:, l * l:(l + 1) * (l + 1)]) for l in range(degree + 1)]
        hit = _SH_FIT_CACHE[degree] = (d, pinv)
    return hit


def sh_rotation_matrices(R: np.ndarray, degree: int):
    """Per-degree matrices M_l with  c'_l = M_l c_l  such that
    sum_k Y_k(d) c'_k == sum_k Y_k(R^T d) c_k  for every direction d (the colour field rotated
    by R).  Each degree-l band is invariant under rotation, so M_l is found exactly by a
    least-squares fit on a fixed set of sample directions."""
    d, pinv = _sh_fit_basis(degree)
    Y_old = _sh_basis_np(degree, d @ np.asarray(R, dtype=np.float64))   # rows are (R^T d)^T
    # Y_new[:, band] @ M = Y_old[:, band]  ->  c' = M c
    return [pinv[l] @ Y_old[:, l * l:(l + 1) * (l + 1)] for l in range(degree + 1)]


def _rotmat_to_quat(R: np.ndarray) -> np.ndarray:
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = [0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s]
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = [0.0] * 4
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    return np.asarray(q, dtype=np.float64)


def _quat_mul(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    aw, ax, ay, az = (a[..., i] for i in range(4))
    bw, bx, by, bz = (b[..., i] for i in range(4))
    return np.stack([aw * bw - ax * bx - ay * by - az * bz,
                     aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw], axis=-1)
