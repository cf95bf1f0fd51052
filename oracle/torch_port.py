"""TEST INFRASTRUCTURE -- not product code.

Pure-torch, op-for-op CPU restatement of the NeRF-SOS volumetric rendering hot path.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this file; the product package (``nerf-sos_amd/``) never does.

What it restates (reference file:line, relative to the upstream repo root):

* ``stratified_z``        <- models/sampler.py:25-74    (StratifiedSampler.forward)
* ``posenc``              <- models/embedder.py:34-48   (PositionEncoder.forward)
* ``mlp_forward``         <- models/nerf_mlp.py:67-100  (MLP.forward)
* ``point_query``         <- models/nerf_mlp.py:179-215 (NeRFMLP.forward, chunk loop)
* ``composite``           <- models/renderer.py:21-85   (VolumetricRenderer.forward)
* ``sample_pdf``          <- models/sampler.py:91-134   (ImportanceSampler.sample_pdf)
* ``importance_z``        <- models/sampler.py:136-170  (ImportanceSampler.forward)
* ``render_rays``         <- models/nerf_net.py:71-130  (NeRFNet.render_rays)
* ``render``              <- models/nerf_net.py:132-195 (NeRFNet.forward)
* ``init_state_dict``     <- models/nerf_net.py:22-56 + models/nerf_mlp.py:40-64 (construction order)

Parity pin: ``tests/golden/make_goldens.py`` imports the real reference in the build
container, asserts this port is bit-identical to it on CPU for every committed case, and
writes the reference's outputs as fixtures.  The same ATen ops are used in the same order so
equality is bitwise on the same torch build.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass
from typing import Callable, Dict, Optional, Tuple

import torch

Tensor = torch.Tensor


@dataclass(frozen=True)
class PortConfig:
    """Construction kwargs of the reference NeRFNet that reach the arithmetic
    (models/nerf_net.py:22-24; run_nerf.py:301-305)."""

    n_samples: int = 64
    n_importance: int = 128
    multires: int = 10
    multires_views: int = 4
    net_depth: int = 8
    net_width: int = 256
    skip: int = 4
    use_semantics: bool = False
    sem_dim: int = 2
    sem_with_coord: bool = False
    # the constructor arguments no shipped config changes (round 4: the generic-architecture kernel)
    use_viewdirs: bool = True     # models/nerf_net.py:23 `viewdirs`
    use_embed: bool = True
    sem_layer: int = 2
    sem_with_geo: bool = False
    white_bkgd: bool = False
    ray_chunk: int = 1024 * 32
    pts_chunk: int = 1024 * 64

    @property
    def xyz_dim(self) -> int:
        return 3 + 6 * self.multires if self.use_embed else 3

    @property
    def dir_dim(self) -> int:
        if not self.use_viewdirs:
            return 0
        return 3 + 6 * self.multires_views if self.use_embed else 3


# --------------------------------------------------------------------------- weights
def _linear_shapes(cfg: PortConfig):
    """(name, out, in) in the reference's construction order (models/nerf_mlp.py:40-64)."""
    W, X, V = cfg.net_width, cfg.xyz_dim, cfg.dir_dim
    out = []
    for i in range(cfg.net_depth):
        fan_in = X if i == 0 else (W + X if i == cfg.skip + 1 else W)
        out.append((f"pts_linears.{i}", W, fan_in))
    if cfg.use_viewdirs:
        out.append(("alpha_linear", 1, W))
        out.append(("feature_linear", W, W))
        out.append(("views_linears.0", W // 2, V + W))
        out.append(("rgb_linear", 3, W // 2))
    else:
        out.append(("output_linear", 4, W))
    if cfg.use_semantics:
        sem_in = W + X if cfg.sem_with_coord else W
        if cfg.sem_layer <= 2:
            out.append(("semantic_linear.0", W // 2, sem_in))
            out.append(("semantic_linear.2", cfg.sem_dim, W // 2))
        else:     # Sequential(Linear, ReLU, *fc_block(W, W) x (sem_layer - 3), Linear(W, W/2), ReLU, Linear(W/2, sem_dim)): nerf_mlp.py:63
            out.append(("semantic_linear.0", W, sem_in))
            n = cfg.sem_layer - 3
            for k in range(n):
                out.append((f"semantic_linear.{2 + k}.0", W, W))
            out.append((f"semantic_linear.{2 + n}", W // 2, W))
            out.append((f"semantic_linear.{4 + n}", cfg.sem_dim, W // 2))
        if cfg.sem_with_geo:
            out.append(("geo_map_sem.0", W // 2, 1))
            out.append(("geo_map_sem.2", cfg.sem_dim, W // 2))
    return out


def _sem_chain(cfg: PortConfig):
    """Names of the semantic head's Linear modules in forward order."""
    return [n for n, _, _ in _linear_shapes(cfg) if n.startswith("semantic_linear.")]


def init_state_dict(cfg: PortConfig, seed: Optional[int] = 0) -> "OrderedDict[str, Tensor]":
    """Default-initialised weights, created with ``nn.Linear`` in the reference's RNG-consuming
    order (coarse net, then fine net), so that under the same ``torch.manual_seed`` the result
    equals the reference's ``state_dict()`` bit for bit."""
    if seed is not None:
        torch.manual_seed(seed)
    sd: "OrderedDict[str, Tensor]" = OrderedDict()
    nets = ["nerf"] + (["nerf_fine"] if cfg.n_importance > 0 else [])
    for net in nets:
        for name, fan_out, fan_in in _linear_shapes(cfg):
            lin = torch.nn.Linear(fan_in, fan_out)
            sd[f"{net}.mlp.{name}.weight"] = lin.weight.detach().clone()
            sd[f"{net}.mlp.{name}.bias"] = lin.bias.detach().clone()
    if cfg.n_importance == 0:
        # models/nerf_net.py:49: nerf_fine IS nerf, so the state dict lists the tensors twice
        for k in [k for k in sd if k.startswith("nerf.")]:
            sd["nerf_fine." + k[len("nerf."):]] = sd[k]
    return sd


def make_peaky(sd: Dict[str, Tensor], gain: float = 40.0, shift: float = -1.5) -> Dict[str, Tensor]:
    """Deterministic transform of a default-init state dict that makes the density field
    spiky (few samples carry almost all the weight), which exercises ``sample_pdf``'s
    search/lerp and the transmittance product far from the flat default-init regime."""
    out = {k: v.clone() for k, v in sd.items()}
    for net in ("nerf", "nerf_fine"):
        k = f"{net}.mlp.alpha_linear"
        if k + ".weight" in out:
            out[k + ".weight"] = out[k + ".weight"] * gain
            out[k + ".bias"] = out[k + ".bias"] * gain + shift
    return out


# --------------------------------------------------------------------------- stages
def stratified_z(near: Tensor, far: Tensor, n_samples: int, t_rand: Optional[Tensor]) -> Tensor:
    """models/sampler.py:46-68.  ``t_rand`` None <=> perturb == 0."""
    n_rays = near.shape[0]
    t_vals = torch.linspace(0.0, 1.0, steps=n_samples, device=near.device)
    z_vals = near * (1.0 - t_vals) + far * t_vals
    z_vals = z_vals.expand([n_rays, n_samples])
    if t_rand is not None:
        mids = 0.5 * (z_vals[..., 1:] + z_vals[..., :-1])
        upper = torch.cat([mids, z_vals[..., -1:]], -1)
        lower = torch.cat([z_vals[..., :1], mids], -1)
        z_vals = lower + (upper - lower) * t_rand
    return z_vals


def ray_points(rays_o: Tensor, rays_d: Tensor, z_vals: Tensor) -> Tensor:
    """models/sampler.py:70,166."""
    return rays_o[..., None, :] + rays_d[..., None, :] * z_vals[..., :, None]


def posenc(x: Tensor, n_freqs: int) -> Tensor:
    """models/embedder.py:34-48: [x, sin(f0 x), cos(f0 x), sin(f1 x), ...], f_k = 2**k."""
    freq_bands = 2.0 ** torch.linspace(0.0, n_freqs - 1, steps=n_freqs)
    embeds = []
    for fn in (torch.sin, torch.cos):
        x_freq = x[..., None].expand(x.shape + (n_freqs,)) * freq_bands
        embeds.append(fn(x_freq).transpose(-1, -2))
    e = torch.stack(embeds, -2).reshape(x.shape[:-1] + (-1,))
    return torch.cat([x, e], -1)


def mlp_forward(sd: Dict[str, Tensor], prefix: str, x: Tensor, cfg: PortConfig,
                tap: Optional[Callable[[str, Tensor], None]] = None) -> Tensor:
    """models/nerf_mlp.py:67-100 on an already-encoded input [P, xyz_dim + dir_dim]."""
    F = torch.nn.functional

    def lin(name, h):
        return F.linear(h, sd[f"{prefix}.mlp.{name}.weight"], sd[f"{prefix}.mlp.{name}.bias"])

    input_pts, input_views = torch.split(x, [cfg.xyz_dim, cfg.dir_dim], dim=-1)
    h = input_pts
    for i in range(cfg.net_depth):
        h = F.relu(lin(f"pts_linears.{i}", h))
        if tap:
            tap(f"h{i}", h)
        if i == cfg.skip:
            h = torch.cat([input_pts, h], -1)
    if not cfg.use_viewdirs:
        return lin("output_linear", h)                               # :97-98
    alpha = lin("alpha_linear", h)
    sem = None
    if cfg.use_semantics:
        s = torch.cat([h, input_pts], dim=-1) if cfg.sem_with_coord else h
        chain = _sem_chain(cfg)
        for k, name in enumerate(chain):
            s = lin(name, s)
            if k < len(chain) - 1:
                s = F.relu(s)
                if tap and k == 0:
                    tap("sem_hidden", s)
        sem = s
        if cfg.sem_with_geo:                                         # :81-83
            mapping = lin("geo_map_sem.2", F.relu(lin("geo_map_sem.0", alpha)))
            sem = sem * mapping
    feature = lin("feature_linear", h)
    if tap:
        tap("feature", feature)
    v = F.relu(lin("views_linears.0", torch.cat([feature, input_views], -1)))
    if tap:
        tap("view_hidden", v)
    rgb = lin("rgb_linear", v)
    return torch.cat([rgb, alpha] + ([sem] if sem is not None else []), -1)


def point_query(sd: Dict[str, Tensor], prefix: str, pts: Tensor, viewdirs: Optional[Tensor], cfg: PortConfig) -> Tensor:
    """models/nerf_mlp.py:179-215: flatten, chunk, encode both inputs (unless use_embed is off), run the MLP."""
    flat = pts.reshape(-1, pts.shape[-1])
    dirs = viewdirs.reshape(-1, viewdirs.shape[-1]) if cfg.use_viewdirs else None
    enc_x = (lambda t: posenc(t, cfg.multires)) if cfg.use_embed else (lambda t: t)
    enc_v = (lambda t: posenc(t, cfg.multires_views)) if cfg.use_embed else (lambda t: t)
    outs = []
    for i in range(0, flat.shape[0], cfg.pts_chunk):
        e = enc_x(flat[i:i + cfg.pts_chunk])
        if dirs is not None:
            e = torch.cat([e, enc_v(dirs[i:i + cfg.pts_chunk])], -1)
        outs.append(mlp_forward(sd, prefix, e, cfg))
    out = torch.cat(outs, 0)
    return out.reshape(list(pts.shape[:-1]) + [out.shape[-1]])


def point_query_lp(sd: Dict[str, Tensor], prefix: str, pts: Tensor, viewdirs: Tensor, cfg: PortConfig,
                   dtype: torch.dtype, lp16: bool = False) -> Tensor:
    """Emulation of the reduced-precision kernels (nerf-sos_amd/csrc/mlp_lp.hip, mlp_lp8.hip; lp16=True: mlp_lp16.hip), NOT of
    the reference: everything that enters an MFMA is rounded to `dtype` (encodings, MFMA weights, activations after ReLU, the
    feature vector), products are accumulated in fp64 here (fp32 on the GPU).
      lp16=False: every bias is an MFMA operand (rounded); the sigma head uses 16-bit weights on the 16-bit activations, the
                  rgb / semantic output heads are fp32 on the un-rounded fp32 hidden layers.
      lp16=True:  the biases of the hidden layers 1-4, 6-7 and of feature_linear are fp32 accumulator initialisers (those of
                  layers 0 and 5, of semantic_linear.0 and of views_linears.0 ride in an operand's pad column: rounded); ALL
                  three output heads are MFMAs: 16-bit weights on the 16-bit-rounded hidden activations, fp32 biases."""
    q = lambda t: t.to(dtype).double()  # noqa: E731
    W = lambda n: sd[f"{prefix}.mlp.{n}.weight"]  # noqa: E731
    B = lambda n: sd[f"{prefix}.mlp.{n}.bias"]  # noqa: E731
    qb = (lambda t: t.double()) if lp16 else q        # a hidden layer's bias  # noqa: E731
    flat = pts.reshape(-1, 3)
    dirs = viewdirs.reshape(-1, 3)
    ex, ed = q(posenc(flat, cfg.multires)), q(posenc(dirs, cfg.multires_views))
    X = cfg.xyz_dim
    h = q(torch.relu(ex @ q(W("pts_linears.0")).T + q(B("pts_linears.0"))).float())
    for i in range(1, cfg.net_depth):
        w = q(W(f"pts_linears.{i}"))
        if i == cfg.skip + 1:
            z = h @ w[:, X:].T + ex @ w[:, :X].T + q(B(f"pts_linears.{i}"))
        else:
            z = h @ w.T + qb(B(f"pts_linears.{i}"))
        h = q(torch.relu(z).float())
    sigma = h @ q(W("alpha_linear")).T + B("alpha_linear").double()
    outs = []
    if cfg.use_semantics:
        w = q(W("semantic_linear.0"))
        z = h @ w[:, :cfg.net_width].T + q(B("semantic_linear.0"))
        if cfg.sem_with_coord:
            z = z + ex @ w[:, cfg.net_width:].T
        hid = torch.relu(z).float()
        if lp16:
            sem = q(hid) @ q(W("semantic_linear.2")).T + B("semantic_linear.2").double()
        else:
            sem = hid.double() @ W("semantic_linear.2").double().T + B("semantic_linear.2").double()
        outs = [sem]
    feat = q((h @ q(W("feature_linear")).T + qb(B("feature_linear"))).float())
    w = q(W("views_linears.0"))
    z = feat @ w[:, :cfg.net_width].T + ed @ w[:, cfg.net_width:].T + q(B("views_linears.0"))
    vh = torch.relu(z).float()
    if lp16:
        rgb = q(vh) @ q(W("rgb_linear")).T + B("rgb_linear").double()
    else:
        rgb = vh.double() @ W("rgb_linear").double().T + B("rgb_linear").double()
    out = torch.cat([rgb, sigma] + outs, -1).float()
    return out.reshape(list(pts.shape[:-1]) + [out.shape[-1]])


def composite(raw: Tensor, z_vals: Tensor, rays_d: Tensor, noise: Optional[Tensor], cfg: PortConfig) -> Dict[str, Tensor]:
    """models/renderer.py:35-85.  ``noise`` is the already-scaled additive sigma noise
    (``randn * raw_noise_std``) or None."""
    dists = z_vals[..., 1:] - z_vals[..., :-1]
    dists = torch.cat([dists, 1e10 * torch.ones_like(dists[..., :1])], -1)
    dists = dists * torch.linalg.norm(rays_d[..., None, :], ord=2, dim=-1)
    rgb = torch.sigmoid(raw[..., :3])
    sigma = raw[..., 3] + (noise if noise is not None else 0.0)
    alpha = 1.0 - torch.exp(-torch.relu(sigma) * dists)
    Ts = torch.cat([torch.ones_like(alpha[..., :1]), 1.0 - alpha + 1e-10], -1)
    Ts = torch.cumprod(Ts, -1)[..., :-1]
    weights = alpha * Ts
    rgb_map = torch.sum(weights[..., None] * rgb, -2)
    sem_map = None
    if cfg.use_semantics:
        sem_map = torch.sum(weights[..., None] * raw[..., 4:], -2)
    depth_map = torch.sum(weights * z_vals, -1, keepdim=True)
    acc_map = torch.sum(weights, -1, keepdim=True)
    depth_map[acc_map <= 1e-10] = 1e10
    disp_map = 1.0 / torch.max(torch.full_like(depth_map, 1e-10), depth_map / acc_map)
    if cfg.white_bkgd:
        rgb_map = rgb_map + (1.0 - acc_map)
        if sem_map is not None:
            sem_map = sem_map + (1.0 - acc_map)
    ret = dict(rgb=rgb_map, disp=disp_map, acc=acc_map, weights=weights, depth=depth_map)
    if sem_map is not None:
        ret["semantics"] = sem_map
    return ret


def pdf_to_cdf(weights_inner: Tensor) -> Tensor:
    """models/sampler.py:93-97."""
    w = weights_inner + 1e-5
    pdf = w / torch.sum(w, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    return torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)


def invert_cdf(bins: Tensor, cdf: Tensor, u: Tensor) -> Tuple[Tensor, Tensor]:
    """models/sampler.py:116-132.  Returns (samples, inds[int64])."""
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.max(torch.zeros_like(inds - 1), inds - 1)
    above = torch.min((cdf.shape[-1] - 1) * torch.ones_like(inds), inds)
    inds_g = torch.stack([below, above], -1)
    shape = [inds_g.shape[0], inds_g.shape[1], cdf.shape[-1]]
    cdf_g = torch.gather(cdf.unsqueeze(1).expand(shape), 2, inds_g)
    bins_g = torch.gather(bins.unsqueeze(1).expand(shape), 2, inds_g)
    denom = cdf_g[..., 1] - cdf_g[..., 0]
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - cdf_g[..., 0]) / denom
    return bins_g[..., 0] + t * (bins_g[..., 1] - bins_g[..., 0]), inds


def sample_pdf(bins: Tensor, weights_inner: Tensor, n_importance: int, u: Optional[Tensor]) -> Tensor:
    """models/sampler.py:91-134.  ``u`` None <=> det (linspace)."""
    cdf = pdf_to_cdf(weights_inner)
    if u is None:
        u = torch.linspace(0.0, 1.0, steps=n_importance).expand(list(cdf.shape[:-1]) + [n_importance])
    return invert_cdf(bins, cdf, u)[0]


def importance_z(z_vals: Tensor, weights: Tensor, n_importance: int, u: Optional[Tensor]) -> Tuple[Tensor, Tensor]:
    """models/sampler.py:155-164.  Returns (z_fine sorted [R, S+N], z_samples [R, N])."""
    mids = 0.5 * (z_vals[..., 1:] + z_vals[..., :-1])
    z_samples = sample_pdf(mids, weights[..., 1:-1], n_importance, u).detach()
    z_fine, _ = torch.sort(torch.cat([z_vals, z_samples], -1), -1)
    return z_fine, z_samples


@dataclass
class Draws:
    """The four random tensors of one ray chunk in train mode, in the reference's draw order
    (SURVEY Appendix A.6): rand[R,S], randn[R,S], rand[R,N], randn[R,S+N].  The sigma-noise
    entries are the raw ``randn`` values (not yet multiplied by raw_noise_std)."""

    t_rand: Optional[Tensor] = None
    noise0: Optional[Tensor] = None
    u: Optional[Tensor] = None
    noise1: Optional[Tensor] = None


def render_rays(sd, cfg: PortConfig, rays_o, rays_d, near, far, viewdirs, raw_noise_std=0.0,
                draws: Optional[Draws] = None, retraw=True, retpts=False) -> Dict[str, Tensor]:
    """models/nerf_net.py:71-130 for one ray chunk, with the random tensors injected."""
    draws = draws or Draws()
    z = stratified_z(near, far, cfg.n_samples, draws.t_rand)
    pts = ray_points(rays_o, rays_d, z)
    raw = point_query(sd, "nerf", pts, viewdirs[..., None, :].expand(pts.shape) if cfg.use_viewdirs else None, cfg)
    n0 = draws.noise0 * raw_noise_std if (raw_noise_std > 0 and draws.noise0 is not None) else None
    ret = composite(raw, z, rays_d, n0, cfg)
    if retraw:
        ret["raw"] = raw
    if retpts:
        ret["pts"] = pts
    if cfg.n_importance > 0:
        ret0 = ret
        z_fine, z_samples = importance_z(z, ret0["weights"], cfg.n_importance, draws.u)
        pts = ray_points(rays_o, rays_d, z_fine)
        raw = point_query(sd, "nerf_fine", pts, viewdirs[..., None, :].expand(pts.shape) if cfg.use_viewdirs else None, cfg)
        n1 = draws.noise1 * raw_noise_std if (raw_noise_std > 0 and draws.noise1 is not None) else None
        ret = composite(raw, z_fine, rays_d, n1, cfg)
        if retraw:
            ret["raw"] = raw
        if retpts:
            ret["pts"] = pts
        ret["z_std"] = torch.std(z_samples, dim=-1, unbiased=False)
        for k in ret0:
            ret[k + "0"] = ret0[k]
    return ret


def render(sd, cfg: PortConfig, ray_batch, bound_batch, raw_noise_std=0.0,
           draws_per_chunk=None, retraw=True) -> Dict[str, Tensor]:
    """models/nerf_net.py:132-195 (flatten, viewdirs, scalar bounds, ray-chunk loop, unflatten)."""
    rays_o, rays_d = ray_batch
    assert rays_o.shape == rays_d.shape
    old_shape = rays_d.shape
    rays_o = rays_o.reshape(-1, 3).float()
    rays_d = rays_d.reshape(-1, 3).float()
    viewdirs = (rays_d / torch.norm(rays_d, dim=-1, keepdim=True)).reshape(-1, 3).float()
    near, far = bound_batch
    if isinstance(near, (int, float)):
        near = near * torch.ones_like(rays_d[..., :1])
    if isinstance(far, (int, float)):
        far = far * torch.ones_like(rays_d[..., :1])
    all_ret: Dict[str, list] = {}
    for ci, i in enumerate(range(0, rays_o.shape[0], cfg.ray_chunk)):
        e = min(i + cfg.ray_chunk, rays_o.shape[0])
        d = draws_per_chunk[ci] if draws_per_chunk else None
        r = render_rays(sd, cfg, rays_o[i:e], rays_d[i:e], near[i:e], far[i:e], viewdirs[i:e],
                        raw_noise_std=raw_noise_std, draws=d, retraw=retraw)
        for k, v in r.items():
            all_ret.setdefault(k, []).append(v)
    out = {k: torch.cat(v, 0) for k, v in all_ret.items()}
    return {k: v.reshape(list(old_shape[:-1]) + list(v.shape[1:])) for k, v in out.items()}


# --------------------------------------------------------------------------- synthetic inputs
def synthetic_rays(n_rays: int, seed: int = 0, H: int = 756, W: int = 1008, focal: float = 850.0):
    """pinhole camera, identity pose, unnormalised directions exactly as
    utils/ray.py:16 (d = [(i-W/2)/f, -(j-H/2)/f, -1]), origin 0; a seeded random pixel subset."""
    g = torch.Generator().manual_seed(seed)
    pix = torch.randperm(H * W, generator=g)[:n_rays]
    j = (pix // W).float()
    i = (pix % W).float()
    d = torch.stack([(i - W * 0.5) / focal, -(j - H * 0.5) / focal, -torch.ones_like(i)], -1)
    o = torch.zeros_like(d)
    return torch.stack([o, d], 0)


NEAR, FAR = 1.2, 14.72  # models/sampler.py:45 comment; data/gen_dataset.py:95-96
