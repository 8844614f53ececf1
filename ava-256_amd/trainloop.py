"""Training-loop counterpart for the raymarch hot path.

The reference's loop is `ddp-train.py:362-445`: forward through the autoencoder, L1 image loss (+ geometry / KL terms
that belong to the conv encoders), `primvolsum` regulariser, `loss.backward()`, per-parameter NaN/Inf -> 0,
`clip_grad_norm_(1.0)`, Adam step, StepLR; one process per GPU under `DistributedDataParallel`, only parameter
gradients cross GPUs.  None of the reference's Python travels to the GPU box and its conv encoders/decoders are out of
scope for this build, so this module keeps the LOOP faithful and puts a small stand-in where
the conv decoder would be:

  SlabDecoderStandIn   emits the same `decout` dict a DecoderAssembler emits (models/decoders/assembler.py:263-269:
                       verts, template [B,K,8,8,8,4] = cat(relu(rgb*25+100), relu(alpha)), primpos, primrot, primscale)
                       from per-primitive parameters and a per-frame code; ~2k parameters per primitive, so K=16384
                       gives a 134 MB gradient all-reduce -- the same order as ava-256's 187.5 MB (SURVEY.md section
                       2.3).  Its GEOMETRY BRANCH has the control flow of assembler.py:100-109,143-253: a predicted
                       guide mesh (returned as `verts`, trained by `vertl1`), replaced by the ground-truth mesh while
                       `gt_geo` is given; primitives placed on the guide mesh; the `adaptwarps` running average of
                       2 / (neighbour distance on the mesh) as primitive scale (the one thing that does not shard by
                       camera, SURVEY.md 8e: here an explicit K-float MAX all-reduce keeps every rank's buffer equal to
                       the single-process one); `residuals_weight` on the pose residuals.
  CodeEncoderStandIn   the VAE bottleneck (models/bottlenecks/vae.py:22-58): mu = W x * 0.1, logstd = W' x * 0.01,
                       z = mu + exp(logstd) * noise in training; gives `kldiv` its (expr_mu, expr_logstd) pair.
  ColorCalStandIn      per-camera + per-identity colour affine with the reference's parametrisation
                       (models/colorcals/colorcal.py:11-31).
  BackgroundMLPStandIn per-pixel background network with the reference's shape (models/bg/mlp2d.py:19-72: two small
                       one-hot MLPs -> 40 + 40 channels, 40 positional-encoding channels, 1x1 convs 120 -> 256 x 5 -> 3,
                       output * 25 + 100), run under torch.autocast(bfloat16) on the GPU: the one genuinely dense
                       contraction next to the raymarch (154 GFLOP per 512^2 image), i.e. where MFMA belongs.
  RaymarchTrainModel   decoder -> rays + Raymarcher (one call, or the reference's two) -> colour calibration -> background -> matting
                       `rayrgb + (1 - rayalpha) * bg`, the tail of Autoencoder.decode (models/autoencoder.py:240-269).
  Trainer              the loop body with the reference's semantics and hyper-parameters (configs/config.yaml:9-21).

What is measured with it (bench.py --mode train) is therefore "iterations/s of the raymarch training path with a
stand-in decoder", not ava-256's full 46.9 M-parameter model; DESIGN.md says so wherever the number appears.
"""
import math
from typing import Callable, Dict, Optional

import torch
from torch import nn

from .raydirs import compute_raydirs
from .raymarcher import Raymarcher
from .scene import make_primitives, rodrigues


def mean_ell_1(pred: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
    """losses.py:12-14 of the reference."""
    return (pred - gt).abs().mean()


def kl_loss_stable(mu: torch.Tensor, logstd: torch.Tensor) -> torch.Tensor:
    """models/bottlenecks/vae.py:17-19 of the reference."""
    return torch.mean(-0.5 + torch.abs(logstd) + 0.5 * mu ** 2 + 0.5 * torch.exp(-2.0 * torch.abs(logstd)), dim=-1)


# configs/config.yaml:17-21 of the reference
REFERENCE_LOSS_WEIGHTS = {"irgbl1": 1.0, "vertl1": 0.1, "kldiv": 1.0e-3, "primvolsum": 0.01}


def _nearest_two(points: torch.Tensor) -> torch.Tensor:
    """[K,2] indices of every point's two nearest other points (brute force in row blocks; construction time only)."""
    K = points.shape[0]
    out = torch.empty((K, 2), dtype=torch.int64)
    for i in range(0, K, 2048):
        D = torch.cdist(points[i:i + 2048].double(), points.double())
        D[torch.arange(D.shape[0]), torch.arange(i, i + D.shape[0])] = float("inf")
        idx = D.topk(min(2, K - 1), largest=False).indices
        out[i:i + 2048] = idx if idx.shape[1] == 2 else idx.expand(-1, 2)
    return out


_SKEW = torch.tensor([[[0., 0., 0.], [0., 0., -1.], [0., 1., 0.]],
                      [[0., 0., 1.], [0., 0., 0.], [-1., 0., 0.]],
                      [[0., -1., 0.], [1., 0., 0.], [0., 0., 0.]]])   # [c][i][j]: [v]x = sum_c v_c * _SKEW[c]


def rodrigues_matrix(rvec: torch.Tensor) -> torch.Tensor:
    """scene.rodrigues (the reference's Rodrigues module, models/utils: theta = sqrt(1e-5 + |v|^2), a = v / theta,
    R = cos I + (1 - cos) a a^T + sin [a]x) written with whole-matrix operations: ~15 kernels forward and ~35 backward on
    [K,3] / [K,3,3] tensors instead of the ~200 that the nine per-element expressions over x, y, z slices cost per
    iteration (83 multiplies, 42 in-place adds, 13 negations ... on [16384] vectors: 0.85 ms of a 9 ms C3 iteration,
    gpurun_out/r04k/ops_C3.txt).  Same values to rounding."""
    theta = torch.sqrt(1e-5 + (rvec * rvec).sum(-1, keepdim=True))            # [...,1]
    a = rvec / theta
    c, sn = torch.cos(theta)[..., None], torch.sin(theta)[..., None]          # [...,1,1]
    eye = torch.eye(3, dtype=rvec.dtype, device=rvec.device)
    skew = (a[..., :, None, None] * _SKEW.to(device=rvec.device, dtype=rvec.dtype)).sum(-3)   # sum_c a_c * _SKEW[c]
    return c * eye + (1.0 - c) * (a[..., :, None] * a[..., None, :]) + sn * skew


def _matmul3(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """a @ b for stacks of 3x3 matrices, written as a broadcast product and a sum (no batched-GEMM launch)."""
    return (a[..., :, :, None] * b[..., None, :, :]).sum(-2)


def assemble_template_eager(tex, opacity, nboxes, boxsize):
    """The reference's statements for the decoder -> raymarch hand-off (what assemble.assemble_template fuses):
    models/decoders/rgb.py:137-143 and geometry.py:183-185 (view / permute / reshape of the conv outputs into per-primitive
    slabs) and assembler.py:261 (`cat([relu(rgb * 25 + 100), relu(alpha)], -1)`).  CPU path of the stand-in decoder."""
    N, B = tex.shape[0], boxsize
    nh = math.isqrt(nboxes)
    rgb = tex.view(N, B, 3, nh, B, nh, B).permute(0, 3, 5, 1, 4, 6, 2).reshape(N, nboxes, B, B, B, 3)
    alpha = opacity.view(N, B, 1, nh, B, nh, B).permute(0, 3, 5, 1, 4, 6, 2).reshape(N, nboxes, B, B, B, 1)
    return torch.cat([torch.relu(rgb * 25.0 + 100.0), torch.relu(alpha)], dim=-1)


class _FrameGain(torch.autograd.Function):
    """template[b] = base * gain[b] (base: the shared slabs [K,s,s,s,4], gain [B]).  One broadcast kernel forward; backward
    as two matrix-vector products over the incoming gradient seen as [B, M] -- g^T gain for the slabs, g base for the
    gains -- each reading it once at HBM speed (rocBLAS gemv; plain autograd would make two full-size temporaries and two
    reductions, and the [1,B] x [B,M] matrix product that hipBLASLt offers for the first is 13 x slower than the gemv:
    1.6 ms against 0.12 ms at C3, gpurun_out/r04k)."""

    @staticmethod
    def forward(ctx, base, gain):
        ctx.save_for_backward(base, gain)
        return gain.view(-1, *([1] * base.dim())) * base[None]

    @staticmethod
    def backward(ctx, g):
        base, gain = ctx.saved_tensors
        g2 = g.reshape(g.shape[0], -1)
        return torch.mv(g2.t(), gain).view_as(base), torch.mv(g2, base.reshape(-1))


class _WideLinear(torch.autograd.Function):
    """y = x W^T + b for a few rows x [B, C] and a very wide W [M, C] (the geometry head: B = 4, C = 16, M = 49152).  The
    input gradient g W reduces over M with a 4 x 16 output; hipBLASLt's pick for that shape takes 112 us at C3
    (profiles/r04z_train_C3_kernel_stats.csv, MT16x16x512; a broadcast product + reduction over a [B, M, C] temporary: 80 us)
    against ~15 for the same product split over M by hand (a batched GEMM with K = 1024 and a 48-term sum)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        return torch.addmm(bias, x, weight.t())

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = g.contiguous()   # (an expanded / transposed upstream gradient is a legal autograd output)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            M = weight.shape[0]
            if M % 1024 == 0 and M > 1024:   # split-K by hand: [S, B, 1024] x [S, 1024, C] -> sum over S
                S = M // 1024
                gx = torch.bmm(g.view(g.shape[0], S, 1024).transpose(0, 1), weight.reshape(S, 1024, -1)).sum(0)
            else:
                gx = g @ weight
        if ctx.needs_input_grad[1]:
            gw = g.t() @ x
        if ctx.needs_input_grad[2]:
            gb = g.sum(0)
        return gx, gw, gb


class CodeEncoderStandIn(nn.Module):
    """VAE bottleneck with the reference's squashing constants and sampling rule (models/bottlenecks/vae.py:22-58)."""

    def __init__(self, in_dim: int = 16, out_dim: int = 16, mean_squash: float = 0.1, std_squash: float = 0.01,
                 seed: int = 0):
        super().__init__()
        self.mu, self.logstd = nn.Linear(in_dim, out_dim), nn.Linear(in_dim, out_dim)
        self.mean_squash, self.std_squash = mean_squash, std_squash
        g = torch.Generator().manual_seed(seed + 31)  # seeded: identical on every rank
        with torch.no_grad():
            for m in (self.mu, self.logstd):
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * math.sqrt(1.0 / in_dim))
                m.bias.zero_()

    def forward(self, x, noise=None):
        mu = self.mu(x) * self.mean_squash
        logstd = self.logstd(x) * self.std_squash
        if self.training:
            z = mu + torch.exp(logstd) * (torch.randn_like(logstd) if noise is None else noise)
        else:
            z = mu
        return z, mu, logstd


class SlabDecoderStandIn(nn.Module):
    # Version of the stand-in's parametrisation.  2 (round 4 on): the per-frame gain multiplies ALL FOUR slab channels -- rgb AND
    # opacity -- inside the frame-broadcast hand-off kernel (csrc/assemble.hip), so it modulates density and reaches the
    # primvolsum / alpha gradients; version 1 scaled rgb only and expanded opacity unchanged, with parameters `rgb` / `alpha` /
    # a K-output gain (now `tex` / `opacity` / a one-output `gain`).  Checkpoints and `train.*` bench numbers of the two
    # versions are not comparable; the version travels in the state_dict as the buffer `stand_in_version`.
    STAND_IN_VERSION = 2

    def __init__(self, K: int, slab: int = 8, code_dim: int = 16, seed: int = 0, geometry: bool = True,
                 alpha_init: float = 0.5, volradius: float = 256.0, vertstd: float = 10.0):
        super().__init__()
        base = make_primitives(1, K, device="cpu", seed=seed, slab=slab)
        self.K, self.slab, self.geometry, self.volradius = K, slab, geometry, float(volradius)
        self.register_buffer("stand_in_version", torch.tensor(self.STAND_IN_VERSION, dtype=torch.int32))
        self.register_buffer("base_pos", base["primpos"][0].clone())
        self.register_buffer("base_rot", base["primrot"][0].clone())
        self.register_buffer("base_scale", base["primscale"][0].clone())
        g = torch.Generator().manual_seed(seed + 17)
        nh = math.isqrt(K)
        # K a square (every shipped configuration: 256, 4096, 16384): the head holds what the reference's conv decoders
        # emit -- a texture `tex [1, 3*slab, nh*slab, nh*slab]` and `opacity [1, slab, ...]` (models/decoders/rgb.py:137-143,
        # geometry.py:183-185) -- and the slabs are made by the fused hand-off kernel (assemble.assemble_template, SURVEY 8f
        # row N2; on CPU by the reference's own view / permute / relu / cat statements); a per-frame gain stands for the
        # view / expression conditioning.  Other K: per-primitive slab parameters and a per-primitive gain, eager.
        self.conv_layout = nh * nh == K
        if self.conv_layout:
            S = nh * slab
            self.tex = nn.Parameter(torch.randn(1, 3 * slab, S, S, generator=g))          # pre-activation, random-init statistics
            self.opacity = nn.Parameter(alpha_init * (1.0 + 0.2 * torch.randn(1, slab, S, S, generator=g)))
            self.gain = nn.Linear(code_dim, 1)
        else:
            self.rgb = nn.Parameter(torch.randn(K, slab, slab, slab, 3, generator=g))
            self.alpha = nn.Parameter(alpha_init * (1.0 + 0.2 * torch.randn(K, slab, slab, slab, 1, generator=g)))
            self.gain = nn.Linear(code_dim, K)  # per-frame, per-primitive brightness (view / expression conditioning)
        self.pos_delta = nn.Parameter(torch.zeros(K, 3))
        self.rotvec = nn.Parameter(torch.zeros(K, 3))
        self.logscale = nn.Parameter(torch.zeros(K, 3))
        with torch.no_grad():  # seeded like everything else: identical on every rank and in every process
            self.gain.weight.copy_(0.01 * torch.randn(self.gain.weight.shape, generator=g))
            self.gain.bias.zero_()
        if geometry:
            # Guide mesh with one vertex per primitive (the shell points, in the millimetre units of the dataset);
            # every primitive sits on the triangle (itself, its two nearest neighbours) -- the stand-in's idxim / barim
            # (assembler.py:118-122) -- and measures its size against those neighbours (assembler.py:147-158,183-194).
            self.register_buffer("vertmean", self.base_pos * self.volradius)
            self.register_buffer("vertstd", torch.tensor(float(vertstd)))
            nn2 = _nearest_two(self.base_pos)
            self.register_buffer("tri_idx", torch.cat([torch.arange(K)[:, None], nn2], dim=1))     # [K,3]
            self.register_buffer("tri_bar", torch.tensor([0.9, 0.05, 0.05]).expand(K, 3).clone())
            self.register_buffer("adaptwarps", torch.zeros(K))                                     # assembler.py:66
            # The reference's placement maps (assembler.py:62-63 idxim / barim [1024,1024,3]) for the primitive counts whose
            # centre grids it defines (256, 16384): every texel of a primitive's block carries that primitive's triangle.
            # On the GPU the placement then runs as ONE kernel (placement.prim_placement) instead of the
            # gather / multiply / sum chain below and its index_add backward.
            from .placement import GRIDS
            if K in GRIDS:
                ny, nx, y0, sy, x0, sx = GRIDS[K]
                T = 1024
                yy, xx = torch.meshgrid(torch.arange(T), torch.arange(T), indexing="ij")
                kmap = (torch.clamp(yy // sy, max=ny - 1) * nx + torch.clamp(xx // sx, max=nx - 1)).reshape(-1)
                self.register_buffer("idxim", self.tri_idx[kmap].reshape(T, T, 3).to(torch.int32).contiguous(), persistent=False)
                self.register_buffer("barim", self.tri_bar[kmap].reshape(T, T, 3).contiguous(), persistent=False)
            self.geo_head = nn.Linear(code_dim, K * 3)
            with torch.no_grad():
                self.geo_head.weight.copy_(0.02 * torch.randn(K * 3, code_dim, generator=g))
                self.geo_head.bias.zero_()

    def _update_adaptwarps(self, pm: torch.Tensor, store: bool) -> torch.Tensor:
        """assembler.py:183-199: size of a primitive = the larger distance to its two mesh neighbours, maximum over the
        batch; adaptwarps <- 2 / size the first time, 0.9 * old + 0.1 * new afterwards.  The reference computes this per
        rank and lets DDP's buffer broadcast overwrite it with rank 0's; here the batch maximum is taken over ALL ranks
        (one K-float MAX all-reduce), so every rank holds what a single process would on the whole batch, with
        broadcast_buffers off.  No host synchronisation (the reference's `.max().item() == 0`)."""
        n1, n2 = self.tri_idx[:, 1], self.tri_idx[:, 2]
        cs = torch.maximum((pm[:, n1] - pm).norm(dim=-1), (pm[:, n2] - pm).norm(dim=-1)).amax(dim=0)  # [K]
        if store:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                dist.all_reduce(cs, op=dist.ReduceOp.MAX)
        warps_vec = 2.0 / cs
        fresh = (self.adaptwarps.max() == 0)
        new = torch.where(fresh, warps_vec, self.adaptwarps * 0.9 + 0.1 * warps_vec)
        if store:
            self.adaptwarps.copy_(new)
            return self.adaptwarps
        # not the running-average phase: the stored buffer, or -- never initialised -- this batch's own value
        return torch.where(fresh, warps_vec, self.adaptwarps)

    def forward(self, code: torch.Tensor, schedule: Optional[Dict[str, object]] = None,
                gt_geo: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """`schedule` = forward_schedule(iternum) (ddp-train.py:371-377); `gt_geo` = the batch's normalised vertices,
        used as the guide mesh while schedule["use_gt_geo"] (assembler.py:105-109)."""
        B = code.shape[0]
        sch = schedule or {}
        rw = min(max(float(sch.get("residuals_weight", 1.0)), 0.0), 1.0)     # assembler.py:241
        gain = 1.0 + 0.1 * torch.tanh(self.gain(code))                      # [B,1] or [B,K]
        if self.conv_layout:
            if self.tex.is_cuda and self.tex.dtype == torch.float32:
                # one pass each way: slabs made and scaled per frame in the hand-off kernel; its backward reads the
                # incoming gradient once (it was: assemble + broadcast multiply, two gemv + the assemble backward)
                from .assemble import assemble_template_frames
                template = assemble_template_frames(self.tex, self.opacity, gain[:, 0].contiguous(), self.K, self.slab)
            else:
                base = assemble_template_eager(self.tex, self.opacity, self.K, self.slab)
                template = _FrameGain.apply(base[0], gain[:, 0])                            # [B,K,s,s,s,4]
        else:
            rgb = torch.relu(self.rgb * 25.0 + 100.0)                        # assembler.py:261
            alpha = torch.relu(self.alpha)
            template = torch.cat([rgb[None] * gain[:, :, None, None, None, None],
                                  alpha[None].expand(B, -1, -1, -1, -1, -1)], dim=-1).contiguous()
        pos_res, rot_res, scale_res = 0.01 * self.pos_delta, 0.1 * self.rotvec, torch.exp(0.1 * self.logscale)
        # On the GPU the residual composition (assembler.py:241-253) is one kernel each way (placement.prim_residuals, SURVEY 8f
        # row N2) instead of ~25 small kernels forward and ~60 backward; on CPU the reference's statements.
        fused_pose = self.pos_delta.is_cuda and self.pos_delta.dtype == torch.float32
        if rw < 1.0 and not fused_pose:                                       # assembler.py:242-245
            pos_res, rot_res, scale_res = pos_res * rw, rot_res * rw, scale_res * rw + (1.0 - rw)
        out = {}
        if self.geometry:
            geo = _WideLinear.apply(code, self.geo_head.weight, self.geo_head.bias).view(B, self.K, 3) * self.vertstd \
                + self.vertmean                                                              # assembler.py:100-103
            out["verts"] = geo
            guide = geo
            if gt_geo is not None and sch.get("use_gt_geo", False):
                guide = gt_geo * self.vertstd + self.vertmean                                # assembler.py:105-109
            if guide.is_cuda and guide.dtype == torch.float32 and hasattr(self, "idxim"):
                from .placement import prim_placement
                pm, _, _ = prim_placement(guide.contiguous(), self.idxim, self.barim, self.volradius, self.K)
            else:                                                                            # assembler.py:118-122,143
                pm = (self.tri_bar[None, :, :, None] * guide[:, self.tri_idx]).sum(dim=2) / self.volradius
            with torch.no_grad():
                aw = self._update_adaptwarps(pm, bool(sch.get("running_avg_scale", False)))
            pos0, scale0 = pm, (aw * 0.8)[:, None]
        else:
            pos0, scale0 = self.base_pos, self.base_scale
        if fused_pose:
            from .placement import prim_residuals
            primpos, primrot, primscale = prim_residuals(pos0, self.base_rot, scale0, pos_res, rot_res, scale_res, rw, B)
            out.update(template=template, primpos=primpos, primrot=primrot, primscale=primscale)
            return out
        if self.geometry:
            primpos = (pm + pos_res[None]).contiguous()
            primscale = ((aw * 0.8)[None, :, None] * scale_res[None]).expand(B, -1, -1).contiguous()
        else:
            primpos = (self.base_pos + pos_res)[None].expand(B, -1, -1).contiguous()
            primscale = (self.base_scale * scale_res)[None].expand(B, -1, -1).contiguous()
        # base_rot @ R as a broadcast product + sum: hipBLASLt's batched GEMM takes 150 us for 16384 3x3 products (and twice
        # that in the backward); two elementwise kernels take ~10 (profiles/r04z_train_C3_kernel_stats.csv)
        primrot = _matmul3(self.base_rot, rodrigues_matrix(rot_res))[None].expand(B, -1, -1, -1).contiguous()
        out.update(template=template, primpos=primpos, primrot=primrot, primscale=primscale)
        return out


class ColorCalStandIn(nn.Module):
    """w = wcam[cam] + wident[id], b = bcam[cam] + bident[id]; image * w + b per channel
    (models/colorcals/colorcal.py:11-31: ones / zeros initialisation, so it starts as the identity)."""

    def __init__(self, ncams: int, nident: int):
        super().__init__()
        self.wcam = nn.Parameter(torch.ones(ncams, 3))
        self.bcam = nn.Parameter(torch.zeros(ncams, 3))
        self.wident = nn.Parameter(torch.zeros(nident, 3))
        self.bident = nn.Parameter(torch.zeros(nident, 3))

    def forward(self, image, camindex, idindex):
        w = self.wcam[camindex] + self.wident[idindex]
        b = self.bcam[camindex] + self.bident[idindex]
        return image * w[:, :, None, None] + b[:, :, None, None]


class _PixelLinearFn(torch.autograd.Function):
    """y = x @ W^T + b for a tall pixel matrix x [M, C_in] (M ~ 10^6, C <= 256).  Forward and input gradient are the GEMMs
    torch would issue anyway; the WEIGHT gradient x^T @ dy has a 256 x 256 output and K = M, for which hipBLASLt picks a
    64x64 tile without split-K (16 workgroups on 256 CUs: 1.9 ms = 72 TFLOP/s at M = 2^20, `profiles/r02h_*`).  Here
    it is a batched GEMM over `chunks` row blocks (1024 workgroups) whose partial products are summed in fp32."""

    @staticmethod
    def forward(ctx, x, weight, bias, chunks):
        ctx.save_for_backward(x, weight)
        ctx.chunks = chunks
        return torch.addmm(bias, x, weight.t())

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gy = gy.contiguous()
        M = x.shape[0]
        S = ctx.chunks if M % ctx.chunks == 0 else 1
        gx = gy @ weight if ctx.needs_input_grad[0] else None
        gw = torch.bmm(gy.view(S, M // S, -1).transpose(1, 2), x.view(S, M // S, -1)).float().sum(0).to(weight.dtype)
        gb = gy.float().sum(0).to(gy.dtype)
        return gx, gw, gb, None


class PixelLinear(nn.Linear):
    """nn.Linear over a [..., C_in] pixel tensor; on CUDA under autocast the weight gradient is the chunked form above."""

    def forward(self, x):
        if x.is_cuda and torch.is_autocast_enabled() and x.shape[:-1].numel() >= 1 << 16:
            dt = torch.get_autocast_dtype("cuda")
            lead = x.shape[:-1]
            y = _PixelLinearFn.apply(x.reshape(-1, x.shape[-1]).to(dt), self.weight.to(dt), self.bias.to(dt), 64)
            return y.view(*lead, -1)
        return super().forward(x)


class BackgroundMLPStandIn(nn.Module):
    """Per-pixel background colour from (camera, identity, pixel position), the shape of models/bg/mlp2d.py:19-72:
    one-hot camera / identity -> Linear(., 256) -> LeakyReLU(0.2) -> Linear(256, 40) each, 20 sin + 20 cos positional
    channels of the normalised pixel coordinates, then 1x1 convolutions 120 -> 256 -> 256 -> 256 -> 256 -> 256 -> 3 with
    LeakyReLU(0.2) between them, output * 25 + 100.  Per pixel that is 120*256 + 4*256^2 + 256*3 MACs = 0.59 MFLOP:
    154 GFLOP per 512 x 512 image forward, the largest dense contraction of the training step.
    `autocast_dtype` (bf16 by default) is applied on CUDA only: the 1x1 convolutions run as MFMA GEMMs."""

    def __init__(self, ncams: int, nident: int, width: int = 256, autocast_dtype=torch.bfloat16, seed: int = 0,
                 fused: bool = True):
        super().__init__()
        self.ncams, self.nident, self.autocast_dtype = ncams, nident, autocast_dtype
        self.fused = fused and width == 256   # CUDA + bf16: the fused MFMA kernels instead of eager autocast GEMMs
        act = lambda: nn.LeakyReLU(0.2)
        self.cammod = nn.Sequential(nn.Linear(ncams, 256), act(), nn.Linear(256, 40))
        self.idmod = nn.Sequential(nn.Linear(nident, 256), act(), nn.Linear(256, 40))
        # the reference's 1x1 Conv2d stack, held as Linear layers over a channels-last pixel matrix [b*h*w, C]: the same
        # arithmetic, and each layer is one plain (pixels x C_in) @ (C_in x C_out) GEMM for hipBLASLt / MFMA
        layers, cin = [], 120
        for _ in range(5):
            layers += [PixelLinear(cin, width), act()]
            cin = width
        layers += [PixelLinear(cin, 3)]
        self.mlp = nn.Sequential(*layers)
        g = torch.Generator().manual_seed(seed + 23)  # seeded: identical on every rank
        with torch.no_grad():
            for m in self.modules():
                if isinstance(m, nn.Linear):
                    fan_in = m.weight[0].numel()
                    m.weight.copy_(torch.randn(m.weight.shape, generator=g) * math.sqrt(2.0 / fan_in))
                    m.bias.zero_()

    def forward(self, camindex, idindex, samplecoords):
        b, h, w = samplecoords.shape[0], samplecoords.shape[1], samplecoords.shape[2]
        dev = samplecoords.device
        if dev.type == "cuda" and self.fused and self.autocast_dtype is torch.bfloat16:
            # one MFMA kernel per direction (csrc/bgmlp.hip).  The two codes are constant over an image: their 80 input
            # channels of the first layer become a per-image bias (fp32, differentiable through plain autograd).
            from .bgmlp import fused_background_mlp
            camenc = self.cammod(torch.nn.functional.one_hot(camindex, self.ncams).float())
            idenc = self.idmod(torch.nn.functional.one_hot(idindex, self.nident).float())
            lin = [m for m in self.mlp if isinstance(m, nn.Linear)]
            bias1 = lin[0].bias + torch.cat([camenc, idenc], dim=-1) @ lin[0].weight[:, :80].t()
            return fused_background_mlp(samplecoords, bias1, lin[0].weight[:, 80:], [(m.weight, m.bias) for m in lin[1:5]],
                                        lin[5].weight, lin[5].bias)
        use_amp = dev.type == "cuda" and self.autocast_dtype is not None
        wdt = self.cammod[0].weight.dtype  # float32; float64 in the tests' replay of a training step
        with torch.autocast(device_type=dev.type, dtype=self.autocast_dtype or torch.bfloat16, enabled=use_amp):
            camenc = self.cammod(torch.nn.functional.one_hot(camindex, self.ncams).to(wdt))
            idenc = self.idmod(torch.nn.functional.one_hot(idindex, self.nident).to(wdt))
            posenc = torch.cat([torch.sin(2 ** i * math.pi * samplecoords) for i in range(10)] +
                               [torch.cos(2 ** i * math.pi * samplecoords) for i in range(10)], dim=-1)   # mlp2d.py:64-68
            x = torch.cat([camenc[:, None, None, :].expand(b, h, w, 40).to(posenc.dtype),
                           idenc[:, None, None, :].expand(b, h, w, 40).to(posenc.dtype), posenc], dim=-1)
            out = self.mlp(x)                                                       # [b,h,w,3]
        if out.dtype in (torch.bfloat16, torch.float16):
            out = out.float()
        return out.permute(0, 3, 1, 2) * 25.0 + 100.0


class RaymarchTrainModel(nn.Module):
    """decoder -> rays -> raymarch -> colour calibration -> background -> matting: the tail of Autoencoder.decode
    (models/autoencoder.py:225-269).  `colorcal` / `bgmodel` may be None (identity / black background, the reference's
    own fallbacks at :255 and :263-267)."""

    def __init__(self, decoder: nn.Module, volradius: float = 256.0, dt: float = 1.0,
                 renderer: Optional[Callable] = None, colorcal: Optional[nn.Module] = None,
                 bgmodel: Optional[nn.Module] = None, encoder: Optional[nn.Module] = None, fused_rays: bool = True,
                 fused_tail: bool = True):
        super().__init__()
        self.fused_rays = fused_rays  # rays made inside the forward march instead of the two statements
        self.fused_tail = fused_tail  # colour calibration + matting + L1 sum as one kernel each way (csrc/pixeltail.hip)
        self.decoder = decoder
        self.encoder = encoder
        self.raymarcher = Raymarcher(volradius, dt)
        self.colorcal = colorcal
        self.bgmodel = bgmodel
        self._renderer = renderer  # CPU tests inject a pure-torch stand-in; None = the gfx950 kernels

    def forward(self, camrot, campos, focal, princpt, pixelcoords, code, schedule=None, camindex=None, idindex=None,
                bg=None, gt_verts=None, noise=None, target=None):
        """`target` (optional, the batch's image): lets the fused decode tail of the GPU path sum |irgbrec - image| in the
        pass that writes irgbrec (returned as `irgbl1_sum`; the Trainer divides by the element count = mean_ell_1)."""
        self.last_schedule = schedule  # ddp-train.py:371-377; consumed by the decoder's geometry branch
        expr_mu = expr_logstd = None
        if self.encoder is not None:                                                 # autoencoder.py: VAE bottleneck
            code, expr_mu, expr_logstd = self.encoder(code, noise)
        decout = self.decoder(code, schedule=schedule, gt_geo=gt_verts)
        have_idx = camindex is not None and idindex is not None
        if (self._renderer is None and self.fused_tail and self.fused_rays and decout.get("warp") is None
                and decout["template"].is_cuda):
            # autoencoder.py:240-265 as two kernels: rays + march (rayrgba stays in the march's [N,H,W,4] layout), then
            # colour calibration + matting + L1 sum in one pass (csrc/pixeltail.hip), whose backward hands the march its
            # upstream gradient in that layout -- no NHWC <-> NCHW split, no eager per-pixel statements
            from .mvpraymarch import mvpraymarch_from_cameras
            from .pixeltail import decode_tail
            rm = self.raymarcher
            rayrgba = mvpraymarch_from_cameras(campos, camrot, focal, princpt, pixelcoords, rm.volume_radius, rm.dt,
                                               (decout["primpos"], decout["primrot"], decout["primscale"]), decout["template"])
            cw = cb = None
            if self.colorcal is not None and have_idx:                               # colorcal.py:28-30
                cw = self.colorcal.wcam[camindex] + self.colorcal.wident[idindex]
                cb = self.colorcal.bcam[camindex] + self.colorcal.bident[idindex]
            if bg is None and self.bgmodel is not None and have_idx:
                bg = self.bgmodel(camindex, idindex, self._samplecoords(pixelcoords))
            irgbrec, ialpha, l1sum = decode_tail(rayrgba, cw, cb, None if bg is None else bg.contiguous(), target)
            return {"irgbrec": irgbrec, "ialpha": ialpha, "primscale": decout["primscale"], "bg": bg,
                    "verts": decout.get("verts"), "expr_mu": expr_mu, "expr_logstd": expr_logstd,
                    "irgbl1_sum": l1sum if target is not None else None}
        if self._renderer is not None:
            rayrgb, rayalpha = self._renderer(camrot, campos, focal, princpt, pixelcoords, decout)
        elif self.fused_rays and decout.get("warp") is None:
            rayrgb, rayalpha, _, _ = self.raymarcher.forward_from_cameras(campos, camrot, focal, princpt, pixelcoords, decout)
        else:                                                                        # autoencoder.py:240-252 as written
            raypos, raydir, tminmax = compute_raydirs(campos, camrot, focal, princpt, pixelcoords,
                                                      self.raymarcher.volume_radius)
            rayrgb, rayalpha, _, _ = self.raymarcher(raypos, raydir, tminmax, decout)
        if self.colorcal is not None and have_idx:                                   # autoencoder.py:254-256
            rayrgb = self.colorcal(rayrgb, camindex, idindex)
        if bg is None and self.bgmodel is not None and have_idx:                     # autoencoder.py:258-261
            bg = self.bgmodel(camindex, idindex, self._samplecoords(pixelcoords))
        if bg is not None:                                                           # autoencoder.py:263-265
            rayrgb = rayrgb + (1.0 - rayalpha) * bg
        return {"irgbrec": rayrgb, "ialpha": rayalpha, "primscale": decout["primscale"], "bg": bg,
                "verts": decout.get("verts"), "expr_mu": expr_mu, "expr_logstd": expr_logstd}


    @staticmethod
    def _samplecoords(pixelcoords):
        return torch.cat([pixelcoords[..., :1] * 2 / (pixelcoords.shape[-2] - 1) - 1,
                          pixelcoords[..., 1:] * 2 / (pixelcoords.shape[-3] - 1) - 1], dim=-1)   # autoencoder.py:231-237


def forward_schedule(iternum: int) -> Dict[str, object]:
    """Forward-pass switches of the first iterations, ddp-train.py:371-377: for iternum < 100 the decoder is driven
    with `running_avg_scale=True`, ground-truth geometry (`gt_geo = verts`) and `residuals_weight = 0`; afterwards
    `False`, `None`, `1.0`.  The Trainer hands them to the model, whose decoder consumes all three
    (SlabDecoderStandIn.forward: running average of `adaptwarps`, ground-truth guide mesh, residual weight)."""
    warm = iternum < 100
    return {"running_avg_scale": warm, "use_gt_geo": warm, "residuals_weight": 0.0 if warm else 1.0}


class Trainer:
    """One optimisation step with the reference's semantics (ddp-train.py:404-442)."""

    def __init__(self, model: nn.Module, lr: float = 2.0e-4, lr_scheduler_iter: int = 10_000, gamma: float = 1.4,
                 clip: float = 1.0, loss_weights: Optional[Dict[str, float]] = None, ddp: bool = False,
                 device_ids=None, bucket_cap_mb: int = 256, graph: bool = False, graph_warmup: int = 3):
        """`graph=True` (single process, GPU): after `graph_warmup` eager iterations of a forward schedule the whole
        iteration -- forward, losses, backward, gradient hygiene, Adam -- is captured once into a hipGraph and replayed: one
        launch per iteration instead of ~350 (the stand-in's small tensors make the eager loop host-bound at C3).  A change of
        the forward schedule (ddp-train.py:371-377, at iteration 100) captures a second graph; the learning-rate schedule
        writes a device scalar the captured Adam reads.  In graph mode `step` returns the graph's OWN output tensors (loss, loss
        parts, `last_grad_norm`): the next replay overwrites them -- read or clone them before the next `step`; the batch must
        keep its shapes (its tensors are copied into the graph's inputs)."""
        self.raw_model = model
        if ddp:
            from torch.nn.parallel import DistributedDataParallel as DDP
            # gradients only; no per-forward buffer broadcast (the reference's default re-broadcasts ~115 MB of
            # buffers every iteration) and one flat bucket sized for point-to-point xGMI
            model = DDP(model, device_ids=device_ids, broadcast_buffers=False, gradient_as_bucket_view=True,
                        bucket_cap_mb=bucket_cap_mb)
        self.model = model
        self.params = [p for p in model.parameters() if p.requires_grad]
        on_gpu = bool(self.params) and self.params[0].is_cuda
        self.graph = bool(graph) and on_gpu and not ddp
        # (the reference's torch.optim.Adam, ddp-train.py:78; on the GPU as ONE multi-tensor kernel instead of the ~10
        #  foreach passes over every parameter -- the same update; for graph replay with the step count and the learning
        #  rate on the device)
        self._lr0, self._lr_iter, self._gamma = float(lr), int(lr_scheduler_iter), float(gamma)
        if self.graph:
            self._lr_dev = torch.tensor(float(lr), device=self.params[0].device, dtype=torch.float32)
            self.optim = torch.optim.Adam(self.params, lr=self._lr_dev, betas=(0.9, 0.999), fused=True, capturable=True)
            self.sched = None
        else:
            self.optim = torch.optim.Adam(self.params, lr=lr, betas=(0.9, 0.999), fused=on_gpu)
            self.sched = torch.optim.lr_scheduler.StepLR(self.optim, step_size=lr_scheduler_iter, gamma=gamma)
        self.ddp = ddp
        self.clip = clip
        self.loss_weights = dict(loss_weights or REFERENCE_LOSS_WEIGHTS)  # configs/config.yaml:17-21
        self.iternum = 0
        import inspect
        self._model_takes_target = "target" in inspect.signature(self.raw_model.forward).parameters
        self._clipper = None
        self.last_grad_norm = None
        self._graphs, self._eager_seen, self._graph_warmup = {}, {}, int(graph_warmup)
        self.graph_replays = 0
        self.graph_check_every, self.graph_recaptures = 64, 0   # replays between looks at the list-overflow flag; graphs dropped

    @classmethod
    def from_config(cls, model, cfg, **kw):
        """`cfg` = config.load_train_config(<the reference's YAML>): its learning rate, schedule, clip and loss weights
        (configs/config.yaml:9-21; ddp-train.py:78,82,404-430,441)."""
        args = dict(cfg["trainer"])
        args.update(kw)
        return cls(model, **args)

    def losses(self, output, batch):
        """ddp-train.py:404-418.  A term whose inputs the model does not produce (no geometry branch, no VAE pair) is
        skipped, like a key absent from the reference's `loss_weights`."""
        out = {}
        if "irgbl1" in self.loss_weights:
            if output.get("irgbl1_sum") is not None:   # summed by the fused decode tail in the pass that wrote irgbrec
                out["irgbl1"] = output["irgbl1_sum"] / output["irgbrec"].numel()
            else:
                out["irgbl1"] = mean_ell_1(output["irgbrec"], batch["image"])
        if "vertl1" in self.loss_weights and output.get("verts") is not None and "verts" in batch:
            dec = self.raw_model.decoder
            out["vertl1"] = mean_ell_1(output["verts"], batch["verts"] * dec.vertstd + dec.vertmean)
        if "primvolsum" in self.loss_weights:
            # ddp-train.py:415 `torch.sum(torch.prod(1 / primscale, dim=-1), dim=-1)` with the three-factor product written out:
            # the backward of torch.prod(dim) counts zeros with `.item()` -- a host synchronisation per iteration, and the
            # one statement of the iteration a hipGraph capture refuses.  Same values.
            inv = 1.0 / output["primscale"]
            out["primvolsum"] = torch.sum(inv[..., 0] * inv[..., 1] * inv[..., 2], dim=-1)
        if "kldiv" in self.loss_weights and output.get("expr_mu") is not None:
            out["kldiv"] = kl_loss_stable(output["expr_mu"], output["expr_logstd"])
        if not out:
            raise ValueError("No losses were computed. We can't train like that!")
        return out

    def total_loss(self, losses):
        """ddp-train.py:424-430 (no (value, weight) tuples on this path: every term is a plain mean)."""
        return sum(self.loss_weights[k] * torch.mean(v) for k, v in losses.items())

    def _iteration(self, batch, schedule):
        """Forward, losses, backward, gradient hygiene, optimiser -- everything of ddp-train.py:371-442 that runs on the device."""
        output = self.model(batch["camrot"], batch["campos"], batch["focal"], batch["princpt"], batch["pixelcoords"],
                            batch["code"], schedule=schedule, camindex=batch.get("camindex"),
                            idindex=batch.get("idindex"), gt_verts=batch.get("verts"), noise=batch.get("noise"),
                            **({"target": batch["image"]} if self._model_takes_target and "irgbl1" in self.loss_weights else {}))
        losses = self.losses(output, batch)
        loss = self.total_loss(losses)
        # single process: drop the gradients, so that backward hands each parameter its gradient tensor instead of
        # zero-filling and then accumulating into it (two passes over every parameter saved); under DDP the gradients are
        # views of the all-reduce bucket and stay where they are
        self.optim.zero_grad(set_to_none=not self.ddp)
        loss.backward()
        # NaN / Inf -> 0 in every gradient, then clip the global 2-norm (ddp-train.py:436-441: two masked assignments
        # per tensor -- ~600 host syncs per iteration on ava-256 -- and clip_grad_norm_).  On the GPU both are two
        # multi-tensor HIP passes with the coefficient computed on the device (gradclip.py); the
        # eager statements remain only for the CPU host-logic tests, where no kernel of this package can run.
        if self.params and self.params[0].is_cuda:
            if self._clipper is None:
                from .gradclip import GradClipper
                self._clipper = GradClipper(self.params[0].device)
            self.last_grad_norm = self._clipper(self.params, self.clip)
        else:
            for p in self.params:
                if p.grad is not None:
                    p.grad.nan_to_num_(nan=0.0, posinf=0.0, neginf=0.0)
            self.last_grad_norm = torch.nn.utils.clip_grad_norm_(self.params, self.clip)
        self.optim.step()
        return loss.detach(), {k: v.detach().mean() for k, v in losses.items()}

    def _advance(self):
        if self.sched is not None:
            self.sched.step()
        self.iternum += 1
        if self.sched is None:  # StepLR by hand (ddp-train.py:82): the captured Adam reads this device scalar
            lr = self._lr0 * self._gamma ** (self.iternum // self._lr_iter)
            if self.iternum % self._lr_iter == 0:
                self._lr_dev.fill_(lr)

    def step(self, batch: Dict[str, torch.Tensor]):
        schedule = forward_schedule(self.iternum)
        if not self.graph:
            out = self._iteration(batch, schedule)
            self._advance()
            return out
        # The graph is the iteration of ONE schedule over ONE batch layout: tensor shapes and the non-tensor entries (a (W, H)
        # pixelcoords tuple) are frozen into it, so they are part of the key -- another layout gets its own graph instead of a
        # silent replay of the captured one (or a broadcasting copy_).
        key = (tuple(sorted(schedule.items())),
               tuple(sorted((k, tuple(v.shape) if torch.is_tensor(v) else repr(v)) for k, v in batch.items())))
        st = self._graphs.get(key)
        if st is None:
            seen = self._eager_seen.get(key, 0)
            if seen < self._graph_warmup:      # lazy initialisation (Adam state, list capacities, index checks) happens here
                self._eager_seen[key] = seen + 1
                out = self._iteration(batch, schedule)
                self._advance()
                return out
            st = self._graphs[key] = self._capture(batch, schedule)
        for k, t in st["in"].items():
            src = batch[k]
            if torch.is_tensor(t) and src.data_ptr() != t.data_ptr():
                t.copy_(src)
        st["graph"].replay()
        self.graph_replays += 1
        st["replays"] += 1
        if st["handoff"] is not None and st["replays"] >= st["next_check"]:
            # The graph froze the march's primitive-list capacity of the moment it was captured (the operators skip their
            # demand feedback while capturing and on replay).  If primitive footprints have grown since, primitives over that
            # capacity sit on the slow ray-centric backward for the rest of the graph's life: look at the forward's overflow
            # flag now and then (one 4-byte read-back).  The flag alone is no reason to re-capture: the capacity policy itself
            # leaves it raised for good in three cases (an outlier clipped to 2 x the 99.9th percentile, the memory budget, the
            # cap of 2048 -- mvpraymarch.wanted_from_histogram / primlist_capacity), and a new graph would carry the same
            # capacity.  So ask what the feedback would choose for the counters of THIS replay; only a larger capacity drops
            # the graph (the next iterations run eagerly, their lists sized from the measurement just noted, and a new graph
            # is captured); otherwise the interval between looks doubles.
            pl_count, nk, (hn, hh, hw, hk, hcap) = st["handoff"]
            flagged = bool(int(pl_count[nk].item()) & 1)          # kFlagListOverflow (csrc/march_common.h)
            recapture = False
            if flagged:
                from .mvpraymarch import capacity_wanted_now
                recapture = capacity_wanted_now(pl_count, hn, hh, hw, hk) > hcap
            if recapture:
                del self._graphs[key]
                self._eager_seen[key] = 0
                self.graph_recaptures += 1
            else:
                if flagged:   # raised by the policy's own clipping: nothing a new capture would change -- look less often
                    st["check_every"] = min(2 * st["check_every"], 1 << 16)
                st["next_check"] = st["replays"] + st["check_every"]
        self.last_grad_norm = st["norm"]
        self._advance()
        return st["loss"], st["parts"]

    def _capture(self, batch, schedule):
        """One iteration recorded into a hipGraph (torch.cuda.CUDAGraph): inputs are copied into tensors the graph owns, every
        intermediate -- gradients included -- lives in the graph's private pool and is reused by each replay.  The operators
        of this package skip their host-side bookkeeping while a stream is being captured (list-capacity feedback, index
        checks), so the capacities of this moment are the graph's."""
        from . import _hooks
        # (non-tensor entries -- a (W, H) pixelcoords tuple -- are part of the batch the iteration reads: passed through)
        static_in = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
        torch.cuda.synchronize()
        self.optim.zero_grad(set_to_none=True)
        g = torch.cuda.CUDAGraph()
        keep = _hooks.keep_raysat
        _hooks.keep_raysat, _hooks.last_pl_count = True, None
        try:
            with torch.cuda.graph(g):
                loss, parts = self._iteration(static_in, schedule)
                norm = self.last_grad_norm
            handoff = None
            if _hooks.last_pl_count is not None:   # the captured forward's counters + flags word (a tensor of the graph's pool)
                handoff = (_hooks.last_pl_count, _hooks.last_flags_index, _hooks.last_handoff_shape)
        finally:
            _hooks.keep_raysat, _hooks.last_raysat, _hooks.last_pl_count = keep, None, None
        # the capture itself ran nothing: the first replay is this iteration
        return {"graph": g, "in": static_in, "loss": loss, "parts": parts, "norm": norm, "handoff": handoff, "replays": 0,
                "check_every": self.graph_check_every, "next_check": self.graph_check_every}


@torch.no_grad()
def make_training_batch(N, H, W, K, device, seed=1112, code_dim=16, target_decoder: Optional[nn.Module] = None,
                        ncams: int = 80, nident: int = 4, target_bg: float = 60.0):
    """Synthetic batch with the keys of the reference's data contract that this path uses:
    cameras, pixel grid, camindex / idindex, a per-frame code and target images.  Targets are renders of a
    differently-seeded stand-in decoder matted over a flat grey background (`target_bg`, in the 0..255 image units of
    the reference), so that the matting term carries gradient into rayalpha like a real captured frame does."""
    from .scene import make_cameras, pixel_grid
    cams = make_cameras(N, H, W, device=device, seed=seed)
    g = torch.Generator(device=device).manual_seed(seed + 5)
    batch = {k: cams[k] for k in ("camrot", "campos", "focal", "princpt")}
    batch["pixelcoords"] = pixel_grid(N, H, W, device=device)
    batch["code"] = torch.randn(N, code_dim, device=device, generator=g)
    batch["camindex"] = torch.arange(N, device=device) % ncams
    batch["idindex"] = (torch.arange(N, device=device) // ncams) % nident
    # normalised ground-truth vertices (one per primitive of the stand-in's guide mesh; SURVEY.md appendix D `verts`)
    # and the VAE's sampling noise, drawn here so that a replay of the step sees the same numbers
    # (one tracked FRAME seen by the batch's cameras -- "1 subject, 80 cams" -- plus a small per-image difference: the
    # adaptwarps running average takes the maximum over the batch, assembler.py:191-192, so independent meshes per image
    # would inflate every primitive with the batch size)
    batch["verts"] = (0.05 * torch.randn(1, K, 3, device=device, generator=g) +
                      0.005 * torch.randn(N, K, 3, device=device, generator=g))
    batch["noise"] = torch.randn(N, code_dim, device=device, generator=g)
    if target_decoder is not None:
        tm = RaymarchTrainModel(target_decoder.to(device))
        bgt = torch.full((N, 3, H, W), float(target_bg), device=device)
        batch["image"] = tm(batch["camrot"], batch["campos"], batch["focal"], batch["princpt"], batch["pixelcoords"],
                            batch["code"], bg=bgt, gt_verts=batch["verts"],
                            schedule={"use_gt_geo": True})["irgbrec"].clone()
    else:
        batch["image"] = torch.zeros(N, 3, H, W, device=device)
    return batch, cams["volradius"]
