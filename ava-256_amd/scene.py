"""Seeded synthetic scenes for tests and bench (no dataset, no checkpoint).

Geometry mimics a fresh ava-256 decoder output: K oriented boxes tiling a head-sized sphere shell
(radius 0.40 volume units), box half-extent 0.625 * spacing (assembler-like 'adaptwarps * 0.8' sizing),
8^3 RGBA slabs with random-init statistics, N pinhole cameras on the front hemisphere at 5.6 volume units
looking at the origin.  Everything is generated with a torch.Generator on the requested device.
"""
import math

import torch


def _normalize(v):
    return v / v.norm(dim=-1, keepdim=True)


def rodrigues(rvec):
    theta = torch.sqrt(1e-5 + (rvec ** 2).sum(-1))
    a = rvec / theta[..., None]
    c, s = torch.cos(theta), torch.sin(theta)
    x, y, z = a[..., 0], a[..., 1], a[..., 2]
    R = torch.stack([x * x + (1 - x * x) * c, x * y * (1 - c) - z * s, x * z * (1 - c) + y * s,
                     x * y * (1 - c) + z * s, y * y + (1 - y * y) * c, y * z * (1 - c) - x * s,
                     x * z * (1 - c) - y * s, y * z * (1 - c) + x * s, z * z + (1 - z * z) * c], dim=-1)
    return R.reshape(rvec.shape[:-1] + (3, 3))


def uvgrid_order(nrm):
    """Permutation that lists shell points like a row-major UV map: ~sqrt(K) latitude rows, each sorted by
    azimuth.  ava-256's decoder emits primitives in row-major order of a 2-D UV grid (128 x 128 for K = 16384),
    so consecutive primitives are neighbours on the surface; the reference's fixed-order heap BVH relies on that.
    A raw Fibonacci spiral has the opposite property (consecutive points are a golden angle apart)."""
    K = nrm.shape[0]
    rows = max(1, int(round(math.sqrt(K))))
    zrank = torch.argsort(torch.argsort(nrm[:, 2], descending=True))          # 0 = top
    row = torch.clamp((zrank.to(torch.float64) * rows / K).floor().to(torch.int64), max=rows - 1)
    az = torch.atan2(nrm[:, 1], nrm[:, 0]).to(torch.float64)
    key = row.to(torch.float64) * 10.0 + (az + math.pi)                        # az + pi in [0, 2 pi] < 10
    return torch.argsort(key)


def make_primitives(N, K, device="cpu", seed=1112, radius=0.40, slab=8, alpha_gain=1.0, dtype=torch.float32,
                    order="uvgrid"):
    g = torch.Generator(device=device).manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g, device=device, dtype=dtype)
    # Fibonacci sphere
    i = torch.arange(K, device=device, dtype=dtype) + 0.5
    z = 1 - 2 * i / K
    rxy = torch.sqrt(torch.clamp(1 - z * z, min=0))
    phi = i * (math.pi * (3.0 - math.sqrt(5.0)))
    nrm = torch.stack([rxy * torch.cos(phi), rxy * torch.sin(phi), z], dim=-1)  # [K,3] outward normal
    if order == "uvgrid":
        nrm = nrm[uvgrid_order(nrm)].contiguous()
    else:
        assert order == "fibonacci"  # adversarial for the fixed-order BVH: no locality between consecutive k
    primpos = (radius * nrm)[None].expand(N, K, 3) + 0.002 * rn(N, K, 3)
    # tangent frame: rows (t, b, n) then transposed (columns are the box axes), times a small random rotation
    up = torch.tensor([0.0, 0.0, 1.0], device=device, dtype=dtype).expand(K, 3)
    alt = torch.tensor([1.0, 0.0, 0.0], device=device, dtype=dtype).expand(K, 3)
    ref = torch.where((nrm[:, 2:3].abs() > 0.9), alt, up)
    t = _normalize(torch.cross(ref, nrm, dim=-1))
    b = torch.cross(nrm, t, dim=-1)
    frame = torch.stack([t, b, nrm], dim=1).transpose(1, 2)  # [K,3,3]
    primrot = torch.matmul(frame[None].expand(N, K, 3, 3), rodrigues(0.01 * rn(N, K, 3)))
    spacing = math.sqrt(4 * math.pi * radius * radius / K)
    primscale = (1.0 / (0.625 * spacing)) * torch.exp(0.01 * rn(N, K, 3))
    template = torch.empty(N, K, slab, slab, slab, 4, device=device, dtype=dtype)
    template[..., :3] = torch.relu(100 + 25 * rn(N, K, slab, slab, slab, 3))
    template[..., 3] = alpha_gain * torch.exp(0.1 * rn(N, K, slab, slab, slab))
    return dict(primpos=primpos.contiguous(), primrot=primrot.contiguous(), primscale=primscale.contiguous(),
                template=template)


def make_cameras(N, H, W, device="cpu", seed=1112, dist=5.6, volradius=256.0, focal_mult=5.0, dtype=torch.float32):
    g = torch.Generator(device=device).manual_seed(seed + 1)
    u = torch.rand(N, generator=g, device=device, dtype=dtype)
    v = torch.rand(N, generator=g, device=device, dtype=dtype)
    az = (u - 0.5) * math.pi * 0.9          # front hemisphere
    el = (v - 0.5) * math.pi * 0.5
    c = torch.stack([torch.sin(az) * torch.cos(el), torch.sin(el), -torch.cos(az) * torch.cos(el)], dim=-1)
    campos = c * dist * volradius            # millimetre-like units; raypos = campos / volradius
    zc = _normalize(-c)
    upv = torch.tensor([0.0, 1.0, 0.0], device=device, dtype=dtype).expand(N, 3)
    xc = _normalize(torch.cross(upv, zc, dim=-1))
    yc = torch.cross(zc, xc, dim=-1)
    camrot = torch.stack([xc, yc, zc], dim=1).contiguous()  # rows = camera axes in world coordinates
    focal = torch.full((N, 2), focal_mult * W, device=device, dtype=dtype)
    princpt = torch.tensor([W * 0.5, H * 0.5], device=device, dtype=dtype).expand(N, 2).contiguous()
    return dict(campos=campos.contiguous(), camrot=camrot, focal=focal, princpt=princpt, volradius=volradius,
                stepsize=1.0 / volradius)


def pixel_grid(N, H, W, device="cpu", dtype=torch.float32):
    py, px = torch.meshgrid(torch.arange(H, device=device, dtype=dtype), torch.arange(W, device=device, dtype=dtype),
                            indexing="ij")
    return torch.stack([px, py], dim=-1)[None].expand(N, H, W, 2).contiguous()


def make_scene(N, H, W, K, device="cpu", seed=1112, alpha_gain=1.0, slab=8, order="uvgrid"):
    s = make_primitives(N, K, device=device, seed=seed, alpha_gain=alpha_gain, slab=slab, order=order)
    s.update(make_cameras(N, H, W, device=device, seed=seed))
    s["pixelcoords"] = pixel_grid(N, H, W, device=device)
    s.update(N=N, H=H, W=W, K=K)
    return s
