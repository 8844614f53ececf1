import sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import ava256_amd as ops
from ava256_amd import _hooks as mm
from helpers import load_krt_400940, to_dev
campos, camrot, focal, princpt = [to_dev(x) for x in load_krt_400940()]
W, H = 1334, 2048
px, py = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32))
pc = torch.from_numpy(np.stack((px, py), -1))[None].cuda()
rp, rd, tm = ops.compute_raydirs(campos, camrot, focal, princpt, pc, 256.0)
print('campos/256', (campos/256).cpu().numpy(), 'tmin<tmax frac', (tm[...,0] < tm[...,1]).float().mean().item())
torch.manual_seed(0)
K = 128**2
dec = dict(template=torch.rand(1,K,8,8,8,4).cuda(), primpos=torch.rand(1,K,3).cuda(), primrot=torch.rand(1,K,3,3).cuda(), primscale=torch.rand(1,K,3).cuda())
diag = torch.zeros(8, dtype=torch.int32, device='cuda'); mm.set_diag_buffer(diag)
import time
with torch.no_grad():
    torch.cuda.synchronize(); t=time.time()
    out = ops.Raymarcher(256.0)(rp, rd, tm, dec)
    torch.cuda.synchronize(); print('time', time.time()-t)
print(mm.read_diag(), 'alpha mean', out[1].mean().item())
