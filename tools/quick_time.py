import sys, time, torch
sys.path.insert(0, '.')
from oracle import cpu_ref as O
from triplaneturbo_amd import ops, functional
g = torch.Generator().manual_seed(0)
R, Hh, Ww, S = 256, 256, 256, 128
cache = (torch.randn(1, 6, 32, R, R, generator=g) * 0.5).cuda().requires_grad_(True)
sw = [w.cuda().requires_grad_(True) for w in O.init_mlp_weights([32, 64, 64, 1], g)]
fw = [w.cuda().requires_grad_(True) for w in O.init_mlp_weights([96, 64, 64, 3], g)]
ro, rd, c2w, cd = O.make_cameras(1, Hh, Ww)
ro, rd, c2w, cd = ro.cuda(), rd.cuda(), c2w.cuda(), cd.cuda()
ts, te = [t.cuda() for t in O.uniform_intervals(Hh * Ww, S, 0.1, 4.0)]
bg = torch.ones(3).cuda()
rc = ops.RenderConfig()
proj = {k: torch.randn(1, Hh, Ww, c, generator=g).cuda() for k, c in (("comp_rgb", 3), ("opacity", 1), ("depth", 1), ("disparity", 1), ("comp_normal_cam_vis", 3))}
def step():
    for t in [cache] + sw + fw: t.grad = None
    out = functional.volume_render(cache, sw, fw, ro, rd, ts, te, bg, cd, c2w, rc, training=True)
    loss = O.synthetic_loss(out, proj)
    loss.backward()
    return out, loss
for _ in range(2): step()
torch.cuda.synchronize()
t0 = time.time()
K = 5
for _ in range(K): out, loss = step()
torch.cuda.synchronize()
dt = (time.time() - t0) / K
print("ms/step", dt * 1e3, "rays/s", Hh * Ww / dt, "loss", loss.item(), "opacity mean", out["opacity"].mean().item())
