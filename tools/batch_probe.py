"""Development probe: extend-kernel cost per ray against batch size, on incoherent bounce rays of cerealbar-synth.
Shows where a launch stops filling the chip (what deep bounces and small per-rank shares pay)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import raytracer_amd as ra
from raytracer_amd import synth, _ffi

sd = synth.build_scene(1.0)
sc = ra.Scene(sd)
cam = synth.cerealbar_camera()
W, H = cam.width, cam.height
n = W * H
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream().cuda_stream
f = np.array(cam.dir[:], np.float32); up = np.array(cam.up[:], np.float32)
r = np.cross(f, up); r /= np.linalg.norm(r); u = np.cross(r, f); u /= np.linalg.norm(u)
hy = np.tan(0.5 * cam.fov * np.pi / 180); hx = hy * W / H
xs, ys = np.meshgrid(np.arange(W), np.arange(H))
sx = (((xs + 0.5) / W) * 2 - 1) * hx; sy = (1 - ((ys + 0.5) / H) * 2) * hy
d = r[None, None] * sx[..., None] + u[None, None] * sy[..., None] + f[None, None]
d /= np.linalg.norm(d, axis=-1, keepdims=True)
hr = np.zeros((n, 8), np.float32); hr[:, 0:3] = np.array(cam.pos[:], np.float32); hr[:, 3] = 1e30; hr[:, 4:7] = d.reshape(-1, 3)
rays = torch.from_numpy(hr).to(dev)
hits = torch.empty((n, 4), dtype=torch.float32, device=dev)
sc.trace_closest_device(rays.data_ptr(), n, hits.data_ptr(), _ffi.HIT_FRONT_AND_BACK, True, stream)
torch.cuda.synchronize()
ok = hits[:, 3].view(torch.int32) >= 0
p = rays[ok, 0:3] + rays[ok, 4:7] * hits[ok, 0:1] * 0.999
g = torch.Generator(device=dev); g.manual_seed(1)
reps = 16
p = p.repeat(reps, 1)
dd = torch.randn((p.shape[0], 3), device=dev, generator=g); dd = dd / dd.norm(dim=1, keepdim=True)
b = torch.zeros((p.shape[0], 8), device=dev); b[:, 0:3] = p; b[:, 3] = 1e30; b[:, 4:7] = dd
b = b[torch.randperm(b.shape[0], device=dev)].contiguous()
hits = torch.empty((b.shape[0], 4), dtype=torch.float32, device=dev)
print("bounce rays available:", b.shape[0], flush=True)
for nn in (16384, 65536, 262144, 1 << 20, 1 << 22, 1 << 24, b.shape[0]):
    nn = min(nn, b.shape[0])
    k = max(3, min(50, (1 << 25) // nn))
    for _ in range(2): sc.trace_closest_device(b.data_ptr(), nn, hits.data_ptr(), _ffi.HIT_FRONT_AND_BACK, True, stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): sc.trace_closest_device(b.data_ptr(), nn, hits.data_ptr(), _ffi.HIT_FRONT_AND_BACK, True, stream)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / k
    print(f"{nn:>10} rays: {ms*1e3:9.1f} us/launch  {ms*1e6/nn:7.3f} ns/ray  {nn/ms/1e3:8.1f} Mrays/s", flush=True)
