"""Development probe: extend-kernel throughput on cerealbar-synth primary and random rays (not the judged bench)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import raytracer_amd as ra
from raytracer_amd import synth, _ffi

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
t0 = time.time(); sd = synth.build_scene(scale); print("gen", sd.n_triangles, "tris", round(time.time() - t0, 2), "s", flush=True)
t0 = time.time(); sc = ra.Scene(sd); print("scene_create", round(time.time() - t0, 2), "s", sc.info(), flush=True)
cam = synth.cerealbar_camera()
W, H = cam.width, cam.height
# primary rays via the library (pixel centres), kept resident in HBM
n = W * H
dev = torch.device("cuda:0")
rays = torch.empty((n, 8), dtype=torch.float32, device=dev)
hits = torch.empty((n, 4), dtype=torch.float32, device=dev)
# fill rays on host using the same camera math (pinhole, pixel centres)
import ctypes as C
f = np.array(cam.dir[:], np.float32); up = np.array(cam.up[:], np.float32)
r = np.cross(f, up); r /= np.linalg.norm(r); u = np.cross(r, f); u /= np.linalg.norm(u)
hy = np.tan(0.5 * cam.fov * np.pi / 180); hx = hy * W / H
xs, ys = np.meshgrid(np.arange(W), np.arange(H))
sx = (((xs + 0.5) / W) * 2 - 1) * hx; sy = (1 - ((ys + 0.5) / H) * 2) * hy
d = r[None, None] * sx[..., None] + u[None, None] * sy[..., None] + f[None, None]
d /= np.linalg.norm(d, axis=-1, keepdims=True)
hr = np.zeros((n, 8), np.float32); hr[:, 0:3] = np.array(cam.pos[:], np.float32); hr[:, 3] = 1e30; hr[:, 4:7] = d.reshape(-1, 3)
rays.copy_(torch.from_numpy(hr))
stream = torch.cuda.current_stream().cuda_stream

def run(tag, rays, k=10):
    nn = rays.shape[0]
    sc.set_counters(True)
    sc.trace_closest_device(rays.data_ptr(), nn, hits.data_ptr(), _ffi.HIT_FRONT_AND_BACK, True, stream)
    if hasattr(sc.lib, "rtx_debug_trav_profile"):   # variant built with -DTRAV_PROFILE
        prof = (C.c_uint64 * 8)()
        sc.lib.rtx_debug_trav_profile.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        sc.lib.rtx_debug_trav_profile(sc.h, prof)
        p = list(prof)
        print(f"   steps/ray: inner {p[3]*64/p[2]:.1f} (avg lanes {p[4]/max(p[3],1):.1f})  leaf {p[5]*64/p[2]:.1f} (avg lanes {p[6]/max(p[5],1):.1f})"
              f"  ideal inner {p[0]/p[2]:.1f} leaf-visits/ray ~{p[1]/p[2]/3:.1f}", flush=True)
        cnt = {"nodes": p[0], "tris": p[1], "rays": p[2]}
    else:
        cnt = sc.get_counters()
    sc.set_counters(False)
    for _ in range(2): sc.trace_closest_device(rays.data_ptr(), nn, hits.data_ptr(), _ffi.HIT_FRONT_AND_BACK, True, stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): sc.trace_closest_device(rays.data_ptr(), nn, hits.data_ptr(), _ffi.HIT_FRONT_AND_BACK, True, stream)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / k
    npr, tpr = cnt["nodes"] / cnt["rays"], cnt["tris"] / cnt["rays"]
    bpr = 48 + npr * 64 + tpr * 48
    hitfrac = float((hits[:nn, 3].view(torch.int32) != -1).float().mean())
    print(f"{tag}: {nn} rays {ms:.3f} ms  {nn/ms/1e3:.1f} Mrays/s  nodes/ray {npr:.1f} tris/ray {tpr:.1f} "
          f"bytes/ray {bpr:.0f}  algo {nn*bpr/ms/1e6:.1f} GB/s ({nn*bpr/ms/1e6/8000*100:.1f}% of 8 TB/s) hit {hitfrac:.3f}", flush=True)

run("primary-1080p", rays)
# incoherent: shuffle the same rays
perm = torch.randperm(n, device=dev)
run("primary-shuffled", rays[perm].contiguous())
# diffuse bounce rays: from hit points in random directions
hh = hits.clone(); ok = hh[:, 3].view(torch.int32) >= 0
p = rays[ok, 0:3] + rays[ok, 4:7] * hh[ok, 0:1] * 0.999
g = torch.Generator(device=dev); g.manual_seed(1)
dd = torch.randn((p.shape[0], 3), device=dev, generator=g); dd = dd / dd.norm(dim=1, keepdim=True)
b = torch.zeros((p.shape[0], 8), device=dev); b[:, 0:3] = p; b[:, 3] = 1e30; b[:, 4:7] = dd
run("bounce-random", b.contiguous())
