"""Steps per ray of the traversal-order model for two tree layouts on the bench scene (host only).
  python tools/bvh_walk_probe.py /tmp/libbvhprobe4.so /tmp/libbvhprobe8.so [tri_scale] [n_rays]
Rays: diffuse-bounce-like (origin on a random triangle, offset 1e-4 along its normal, cosine-ish direction about it) and
camera rays of the bench view; both checked for identical hits between the layouts (the result must not depend on the tree)."""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from raytracer_amd import synth  # noqa: E402

libs = [a for a in sys.argv[1:] if a.endswith(".so")]
rest = [a for a in sys.argv[1:] if not a.endswith(".so")]
tri_scale = float(rest[0]) if rest else 1.0
n = int(rest[1]) if len(rest) > 1 else 200000
sd = synth.build_scene(tri_scale=tri_scale, with_textures=False)
desc, keep = sd.to_desc()
rng = np.random.default_rng(7)
V = np.concatenate([m.v[m.f] for m in sd.meshes if m.f.shape[0] > 0])          # (T, 3, 3)
pick = rng.integers(0, V.shape[0], n)
tri = V[pick]
w = rng.random((n, 3)).astype(np.float32)
w /= w.sum(1, keepdims=True)
P = (tri * w[:, :, None]).sum(1)
N = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
N /= np.maximum(np.linalg.norm(N, axis=1, keepdims=True), 1e-30)
N *= np.where(rng.random((n, 1)) < 0.5, 1.0, -1.0)
D = rng.normal(size=(n, 3))
D /= np.linalg.norm(D, axis=1, keepdims=True)
D = np.where((D * N).sum(1, keepdims=True) < 0, -D, D)
rays = np.zeros((n, 8), np.float32)
rays[:, 0:3] = P + 1e-4 * N
rays[:, 3] = 1e30
rays[:, 4:7] = D
cam = synth.cerealbar_camera()
# camera rays of the bench view (pixel centres of a coarse grid)
Wc, Hc = 640, 360
f = np.array(cam.dir, np.float32); up = np.array(cam.up, np.float32)
r = np.cross(f, up); r /= np.linalg.norm(r); u = np.cross(r, f); u /= np.linalg.norm(u)
hy = np.tan(0.5 * np.radians(cam.fov)); hx = hy * cam.width / cam.height
xs, ys = np.meshgrid((np.arange(Wc) + 0.5) / Wc * 2 - 1, 1 - (np.arange(Hc) + 0.5) / Hc * 2)
dirs = r[None, None] * (xs * hx)[..., None] + u[None, None] * (ys * hy)[..., None] + f[None, None]
dirs /= np.linalg.norm(dirs, axis=2, keepdims=True)
crays = np.zeros((Wc * Hc, 8), np.float32)
crays[:, 0:3] = np.array(cam.pos, np.float32)
crays[:, 3] = 1e30
crays[:, 4:7] = dirs.reshape(-1, 3)
res = {}
for path in libs:
    lib = C.CDLL(path)
    for name, rr in (("bounce", rays), ("camera", crays)):
        t = np.zeros(rr.shape[0], np.float32); prim = np.zeros(rr.shape[0], np.int32); cnt = (C.c_uint64 * 6)()
        t0 = time.time()
        rc = lib.bvh_probe(C.byref(desc), rr.ctypes.data_as(C.c_void_p), C.c_size_t(rr.shape[0]), t.ctypes.data_as(C.c_void_p),
                           prim.ctypes.data_as(C.c_void_p), cnt)
        assert rc == 0, rc
        res[(path, name)] = (t, prim)
        print(f"{path} {name}: nodes/ray {cnt[0] / rr.shape[0]:.2f} tris/ray {cnt[1] / rr.shape[0]:.2f}  tree nodes {cnt[2]} x {cnt[4]} B depth {cnt[3]}"
              f"  bytes/ray {48 + cnt[0] / rr.shape[0] * cnt[4] + cnt[1] / rr.shape[0] * 48:.0f}  hit {np.mean(prim >= 0):.3f}  ({time.time() - t0:.1f} s)", flush=True)
if len(libs) == 2:
    for name in ("bounce", "camera"):
        a, b = res[(libs[0], name)], res[(libs[1], name)]
        print(name, "identical hits:", bool(np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0])))
