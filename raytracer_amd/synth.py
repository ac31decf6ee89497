"""`cerealbar-synth`: procedural stand-in for the geometry of the reference's assets/cerealBar.xml.

The 22 OBJ files of the reference scene are Git-LFS pointers and `textures/bumpMap.png` is missing (SURVEY.md F3,
F4), so the geometry and the three textures are synthesised here from a fixed seed. Everything else is transcribed
from the scene file: the object list and its material bindings (assets/cerealBar.xml:5-51), the 17 Blinn materials
(:58-177), the environment colour and transform (:180-185), the three sphere lights (:188-207) and the camera
(:210-218). Triangle budgets per object follow SURVEY.md §8(d) (proportional to the LFS pointer sizes), times
`tri_scale`.
"""
import numpy as np

from . import _ffi
from .scene import SceneData, Mesh, Material, TexMap, Light, make_camera

SEED = 0xCE2EA1BA2

# object name -> (material, full-scale triangle budget)           assets/cerealBar.xml:5-51
OBJECTS = [
    ("teapots", "glass", 315_000), ("teapotLogos", "logo", 11_000), ("roundPlates", "porcelain", 15_000),
    ("squarePlates", "porcelain", 1_100), ("spoons", "metal", 12_000), ("bowls", "porcelain", 25_000),
    ("fruitLoopsRed", "fruitLoopsRed", 58_000), ("fruitLoopsBlue", "fruitLoopsBlue", 58_000),
    ("fruitLoopsGreen", "fruitLoopsGreen", 58_000), ("fruitLoopsYellow", "fruitLoopsYellow", 58_000),
    ("fruitLoopsPurple", "fruitLoopsPurple", 58_000), ("fruitLoopsOrange", "fruitLoopsOrange", 58_000),
    ("cocoaPuffs", "cocoaPuffs", 80_000), ("frostedFlakes", "frostedFlakes", 28_000),
    ("cookieCrispsCookies", "cookieCrisps", 1_600_000), ("cookieCrispsChips", "cocoaPuffs", 640_000),
    ("chex", "chex", 3_500_000), ("table", "table", 12), ("windowFramePillar", "plastic", 80),
    ("windowPanes", "glass", 20), ("backWall", "table", 2), ("bgPlate", "bgPlate", 80),
]


def _materials():
    """assets/cerealBar.xml:58-177 (colour = rgb * value; unspecified terms keep the framework defaults)."""
    W = (1.0, 1.0, 1.0)
    Z = (0.0, 0.0, 0.0)

    def cereal(name, kd, gloss, bump):
        return Material(name, diffuse=kd, specular=W, emission=Z, glossiness=gloss, bump_tex=0, bump_scale=bump)

    return [
        Material("glass", diffuse=Z, specular=Z, glossiness=100000.0, refraction=W, ior=1.5, emission=Z,
                 absorption=(0.75, 0.15, 0.15)),
        Material("logo", diffuse=(0.75, 0.0, 0.0), specular=W, glossiness=500.0, emission=Z),
        Material("porcelain", diffuse=W, specular=W, glossiness=1000.0, reflection=(0.3, 0.3, 0.3), emission=Z),
        Material("metal", diffuse=(0.05, 0.05, 0.05), specular=W, glossiness=300.0, reflection=W, emission=Z),
        cereal("fruitLoopsRed", (0.677, 0.033173, 0.033173), 300.0, 2.0),
        cereal("fruitLoopsBlue", (0.079827, 0.451, 0.451), 300.0, 2.0),
        cereal("fruitLoopsGreen", (0.213105, 0.523, 0.100416), 300.0, 2.0),
        cereal("fruitLoopsYellow", (0.735, 0.735, 0.09849), 300.0, 2.0),
        cereal("fruitLoopsOrange", (1.0, 0.366667, 0.0), 300.0, 2.0),
        cereal("fruitLoopsPurple", (0.11, 0.06, 0.5), 300.0, 2.0),
        cereal("cocoaPuffs", (0.169, 0.0337437, 0.00338), 300.0, 3.0),
        cereal("frostedFlakes", (0.884, 0.475592, 0.125645), 200.0, 2.0),
        cereal("cookieCrisps", (0.826, 0.480732, 0.28084), 300.0, 5.0),
        cereal("chex", (1.0, 0.5695, 0.262), 100.0, 1.0),
        Material("table", diffuse=(0.689, 0.0918667, 0.0), specular=W, glossiness=500.0, emission=Z),
        Material("plastic", diffuse=(0.01764, 0.0275707, 0.02), specular=(0.3, 0.3, 0.3), glossiness=500.0, emission=Z),
        Material("bgPlate", diffuse=W, diffuse_tex=1, specular=Z, glossiness=1.0, emission=W, emission_tex=1),
    ]


# ---- textures (synthetic, seeded) -------------------------------------------------------------------------
def _value_noise(rng, h, w, cells):
    g = rng.random((cells + 1, cells + 1)).astype(np.float32)
    g[-1, :] = g[0, :]
    g[:, -1] = g[:, 0]
    ys = np.linspace(0, cells, h, endpoint=False, dtype=np.float32)
    xs = np.linspace(0, cells, w, endpoint=False, dtype=np.float32)
    y0 = ys.astype(np.int32)
    x0 = xs.astype(np.int32)
    fy = (ys - y0)[:, None]
    fx = (xs - x0)[None, :]
    fy = fy * fy * (3 - 2 * fy)
    fx = fx * fx * (3 - 2 * fx)
    a = g[y0][:, x0]
    b = g[y0][:, x0 + 1]
    c = g[y0 + 1][:, x0]
    d = g[y0 + 1][:, x0 + 1]
    return (a * (1 - fx) * (1 - fy) + b * fx * (1 - fy) + c * (1 - fx) * fy + d * fx * fy).astype(np.float32)


def make_textures(seed=SEED):
    """[0] bump map 512x512 RGB8 (stands in for textures/bumpMap.png), [1] bgPlate 1280x853 RGB8,
    [2] environment 1024x512 RGB32F (stands in for alps_field_1k.hdr)."""
    rng = np.random.default_rng(seed ^ 0x7E77)
    n = 0.6 * _value_noise(rng, 512, 512, 32) + 0.4 * _value_noise(rng, 512, 512, 96)
    bump = np.repeat((np.clip(n, 0, 1) * 255).astype(np.uint8)[:, :, None], 3, axis=2)
    yy, xx = np.meshgrid(np.linspace(0, 1, 853, dtype=np.float32), np.linspace(0, 1, 1280, dtype=np.float32), indexing="ij")
    blobs = _value_noise(rng, 853, 1280, 12)
    bg = np.stack([0.35 + 0.4 * yy + 0.2 * blobs, 0.45 + 0.3 * yy + 0.15 * blobs, 0.6 + 0.25 * (1 - yy) + 0.1 * blobs], axis=2)
    bgplate = (np.clip(bg, 0, 1) * 255).astype(np.uint8)
    v, u = np.meshgrid(np.linspace(0, 1, 512, dtype=np.float32), np.linspace(0, 1, 1024, dtype=np.float32), indexing="ij")
    sky = np.stack([0.25 + 0.5 * v, 0.4 + 0.45 * v, 0.8 + 0.1 * v], axis=2)
    sky = np.where((v > 0.55)[:, :, None], np.array([0.18, 0.22, 0.1], dtype=np.float32)[None, None, :] * (1.2 - v)[:, :, None], sky)
    sun = np.exp(-(((u - 0.3) / 0.02) ** 2 + ((v - 0.25) / 0.03) ** 2))
    env = (sky + 40.0 * sun[:, :, None] * np.array([1.0, 0.9, 0.7], dtype=np.float32)).astype(np.float32)
    env += 0.05 * _value_noise(rng, 512, 1024, 16)[:, :, None]
    return bump, bgplate, np.ascontiguousarray(env, dtype=np.float32)


def _rot(axis, deg):
    a = np.deg2rad(deg)
    c, s = np.cos(a), np.sin(a)
    if axis == 0:
        return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])
    if axis == 1:
        return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])


def _xf34(m3, t=(0, 0, 0)):
    out = np.zeros((3, 4), dtype=np.float64)
    out[:, :3] = m3
    out[:, 3] = t
    return out.astype(np.float32).reshape(12)


# ---- parametric pieces -------------------------------------------------------------------------------------
def grid_mesh(P, N, nu, nv, wrap_u=False, wrap_v=False):
    """Triangulates a (nu+1) x (nv+1) parametric grid. P, N: callables (u, v) -> (..., 3)."""
    u = np.linspace(0, 1, nu + 1)
    v = np.linspace(0, 1, nv + 1)
    U, Vv = np.meshgrid(u, v, indexing="ij")
    V = P(U, Vv).reshape(-1, 3)
    Nn = N(U, Vv).reshape(-1, 3)
    UV = np.stack([U, Vv, np.zeros_like(U)], axis=-1).reshape(-1, 3)
    i, j = np.meshgrid(np.arange(nu), np.arange(nv), indexing="ij")
    a = (i * (nv + 1) + j).ravel()
    b = ((i + 1) * (nv + 1) + j).ravel()
    c = ((i + 1) * (nv + 1) + j + 1).ravel()
    d = (i * (nv + 1) + j + 1).ravel()
    F = np.concatenate([np.stack([a, b, c], 1), np.stack([a, c, d], 1)], 0)
    return V.astype(np.float32), Nn.astype(np.float32), UV.astype(np.float32), F.astype(np.int32)


def torus(R, r, nu, nv):
    def P(u, v):
        a, b = 2 * np.pi * u, 2 * np.pi * v
        return np.stack([(R + r * np.cos(b)) * np.cos(a), r * np.sin(b), (R + r * np.cos(b)) * np.sin(a)], -1)

    def N(u, v):
        a, b = 2 * np.pi * u, 2 * np.pi * v
        return np.stack([np.cos(b) * np.cos(a), np.sin(b), np.cos(b) * np.sin(a)], -1)

    # orientation: make e1 x e2 point outward
    V, Nn, UV, F = grid_mesh(P, N, nu, nv)
    return V, Nn, UV, F[:, ::-1].copy()


def ellipsoid(rx, ry, rz, nu, nv, power=1.0, flip=False):
    """Superellipsoid-ish closed shell; power<1 squares it off (chex pillows)."""
    def sp(x):
        return np.sign(x) * np.abs(x) ** power

    def P(u, v):
        a, b = 2 * np.pi * u, np.pi * (v - 0.5)
        return np.stack([rx * sp(np.cos(b)) * sp(np.cos(a)), ry * sp(np.sin(b)), rz * sp(np.cos(b)) * sp(np.sin(a))], -1)

    def N(u, v):
        a, b = 2 * np.pi * u, np.pi * (v - 0.5)
        n = np.stack([np.cos(b) * np.cos(a) / rx, np.sin(b) / ry, np.cos(b) * np.sin(a) / rz], -1)
        return n / np.linalg.norm(n, axis=-1, keepdims=True)

    V, Nn, UV, F = grid_mesh(P, N, nu, nv)
    if flip:
        return V, -Nn, UV, F
    return V, Nn, UV, F[:, ::-1].copy()


def wavy_disc(r, amp, nu, nv):
    def P(u, v):
        a = 2 * np.pi * u
        rr = r * v
        return np.stack([rr * np.cos(a), amp * np.sin(3 * a) * v * v, rr * np.sin(a)], -1)

    def N(u, v):
        return np.broadcast_to(np.array([0.0, 1.0, 0.0]), u.shape + (3,)).copy()

    V, Nn, UV, F = grid_mesh(P, N, nu, nv)
    # two-sided thin flake: duplicate with flipped winding slightly below
    V2 = V.copy()
    V2[:, 1] -= 0.004
    return (np.concatenate([V, V2]), np.concatenate([Nn, -Nn]), np.concatenate([UV, UV]),
            np.concatenate([F, F[:, ::-1] + len(V)]))


def bowl(r, nu, nv, thickness=0.03):
    """Open hemispherical shell: outer surface, inner surface, rim."""
    def Po(u, v):
        a, b = 2 * np.pi * u, 0.5 * np.pi * v  # b: 0 = bottom pole, pi/2 = rim
        return np.stack([r * np.sin(b) * np.cos(a), r - r * np.cos(b), r * np.sin(b) * np.sin(a)], -1)

    def No(u, v):
        a, b = 2 * np.pi * u, 0.5 * np.pi * v
        return np.stack([np.sin(b) * np.cos(a), -np.cos(b), np.sin(b) * np.sin(a)], -1)

    ri = r - thickness

    def Pi(u, v):
        a, b = 2 * np.pi * u, 0.5 * np.pi * v
        return np.stack([ri * np.sin(b) * np.cos(a), r - ri * np.cos(b), ri * np.sin(b) * np.sin(a)], -1)

    Vo, No_, UVo, Fo = grid_mesh(Po, No, nu, nv)
    Vi, Ni_, UVi, Fi = grid_mesh(Pi, lambda u, v: -No(u, v), nu, nv)
    # rim ring
    def Pr(u, v):
        a = 2 * np.pi * u
        rr = ri + (r - ri) * v
        return np.stack([rr * np.cos(a), np.full_like(a, r), rr * np.sin(a)], -1)

    Vr, Nr, UVr, Fr = grid_mesh(Pr, lambda u, v: np.broadcast_to(np.array([0.0, 1.0, 0.0]), u.shape + (3,)).copy(), nu, 1)
    V = np.concatenate([Vo, Vi, Vr])
    N = np.concatenate([No_, Ni_, Nr])
    UV = np.concatenate([UVo, UVi, UVr])
    F = np.concatenate([Fo[:, ::-1], Fi + len(Vo), Fr + len(Vo) + len(Vi)])
    return V.astype(np.float32), N.astype(np.float32), UV.astype(np.float32), F.astype(np.int32)


def box(lo, hi):
    lo, hi = np.asarray(lo, dtype=np.float32), np.asarray(hi, dtype=np.float32)
    c = np.array([[lo[0], lo[1], lo[2]], [hi[0], lo[1], lo[2]], [hi[0], hi[1], lo[2]], [lo[0], hi[1], lo[2]],
                  [lo[0], lo[1], hi[2]], [hi[0], lo[1], hi[2]], [hi[0], hi[1], hi[2]], [lo[0], hi[1], hi[2]]], dtype=np.float32)
    quads = [(0, 3, 2, 1, (0, 0, -1)), (4, 5, 6, 7, (0, 0, 1)), (0, 1, 5, 4, (0, -1, 0)), (3, 7, 6, 2, (0, 1, 0)),
             (0, 4, 7, 3, (-1, 0, 0)), (1, 2, 6, 5, (1, 0, 0))]
    V, N, UV, F = [], [], [], []
    for a, b, cc, d, n in quads:
        base = len(V)
        V += [c[a], c[b], c[cc], c[d]]
        N += [n] * 4
        UV += [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0)]
        F += [(base, base + 1, base + 2), (base, base + 2, base + 3)]
    return np.array(V, np.float32), np.array(N, np.float32), np.array(UV, np.float32), np.array(F, np.int32)


def quad_grid(p0, du, dv, nu, nv):
    p0, du, dv = (np.asarray(x, dtype=np.float64) for x in (p0, du, dv))
    n = np.cross(du, dv)
    n = n / np.linalg.norm(n)
    return grid_mesh(lambda u, v: p0 + u[..., None] * du + v[..., None] * dv,
                     lambda u, v: np.broadcast_to(n, u.shape + (3,)).copy(), nu, nv)


def _random_rotations(rng, k):
    q = rng.normal(size=(k, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    return np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], 1),
                     np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], 1),
                     np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], 1)], 1)


def instance(template, R, T, S=None):
    """template (V, N, UV, F) placed K times: V' = R_k (S_k V) + T_k."""
    V, N, UV, F = template
    K = R.shape[0]
    Vs = V[None, :, :] * (1.0 if S is None else S[:, None, None])
    Vk = np.einsum("kij,knj->kni", R, Vs) + T[:, None, :]
    Nk = np.einsum("kij,nj->kni", R, N)
    UVk = np.broadcast_to(UV[None], (K,) + UV.shape)
    Fk = F[None, :, :] + (np.arange(K) * V.shape[0])[:, None, None]
    return (Vk.reshape(-1, 3).astype(np.float32), Nk.reshape(-1, 3).astype(np.float32),
            np.ascontiguousarray(UVk.reshape(-1, 3), dtype=np.float32), Fk.reshape(-1, 3).astype(np.int32))


def merge(parts):
    V, N, UV, F = [], [], [], []
    base = 0
    for v, n, uv, f in parts:
        V.append(v)
        N.append(n)
        UV.append(uv)
        F.append(f + base)
        base += v.shape[0]
    return np.concatenate(V), np.concatenate(N), np.concatenate(UV), np.concatenate(F)


# piles: (centre x, z, radius, base height)
BOWLS = [(0.9, 0.0, 0.45), (-0.35, -1.25, 0.42), (2.05, -1.1, 0.45), (0.1, 1.3, 0.38), (1.9, 0.9, 0.4)]
PLATES = [(-0.9, 0.4, 0.5), (3.0, -0.2, 0.5)]


def _pile_positions(rng, k, sites, spread_table=0.25):
    """Random points inside the bowls' lower volume (plus a fraction scattered over the table)."""
    site = rng.integers(0, len(sites), size=k)
    cx = np.array([s[0] for s in sites])[site]
    cz = np.array([s[1] for s in sites])[site]
    rad = np.array([s[2] for s in sites])[site]
    a = rng.random(k) * 2 * np.pi
    rr = np.sqrt(rng.random(k)) * rad * 0.8
    h = rng.random(k)
    # inside the bowl the floor rises with radius: y_floor = r - sqrt(r^2 - rr^2)
    floor = rad - np.sqrt(np.maximum(rad * rad - rr * rr, 0.0))
    y = floor + 0.03 + h * (rad * 0.75 - floor).clip(0.02)
    P = np.stack([cx + rr * np.cos(a), y, cz + rr * np.sin(a)], 1)
    loose = rng.random(k) < spread_table
    P[loose, 0] = rng.uniform(-1.4, 3.4, size=loose.sum())
    P[loose, 2] = rng.uniform(-2.4, 2.6, size=loose.sum())
    P[loose, 1] = 0.03
    return P


def build_meshes(tri_scale=1.0, seed=SEED, only=None):
    """Returns a list of (name, material name, V, N, UV, F) in the XML's object order."""
    rng = np.random.default_rng(seed)
    out = []
    for name, mtl, budget in OBJECTS:
        if only is not None and name not in only:
            # keep the RNG stream independent of `only`
            rng = np.random.default_rng(rng.integers(1 << 62))
            continue
        sub = np.random.default_rng(rng.integers(1 << 62))
        rng = np.random.default_rng(rng.integers(1 << 62))
        want = max(2, int(budget * tri_scale))

        def cereal(template, sites, smin=0.85, smax=1.15, loose=0.25):
            k = max(1, want // template[3].shape[0])
            return instance(template, _random_rotations(sub, k), _pile_positions(sub, k, sites, loose), sub.uniform(smin, smax, k))

        if name.startswith("fruitLoops"):
            geo = cereal(torus(0.035, 0.014, 16, 8), BOWLS[:2] + PLATES[:1])
        elif name == "cocoaPuffs":
            geo = cereal(ellipsoid(0.03, 0.03, 0.03, 12, 8), BOWLS[2:3])
        elif name == "frostedFlakes":
            geo = cereal(wavy_disc(0.05, 0.012, 12, 3), BOWLS[3:4])
        elif name == "cookieCrispsCookies":
            geo = cereal(ellipsoid(0.04, 0.012, 0.04, 16, 8), BOWLS[4:5] + PLATES[1:], loose=0.35)
        elif name == "cookieCrispsChips":
            geo = cereal(ellipsoid(0.008, 0.005, 0.008, 8, 4), BOWLS[4:5] + PLATES[1:], loose=0.35)
        elif name == "chex":
            geo = cereal(ellipsoid(0.04, 0.012, 0.04, 16, 12, power=0.45), BOWLS + PLATES, loose=0.4)
        elif name == "bowls":
            n = max(4, int(np.sqrt(want / (len(BOWLS) * 4.5))))
            parts = []
            for x, z, r in BOWLS:
                v, nn, uv, f = bowl(r, 2 * n, n)
                parts.append((v + np.array([x, 0.0, z], np.float32), nn, uv, f))
            geo = merge(parts)
        elif name == "roundPlates":
            n = max(6, int(np.sqrt(want / (len(PLATES) * 4))))
            parts = []
            for x, z, r in PLATES:
                v, nn, uv, f = ellipsoid(r, 0.02, r, 2 * n, n)
                parts.append((v + np.array([x, 0.021, z], np.float32), nn, uv, f))
            geo = merge(parts)
        elif name == "squarePlates":
            n = max(3, int(np.sqrt(want / 4)))
            v, nn, uv, f = ellipsoid(0.45, 0.02, 0.45, 2 * n, n, power=0.3)
            geo = (v + np.array([1.0, 0.021, -2.2], np.float32), nn, uv, f)
        elif name == "spoons":
            n = max(4, int(np.sqrt(want / 8)))
            parts = []
            for x, z, ang in [(0.2, 0.55, 20.0), (2.6, 0.5, -35.0)]:
                Rm = _rot(1, ang).astype(np.float32)
                v, nn, uv, f = ellipsoid(0.07, 0.015, 0.045, 2 * n, n)
                hv, hn, huv, hf = ellipsoid(0.16, 0.008, 0.012, 2 * n, n)
                hv = hv + np.array([0.2, 0.0, 0.0], np.float32)
                for vv, nv_, uvv, ff in ((v, nn, uv, f), (hv, hn, huv, hf)):
                    parts.append((vv @ Rm.T + np.array([x, 0.02, z], np.float32), nv_ @ Rm.T, uvv, ff))
            geo = merge(parts)
        elif name == "teapots":
            n = max(6, int(np.sqrt(want / (2 * 2 * 4 + 2 * 2))))
            parts = []
            for x, z in [(1.45, 0.75), (-0.2, -0.2)]:
                c = np.array([x, 0.27, z], np.float32)
                ov, on, ouv, of = ellipsoid(0.3, 0.26, 0.3, 2 * n, n)
                iv, inn, iuv, if_ = ellipsoid(0.28, 0.24, 0.28, 2 * n, n, flip=True)
                parts += [(ov + c, on, ouv, of), (iv + c, inn, iuv, if_)]
                tv, tn, tuv, tf = torus(0.12, 0.025, max(8, n), max(6, n // 2))
                Rz = _rot(0, 90.0).astype(np.float32)
                parts.append((tv @ Rz.T + c + np.array([0.36, 0.02, 0.0], np.float32), tn @ Rz.T, tuv, tf))
            geo = merge(parts)
        elif name == "teapotLogos":
            n = max(4, int(np.sqrt(want / 8)))
            parts = []
            for x, z in [(1.45, 0.75), (-0.2, -0.2)]:
                v, nn, uv, f = ellipsoid(0.08, 0.08, 0.004, 2 * n, n)
                parts.append((v + np.array([x, 0.3, z + 0.302], np.float32), nn, uv, f))
            geo = merge(parts)
        elif name == "table":
            geo = box((-3.0, -0.1, -4.0), (5.0, 0.0, 4.0))
        elif name == "backWall":
            geo = quad_grid((-3.0, 0.0, -3.6), (8.0, 0.0, 0.0), (0.0, 4.0, 0.0), 1, 1)
            # the wall has a window opening in the real scene; here it stops below it
            geo = (geo[0] * np.array([1, 0.2, 1], np.float32), geo[1], geo[2], geo[3])
        elif name == "windowFramePillar":
            geo = merge([box((x - 0.06, 0.8, -3.62), (x + 0.06, 4.0, -3.5)) for x in (-3.0, -0.3, 2.3, 5.0)] +
                        [box((-3.0, 0.74, -3.62), (5.0, 0.8, -3.5))] + [box((-3.0, 2.4, -3.6), (5.0, 2.46, -3.52))])
        elif name == "windowPanes":
            geo = box((-3.0, 0.8, -3.58), (5.0, 4.0, -3.56))
        elif name == "bgPlate":
            geo = quad_grid((-8.0, -1.0, -9.0), (18.0, 0.0, 0.0), (0.0, 9.0, 0.0), 8, 5)
        else:
            raise KeyError(name)
        out.append((name, mtl) + tuple(geo))
    return out


def build_scene(tri_scale=1.0, seed=SEED, with_textures=True):
    """The cerealbar-synth SceneData (all 22 objects, identity transforms) and its camera at 1920x1080."""
    mats = _materials()
    mat_index = {m.name: i for i, m in enumerate(mats)}
    sd = SceneData()
    sd.materials = mats
    for name, mtl, V, N, UV, F in build_meshes(tri_scale, seed):
        sd.meshes.append(Mesh(V, F, vn=N, vt=UV, material=mat_index[mtl], name=name))
    bump, bgplate, env = make_textures(seed)
    flip_y = _xf34(np.diag([1.0, -1.0, 1.0]))                      # <scale x="1" y="-1" z="1"/>  (:170,175)
    env_m = np.diag([30.0] * 3) @ _rot(2, 30.0) @ _rot(1, 90.0) @ _rot(0, 180.0)   # :181-184
    sd.texmaps = [TexMap(bump), TexMap(bgplate, flip_y), TexMap(env, _xf34(np.linalg.inv(env_m)))]
    if not with_textures:
        for m in sd.materials:
            m.bump_tex = -1
    sd.environment = (0.5 * 5, 0.6 * 5, 0.7 * 5)                    # :180
    sd.environment_tex = 2
    I = (1.0 * 5, 0.549925 * 5, 0.235 * 5)                          # :189
    sd.lights = [Light((1.9, 6, 2.9), I, 1.0, 1.0), Light((-0.8, 8, 0.1), I, 1.0, 1.0), Light((1.1, 6, -2.2), I, 1.0, 1.0)]
    return sd


def cerealbar_camera(width=1920, height=1080):
    """assets/cerealBar.xml:210-218"""
    return make_camera((-1.6766, 0.6769, 4.6929), (0.88, 0.3022, 0.0), (0, 1, 0), fov=24.0, focaldist=3.5, dof=0.02,
                       width=width, height=height, srgb=True)


def single_mesh_scene(name="teapots", tri_scale=1.0, seed=SEED):
    """Config C1: one generated mesh, no lights, with a framing pinhole camera (512x512)."""
    sd = SceneData()
    sd.materials = _materials()
    mat_index = {m.name: i for i, m in enumerate(sd.materials)}
    for n, mtl, V, N, UV, F in build_meshes(tri_scale, seed, only=(name,)):
        sd.meshes.append(Mesh(V, F, vn=N, vt=UV, material=mat_index[mtl], name=n))
    for m in sd.materials:
        m.bump_tex = -1
        m.diffuse_tex = m.emission_tex = -1
    V = sd.meshes[0].v
    lo, hi = V.min(0), V.max(0)
    c = 0.5 * (lo + hi)
    ext = float(np.linalg.norm(hi - lo))
    cam = make_camera(c + np.array([0.35, 0.45, 1.0], np.float32) * ext * 1.1, c, (0, 1, 0), fov=40.0, width=512, height=512)
    return sd, cam
