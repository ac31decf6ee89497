"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol include/rtx.h declares, its POD
structs have the sizes the Python mirror assumes, the host-side flattening logic (BVH build, tile sharding) is
correct, and compute entry points fail loudly without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import numpy as np
import pytest

import raytracer_amd as ra
from raytracer_amd import _ffi, synth
from raytracer_amd.scene import SceneData, Mesh, Material

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rtx.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rtx_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = _ffi.load_library()
    names = declared_functions()
    assert len(names) >= 25
    bound = {n for n, _, _ in _ffi.SYMBOLS}
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/rtx.h but not exported"
        assert n in bound, f"{n} declared in include/rtx.h but not bound in raytracer_amd/_ffi.py"
    assert lib.rtx_version().decode().endswith("(gfx950)")


def test_struct_sizes_match_header():
    names = ["rtx_mesh_desc", "rtx_texmap_desc", "rtx_material_desc", "rtx_light_desc", "rtx_scene_desc", "rtx_camera",
             "rtx_ray", "rtx_hit", "rtx_render_params", "rtx_film", "rtx_stats", "rtx_photon_params", "rtx_photon"]
    mirror = [_ffi.MeshDesc, _ffi.TexMapDesc, _ffi.MaterialDesc, _ffi.LightDesc, _ffi.SceneDesc, _ffi.Camera, _ffi.Ray,
              _ffi.Hit, _ffi.RenderParams, _ffi.Film, _ffi.Stats, _ffi.PhotonParams, _ffi.Photon]
    prog = '#include <stdio.h>\n#include "rtx.h"\nint main(){' + "".join(f'printf("%zu\\n", sizeof({n}));' for n in names) + "return 0;}"
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "s.c"), "w").write(prog)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "s.c"), "-o", os.path.join(d, "s")])
        sizes = [int(x) for x in subprocess.check_output([os.path.join(d, "s")]).split()]
    assert sizes == [C.sizeof(m) for m in mirror]


def _needs_no_gpu():
    if ra.device_count() > 0:
        pytest.skip("a GPU is present")


def test_compute_fails_loudly_without_device():
    """No CPU fallback: without a HIP device every compute entry point reports RTX_ERR_NO_DEVICE."""
    _needs_no_gpu()
    lib = _ffi.load_library()
    sd, cam = synth.single_mesh_scene("teapots", 0.005)
    with pytest.raises(ra.RtxError, match="no HIP device"):
        ra.Scene(sd)
    assert lib.rtx_set_device(0) == -2
    assert b"no HIP device" in lib.rtx_last_error()


def test_missing_library_is_an_error(tmp_path):
    with pytest.raises(RuntimeError, match="not found"):
        _ffi.load_library(str(tmp_path / "librtx_amd.so"))


def _validate(sd):
    lib = _ffi.load_library()
    desc, keep = sd.to_desc()
    out = (C.c_uint64 * 3)()
    rc = lib.rtx_debug_validate_bvh(C.byref(desc), out)
    return rc, [int(x) for x in out], lib.rtx_last_error().decode()


def test_host_bvh_flattening_is_consistent():
    """Every triangle lands in exactly one leaf and every quantised child box contains its subtree."""
    sd = synth.build_scene(0.01)
    rc, out, err = _validate(sd)
    assert rc == 0, err
    assert out[0] == sd.n_triangles and out[1] > 0 and 3 * (out[2] + 2) <= 128
    # transformed instance + mesh without normals / texcoords
    V, N, UV, F = synth.torus(1.0, 0.3, 12, 6)
    s2 = SceneData()
    s2.materials = [Material()]
    s2.meshes = [Mesh(V, F, material=0), Mesh(V, F, vn=N, material=0, xform=[2, 0, 0, 1, 0, 1, 0, 0, 0, 0, 0.5, -3])]
    rc, out, err = _validate(s2)
    assert rc == 0 and out[0] == 2 * len(F), err
    # degenerate inputs: coincident triangles, a single triangle, an empty scene
    s3 = SceneData()
    s3.materials = [Material()]
    s3.meshes = [Mesh(np.tile(V[:3], (1, 1)), np.tile([[0, 1, 2]], (40, 1)), material=0)]
    assert _validate(s3)[0] == 0
    s3.meshes = [Mesh(V[:3], [[0, 1, 2]], material=0)]
    assert _validate(s3)[1][0] == 1
    assert _validate(SceneData())[0] == 0


def test_host_rejects_bad_input():
    V, N, UV, F = synth.torus(1.0, 0.3, 8, 4)
    sd = SceneData()
    sd.materials = [Material()]
    bad = F.copy()
    bad[3, 1] = len(V) + 5
    sd.meshes = [Mesh(V, bad, material=0)]
    rc, _, err = _validate(sd)
    assert rc == -1 and "index out of range" in err
    Vn = V.copy()
    Vn[7, 2] = np.nan
    sd.meshes = [Mesh(Vn, F, material=0)]
    rc, _, err = _validate(sd)
    assert rc == -1 and "non-finite" in err
    sd.meshes = [Mesh(V, F, material=3)]
    rc, _, err = _validate(sd)
    assert rc == -1 and "material" in err


@pytest.mark.parametrize("W,H,world", [(64, 36, 1), (70, 37, 2), (1920, 1080, 8), (15, 3, 4)])
def test_tile_sharding_partitions_the_frame(W, H, world):
    """Buckets are 16x16, row-major (RayCaster.cpp:1152-1158); bucket b goes to rank b % world; every pixel is owned
    by exactly one rank; a 64-entry wave block never straddles two buckets."""
    lib = _ffi.load_library()
    seen = np.zeros(W * H, np.int32)
    nbx = (W + 15) // 16
    for rank in range(world):
        n = lib.rtx_debug_pixel_list(W, H, 16, rank, world, None, 0)
        assert n % 64 == 0
        buf = np.zeros(max(n, 1), np.int32)
        assert lib.rtx_debug_pixel_list(W, H, 16, rank, world, buf.ctypes.data, n) == n
        buf = buf[:n]
        valid = buf[buf >= 0]
        np.add.at(seen, valid, 1)
        bucket = (valid // W // 16) * nbx + (valid % W) // 16
        assert (bucket % world == rank).all()
        for blk in buf.reshape(-1, 64):
            v = blk[blk >= 0]
            if len(v):
                assert len(np.unique((v // W // 16) * nbx + (v % W) // 16)) == 1
    assert (seen == 1).all()


@pytest.mark.parametrize("blocks", [1, 2, 63, 64, 65, 1000, 260000])
@pytest.mark.parametrize("mult", [1, 3])
def test_segmented_queue_layout(blocks, mult):
    """csrc/segqueue.h: a producer block appends to the segment its index selects; segment regions must be disjoint,
    ordered, inside the queue's capacity and large enough for everything the segment's blocks can append."""
    import ctypes as C
    lib = _ffi.load_library()
    start = (C.c_uint32 * 64)()
    seg = (C.c_uint32 * blocks)()
    assert lib.rtx_debug_queue_layout(blocks, 256, mult, start, seg) == 0
    start = np.array(start[:], np.int64)
    seg = np.array(seg[:], np.int64)
    assert (np.diff(seg) >= 0).all() and seg.min() >= 0 and seg.max() <= 63
    cap = blocks * 256 * mult
    assert (np.diff(start) >= 0).all() and start[0] == 0 and start[-1] <= cap
    ends = np.append(start[1:], cap)
    per_seg = np.bincount(seg, minlength=64) * 256 * mult     # worst case: nothing dropped
    assert (start + per_seg <= ends).all()
    # the first block of a segment writes at the segment start
    first = np.full(64, -1, np.int64)
    for b in range(blocks - 1, -1, -1):
        first[seg[b]] = b
    used = first >= 0
    assert (start[used] == first[used] * 256 * mult).all()


@pytest.mark.parametrize("blocks", [1, 3, 4, 5, 255, 256, 257, 1000, 260001])
@pytest.mark.parametrize("run", [1, 4, 16])
def test_interleaved_queue_layout(blocks, run):
    """csrc/segqueue.h, SegQ::run (the pair queue between shade_kernel and nee_kernel): producer blocks go round the 64
    segments in runs of `run` blocks. Whatever the grid, the segments' regions must tile the queue's capacity exactly - each
    as large as everything its blocks can append - and neighbouring runs must land in different segments."""
    import ctypes as C
    lib = _ffi.load_library()
    mult = 3
    start = (C.c_uint32 * 64)()
    seg = (C.c_uint32 * blocks)()
    assert lib.rtx_debug_queue_layout_run(blocks, 256, mult, run, start, seg) == 0
    start = np.array(start[:], np.int64)
    seg = np.array(seg[:], np.int64)
    assert seg.min() >= 0 and seg.max() <= 63
    assert (seg == (np.arange(blocks) // run) % 64).all()
    per_seg = np.bincount(seg, minlength=64) * 256 * mult
    assert start[0] == 0 and (start[1:] == np.cumsum(per_seg)[:-1]).all()      # dense: no gap, no overlap
    assert start[-1] + per_seg[-1] == blocks * 256 * mult
    # run = 0 is the contiguous layout of rtx_debug_queue_layout
    a = (C.c_uint32 * 64)()
    b = (C.c_uint32 * 64)()
    assert lib.rtx_debug_queue_layout_run(blocks, 256, mult, 0, a, None) == 0 and lib.rtx_debug_queue_layout(blocks, 256, mult, b, None) == 0
    assert a[:] == b[:]
