"""CPU tests of the flattened tree through the host model of the kernels' traversal order (rtx_debug_host_walk; csrc/bvh_build.cpp
host_walk_closest): the tree loses no hit (against the oracle's brute-force loop of objects.cpp:257-264, ids and t bit for bit), and
the 8-wide layout kept behind -DRTX_BVH8=1 (DESIGN.md 3 "BVH8": measured not to pay, profiles/r03/bvh8_model.txt) builds, validates
and finds the same hits."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from raytracer_amd import _ffi, synth
from tests.helpers import random_rays, global_prim
from tests.oracle_ffi import OracleScene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _walk(fn, desc, rays):
    n = rays.shape[0]
    t = np.zeros(n, np.float32)
    prim = np.zeros(n, np.int32)
    cnt = (C.c_uint64 * 6)()
    rc = fn(C.byref(desc), C.c_void_p(rays.ctypes.data), C.c_size_t(n), C.c_void_p(t.ctypes.data), C.c_void_p(prim.ctypes.data), cnt)
    assert rc == 0, rc
    return t, prim, [int(x) for x in cnt]


@pytest.fixture(scope="module")
def small():
    sd = synth.build_scene(tri_scale=0.004, with_textures=False)
    desc, keep = sd.to_desc()
    osc = OracleScene(sd)
    rays = random_rays(np.random.default_rng(11), 6000, osc.bounds())
    return sd, desc, keep, osc, rays


def test_host_walk_finds_the_brute_force_hits(small):
    sd, desc, keep, osc, rays = small
    lib = _ffi.load_library()
    t, prim, cnt = _walk(lib.rtx_debug_host_walk, desc, rays)
    o = osc.trace_closest(rays, _ffi.HIT_FRONT_AND_BACK, use_bvh=False)
    face_base = np.concatenate([[0], np.cumsum([m.nf for m in sd.meshes])])[:-1]
    op = global_prim(o, face_base)
    assert (prim == op).all()
    hit = prim >= 0
    assert hit.sum() > 1000
    assert (t[hit] == o["t"][hit]).all()
    assert cnt[0] > 0 and cnt[1] > 0 and cnt[2] > 0


def test_eight_wide_layout_builds_and_agrees(small, tmp_path):
    """The Node8 builder (compiled out of the library: RTX_BVH8=0) through the host-only probe of tools/micro: the tree validates
    (every triangle once, quantised boxes conservative, transmissive masks right) and the group-order walk finds the same hits."""
    sd, desc, keep, osc, rays = small
    so = str(tmp_path / "libbvhprobe8.so")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-DRTX_BVH8=1", "-I", os.path.join(ROOT, "raytracer_amd", "csrc"),
                           os.path.join(ROOT, "tools", "micro", "bvh_probe.cpp"), os.path.join(ROOT, "raytracer_amd", "csrc", "bvh_build.cpp"),
                           "-o", so, "-pthread"])
    probe = C.CDLL(so)
    t8, prim8, cnt8 = _walk(probe.bvh_probe, desc, rays)
    lib = _ffi.load_library()
    t4, prim4, cnt4 = _walk(lib.rtx_debug_host_walk, desc, rays)
    assert (prim8 == prim4).all() and (t8 == t4).all()
    assert cnt8[4] == 128 and cnt8[2] < cnt4[2]   # 128-byte nodes, fewer of them
