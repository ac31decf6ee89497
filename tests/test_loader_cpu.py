"""Scene ingest of the C++ host mirror (raytracer_amd/host/SceneLoader.h): the XML dialect of
assets/cerealBar.xml and Wavefront OBJ meshes. Host code only (g++), no GPU."""
import json
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_XML = "/root/reference/assets/cerealBar.xml"   # present in the build container only


@pytest.fixture(scope="module")
def loader(tmp_path_factory):
    exe = tmp_path_factory.mktemp("loader") / "loader_test"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", str(exe), os.path.join(ROOT, "tests", "host", "loader_test.cpp")])

    def run(path, *args):
        r = subprocess.run([str(exe), str(path), *args], capture_output=True, text=True)
        return r.returncode, json.loads(r.stdout)
    return run


SCENE = """<xml>
  <scene>
    <!-- a comment with <object> inside -->
    <object type="obj" name="quad.obj" material="checker">
      <scale x="2" y="1" z="1"/>
      <rotate angle="90" z="1"/>
      <translate x="1" y="2" z="3"/>
      <object type="sphere" name="ball" material="glass">
        <scale value="0.5"/>
        <translate x="0" y="0" z="1"/>
      </object>
    </object>
    <object type="plane" name="floor" material="checker"/>
    <material type="blinn" name="glass">
      <diffuse value="0"/>
      <specular value="0"/>
      <glossiness value="100000"/>
      <refraction value="1" index="1.5"/>
      <absorption r="0.75" g="0.15" b="0.15"/>
    </material>
    <material type="phong" name="checker">
      <diffuse r="1" g="0.5" b="0.25" texture="tex.ppm" value="0.5">
        <scale x="1" y="-1" z="1"/>
      </diffuse>
      <specular r="1" g="1" b="1" value="0.3"/>
      <bump texture="tex.ppm" value="2"/>
    </material>
    <environment texture="env.pfm" r="0.5" g="0.6" b="0.7" value="5">
      <rotate angle="180" x="1"/>
      <scale value="30"/>
    </environment>
    <light type="point" name="l0">
      <intensity r="1" g="0.549925" b="0.235" value="5"/>
      <position x="1.9" y="6" z="2.9"/>
      <size value="1"/>
      <attenuation value="1"/>
    </light>
    <light type="ambient" name="amb"><intensity value="0.1"/></light>
  </scene>
  <camera gamma="sRGB">
    <position x="0" y="0" z="5"/> <target x="0" y="0" z="0"/> <up x="0" y="1" z="0"/>
    <fov value="24"/> <focaldist value="3.5"/> <dof value="0.02"/> <width value="320"/> <height value="180"/>
  </camera>
</xml>
"""

OBJ = """# a quad as one polygon with v/vt/vn, and a triangle with negative indices in another material
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
vt 0 0
vt 1 0
vt 1 1
vt 0 1
vn 0 0 1
usemtl first
f 1/1/1 2/2/1 3/3/1 4/4/1
usemtl second
f -4/-4/-1 -3/-3/-1 -1/-1/-1
"""


def test_xml_and_obj_subset(loader, tmp_path):
    (tmp_path / "scene.xml").write_text(SCENE)
    (tmp_path / "quad.obj").write_text(OBJ)
    tex = np.arange(4 * 2 * 3, dtype=np.uint8).reshape(2, 4, 3)
    (tmp_path / "tex.ppm").write_bytes(b"P6\n# c\n4 2\n255\n" + tex.tobytes())
    env = np.linspace(0, 3, 8 * 4 * 3, dtype="<f4").reshape(4, 8, 3)
    (tmp_path / "env.pfm").write_bytes(b"PF\n8 4\n-1.0\n" + env[::-1].tobytes())
    rc, s = loader(tmp_path / "scene.xml")
    assert rc == 0, s
    kinds = [o["kind"] for o in s["objects"]]
    assert kinds == ["mesh", "sphere", "plane"]                      # DFS pre-order
    quad = s["objects"][0]
    assert (quad["nf"], quad["nv"], quad["nvt"], quad["nvn"]) == (3, 4, 4, 1)
    assert quad["faces"] == [[0, 1, 2], [0, 2, 3], [0, 1, 3]] and quad["face_mtl"] == [0, 0, 1]
    # scale, then rotate 90 deg about z, then translate: x' = -y + 1, y' = 2x + 2, z' = z + 3
    assert np.allclose(np.array(quad["world"]).reshape(3, 4), [[0, -1, 0, 1], [2, 0, 0, 2], [0, 0, 1, 3]], atol=1e-6)
    ball = np.array(s["objects"][1]["world"]).reshape(3, 4)           # parent * (translate * scale 0.5)
    assert np.allclose(ball, [[0, -0.5, 0, 1], [1, 0, 0, 2], [0, 0, 0.5, 4]], atol=1e-6)
    assert s["objects"][1]["material"] == "glass" and s["objects"][2]["material"] == "checker"
    g, c = s["materials"]["glass"], s["materials"]["checker"]
    assert g["type"] == 0 and g["diffuse"] == [0, 0, 0] and g["refraction"] == [1, 1, 1] and g["ior"] == 1.5
    assert np.allclose(g["absorption"], [0.75, 0.15, 0.15]) and g["glossiness"] == 100000
    assert c["type"] == 1 and np.allclose(c["diffuse"], [0.5, 0.25, 0.125]) and np.allclose(c["specular"], [0.3] * 3)
    assert c["bump"] and c["bump_value"] == 2 and c["diffuse_tex"] and c["diffuse_tex_size"] == [4, 2]
    assert np.allclose(np.array(c["diffuse_tex_itm"]).reshape(3, 4), [[1, 0, 0, 0], [0, -1, 0, 0], [0, 0, 1, 0]])
    assert len(s["lights"]) == 1 and np.allclose(s["lights"][0]["intensity"], [5, 2.749625, 1.175]) and s["lights"][0]["size"] == 1
    e = s["environment"]
    assert np.allclose(e["value"], [2.5, 3.0, 3.5]) and e["tex"]
    assert np.allclose(np.array(e["itm"]).reshape(3, 4), np.diag([1 / 30, -1 / 30, -1 / 30]) @ np.eye(3, 4), atol=1e-7)
    cam = s["camera"]
    assert cam["pos"] == [0, 0, 5] and np.allclose(cam["dir"], [0, 0, -1]) and np.allclose(cam["up"], [0, 1, 0])
    assert (cam["fov"], cam["focaldist"], cam["width"], cam["height"], cam["srgb"]) == (24, 3.5, 320, 180, True)
    assert np.isclose(cam["dof"], 0.02)
    assert s["warnings"] == 1                                         # the ambient light


def test_errors_are_reported(loader, tmp_path):
    (tmp_path / "bad.xml").write_text("<xml><scene><object type='obj' name='nope.obj'></scene></xml>")
    rc, s = loader(tmp_path / "bad.xml")
    assert rc == 1 and "mismatched" in s["error"]
    (tmp_path / "missing.xml").write_text("<xml><scene><object type='obj' name='nope.obj'/></scene></xml>")
    rc, s = loader(tmp_path / "missing.xml")
    assert rc == 1 and "nope.obj" in s["error"]
    rc, s = loader(tmp_path / "missing.xml", "--allow-missing")
    assert rc == 0 and s["objects"] == [] and s["warnings"] == 1


@pytest.mark.skipif(not os.path.exists(REF_XML), reason="the reference's scene file exists in the build container only")
def test_reference_scene_file(loader):
    """assets/cerealBar.xml itself (its OBJ files are git-lfs pointers, so meshes are skipped): materials, lights,
    environment and camera as SURVEY.md 8a rows B/L/X list them and as raytracer_amd/synth.py transcribes them."""
    from raytracer_amd import synth
    rc, s = loader(REF_XML, "--allow-missing")
    assert rc == 0, s
    assert len(s["materials"]) == 17 and len(s["lights"]) == 3
    glass = s["materials"]["glass"]
    assert glass["refraction"] == [1, 1, 1] and glass["ior"] == 1.5 and glass["glossiness"] == 100000
    assert np.allclose(glass["absorption"], [0.75, 0.15, 0.15])
    assert np.allclose(s["materials"]["porcelain"]["reflection"], [0.3] * 3)
    assert s["materials"]["fruitLoopsRed"]["bump_value"] == 2
    for L in s["lights"]:
        assert np.allclose(L["intensity"], [5, 2.749625, 1.175]) and L["size"] == 1 and L["attenuation"] == 1
    assert np.allclose(s["environment"]["value"], [2.5, 3.0, 3.5])
    cam = s["camera"]
    assert np.allclose(cam["pos"], [-1.6766, 0.6769, 4.6929]) and cam["fov"] == 24 and cam["focaldist"] == 3.5
    assert (cam["width"], cam["height"], cam["srgb"]) == (1920, 1080, True)
    # the synthetic stand-in scene uses the same material table
    by_name = {m.name: m for m in synth._materials()}
    for name, m in s["materials"].items():
        assert name in by_name, name
        assert np.allclose(m["diffuse"], by_name[name].diffuse, atol=1e-6), name
        assert np.isclose(m["glossiness"], by_name[name].glossiness), name


def test_camera_keyframes(loader, tmp_path):
    """<keyframe time> children of <camera> (assets/cerealBar.xml:220-271 holds them commented out; same syntax here)
    and Camera::EvaluateAtTime as the frame loop of Main.cpp:381-401 calls it: piecewise linear in position, target and
    fov, constant outside the keyed range, frame rebuilt from the <up> hint."""
    (tmp_path / "anim.xml").write_text("""<xml><scene/>
      <camera>
        <position x="0" y="0" z="5"/> <target x="0" y="0" z="0"/> <up x="0" y="1" z="0"/> <fov value="30"/>
        <keyframe time="2.0"> <position x="4" y="0" z="0"/> <target x="0" y="0" z="0"/> <fov value="50"/> </keyframe>
        <keyframe time="0.0"> <position x="0" y="0" z="4"/> <target x="0" y="0" z="0"/> <fov value="30.00"/> </keyframe>
      </camera></xml>""")
    rc, s = loader(tmp_path / "anim.xml", "--at", "-1", "0", "1", "2", "9")
    assert rc == 0, s
    kf = s["camera"]["keyframes"]
    assert [k["time"] for k in kf] == [0.0, 2.0] and kf[1]["pos"] == [4, 0, 0] and kf[1]["fov"] == 50      # sorted by time
    ev = s["camera"]["evaluated"]
    assert ev[0]["pos"] == ev[1]["pos"] == [0, 0, 4] and ev[3]["pos"] == ev[4]["pos"] == [4, 0, 0]
    mid = ev[2]
    assert np.allclose(mid["pos"], [2, 0, 2]) and mid["fov"] == pytest.approx(40.0)
    assert np.allclose(mid["dir"], np.array([-1, 0, -1]) / np.sqrt(2), atol=1e-6)
    assert np.allclose(mid["up"], [0, 1, 0], atol=1e-6)
    assert s["camera"]["pos"] == [0, 0, 5]                        # the static camera is untouched by loading keyframes


@pytest.fixture(scope="module")
def image_decoder(tmp_path_factory):
    d = tmp_path_factory.mktemp("imgdec")
    exe = d / "image_test"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", str(exe), os.path.join(ROOT, "tests", "host", "image_test.cpp")])

    def run(path):
        raw = d / "out.raw"
        r = subprocess.run([str(exe), str(path), str(raw)], capture_output=True, text=True)
        info = json.loads(r.stdout)
        if r.returncode != 0:
            return info, None
        a = np.fromfile(raw, np.float32 if info["float"] else np.uint8).reshape(info["h"], info["w"], 3)
        return info, a
    return run


def _png_bytes(rows, w, h, depth, ctype, filters, level=9, strategy=0, palette=None):
    """A PNG file from raw scanline bytes, with the given filter type per row (PNG specification, section 9)."""
    import struct
    import zlib
    bpp = max(1, {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype] * depth // 8)
    out = bytearray()
    prev = np.zeros(len(rows[0]), np.int32)
    for y, row in enumerate(rows):
        cur = np.frombuffer(bytes(row), np.uint8).astype(np.int32)
        a = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]])
        c = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]])
        ft = filters[y % len(filters)]
        if ft == 0: pred = 0
        elif ft == 1: pred = a
        elif ft == 2: pred = prev
        elif ft == 3: pred = (a + prev) // 2
        else:
            p = a + prev - c
            pa, pb, pc = abs(p - a), abs(p - prev), abs(p - c)
            pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, prev, c))
        out.append(ft)
        out += ((cur - pred) & 255).astype(np.uint8).tobytes()
        prev = cur
    co = zlib.compressobj(level, zlib.DEFLATED, 15, 9, strategy)
    z = co.compress(bytes(out)) + co.flush()

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    f = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0))
    if palette is not None:
        f += chunk(b"PLTE", bytes(palette))
    half = len(z) // 2                                   # two IDAT chunks: the stream continues across them
    return f + chunk(b"IDAT", z[:half]) + chunk(b"IDAT", z[half:]) + chunk(b"IEND", b"")


def test_png_decoder(image_decoder, tmp_path):
    """host/ImageDecode.h LoadPng: every colour type / bit depth the decoder accepts, the five scanline filters, stored /
    fixed / dynamic Huffman blocks, split IDAT chunks."""
    import zlib
    rng = np.random.default_rng(3)
    w, h = 37, 23
    smooth = (np.add.outer(np.arange(h) * 5, np.arange(w) * 3)[..., None] + np.array([0, 40, 90])) % 256
    img = np.where(rng.random((h, w, 1)) < 0.3, rng.integers(0, 256, (h, w, 3)), smooth).astype(np.uint8)
    cases = [("rgb8", 8, 2, [row.tobytes() for row in img], img)]
    rgba = np.concatenate([img, rng.integers(0, 256, (h, w, 1), dtype=np.uint8)], 2)
    cases.append(("rgba8", 8, 6, [row.tobytes() for row in rgba], img))
    grey = img[..., :1]
    cases.append(("grey8", 8, 0, [row.tobytes() for row in grey], np.repeat(grey, 3, 2)))
    ga = np.concatenate([grey, rgba[..., 3:]], 2)
    cases.append(("greyalpha8", 8, 4, [row.tobytes() for row in ga], np.repeat(grey, 3, 2)))
    hi = img.astype(np.uint16) * 256 + rng.integers(0, 256, img.shape).astype(np.uint16)
    cases.append(("rgb16", 16, 2, [row.astype(">u2").tobytes() for row in hi], img))
    for name, depth, ctype, rows, expect in cases:
        for level, strategy in ((9, 0), (0, 0), (6, zlib.Z_FIXED)):
            p = tmp_path / f"{name}_{level}_{strategy}.png"
            p.write_bytes(_png_bytes(rows, w, h, depth, ctype, [0, 1, 2, 3, 4], level, strategy))
            info, a = image_decoder(p)
            assert a is not None and (info["w"], info["h"], info["float"]) == (w, h, False), (name, info)
            assert np.array_equal(a, expect), (name, level, strategy)
    # palette, 8 and 4 bits per index; 1-bit grey
    pal = rng.integers(0, 256, (16, 3), dtype=np.uint8)
    idx = rng.integers(0, 16, (h, w), dtype=np.uint8)
    p = tmp_path / "pal8.png"
    p.write_bytes(_png_bytes([row.tobytes() for row in idx], w, h, 8, 3, [0, 2], palette=pal.tobytes()))
    assert np.array_equal(image_decoder(p)[1], pal[idx])
    packed = [np.packbits(np.unpackbits(row[:, None], axis=1)[:, 4:].ravel()).tobytes() for row in idx]
    p = tmp_path / "pal4.png"
    p.write_bytes(_png_bytes(packed, w, h, 4, 3, [0, 1], palette=pal.tobytes()))
    assert np.array_equal(image_decoder(p)[1], pal[idx])
    bits = rng.integers(0, 2, (h, w), dtype=np.uint8)
    p = tmp_path / "grey1.png"
    p.write_bytes(_png_bytes([np.packbits(row).tobytes() for row in bits], w, h, 1, 0, [0]))
    assert np.array_equal(image_decoder(p)[1], np.repeat((bits * 255)[..., None], 3, 2))
    # not a PNG, truncated stream: refused, not crashed
    (tmp_path / "bad.png").write_bytes(b"\x89PNG\r\n\x1a\n" + b"\0" * 40)
    assert image_decoder(tmp_path / "bad.png")[1] is None
    good = _png_bytes(cases[0][3], w, h, 8, 2, [4])
    (tmp_path / "cut.png").write_bytes(good[:len(good) // 2])
    assert image_decoder(tmp_path / "cut.png")[1] is None


def _rgbe_to_float(rgbe):
    e = rgbe[..., 3].astype(np.int32)
    f = np.ldexp(np.float32(1.0), e - 136).astype(np.float32)
    out = rgbe[..., :3].astype(np.float32) * f[..., None]
    out[e == 0] = 0
    return out


def test_radiance_hdr_decoder(image_decoder, tmp_path):
    """host/ImageDecode.h LoadHdr: flat and run-length encoded scanlines against a numpy RGBE conversion."""
    rng = np.random.default_rng(9)
    w, h = 40, 9
    rgbe = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    rgbe[..., 3] = rng.integers(100, 150, (h, w))
    rgbe[2, 5:30] = rgbe[2, 5]                               # a long run
    rgbe[4, :, 3] = 0                                          # zero exponent = black
    header = b"#?RADIANCE\n# made by a test\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (h, w)
    (tmp_path / "flat.hdr").write_bytes(header + rgbe.tobytes())
    info, a = image_decoder(tmp_path / "flat.hdr")
    assert (info["w"], info["h"], info["float"]) == (w, h, True) and np.array_equal(a, _rgbe_to_float(rgbe))

    def rle(row):                                              # new-style RLE of one scanline (runs >= 3 as runs)
        out = bytearray([2, 2, w >> 8, w & 255])
        for ch in range(4):
            v = row[:, ch]
            x = 0
            while x < w:
                r = 1
                while x + r < w and r < 127 and v[x + r] == v[x]:
                    r += 1
                if r >= 3:
                    out += bytes([128 + r, v[x]])
                    x += r
                else:
                    s = x
                    while x < w and x - s < 128:
                        r = 1
                        while x + r < w and r < 3 and v[x + r] == v[x]:
                            r += 1
                        if r >= 3:
                            break
                        x += 1
                    out += bytes([x - s]) + v[s:x].tobytes()
        return bytes(out)
    (tmp_path / "rle.hdr").write_bytes(header + b"".join(rle(row) for row in rgbe))
    info, a = image_decoder(tmp_path / "rle.hdr")
    assert a is not None and np.array_equal(a, _rgbe_to_float(rgbe))
    (tmp_path / "cut.hdr").write_bytes((header + rgbe.tobytes())[:200])
    assert image_decoder(tmp_path / "cut.hdr")[1] is None


@pytest.mark.skipif(not os.path.exists(REF_XML), reason="the reference's texture files exist in the build container only")
def test_reference_texture_files(image_decoder):
    """The two textures the reference ships (assets/textures): bgPlate.png against PIL's decoder when PIL is there,
    alps_field_1k.hdr (the environment of cerealBar.xml:180) by size and range."""
    tex = os.path.join(os.path.dirname(REF_XML), "textures")
    info, a = image_decoder(os.path.join(tex, "bgPlate.png"))
    assert a is not None and not info["float"] and a.std() > 1
    try:
        from PIL import Image
        ref = np.asarray(Image.open(os.path.join(tex, "bgPlate.png")).convert("RGB"))
        assert np.array_equal(a, ref)
    except ImportError:
        pass
    info, e = image_decoder(os.path.join(tex, "alps_field_1k.hdr"))
    assert (info["w"], info["h"], info["float"]) == (1024, 512, True)
    assert np.isfinite(e).all() and e.min() >= 0 and e.max() > 100 and 0.1 < e.mean() < 10     # a sunlit sky
