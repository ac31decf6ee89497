"""Runs the C++ host-mirror parity program (tests/host/host_parity.cpp): the reference's entry points RayCaster,
IntersectNode, IntersectSceneWithLights, CollectSceneNodeObjects, ShadeInfoAdapter::TraceShadowRay and
PhotonMapBuilder, re-hosted over the C ABI, checked against the oracle."""
import os
import subprocess

import pytest

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host")


@pytest.mark.gpu
def test_host_mirror_parity(rtx):
    exe = os.path.join(HERE, "host_parity")
    assert os.path.exists(exe), "tests/host/host_parity is built by __graft_entry__.build()"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(r.stdout[-3000:], r.stderr[-2000:])
    assert r.returncode == 0, r.stdout[-3000:]


def test_host_mirror_compiles():
    """CPU check: the mirror headers and the test program build against include/rtx.h and link the C ABI."""
    subprocess.check_call(["make", "-C", HERE])
    assert os.path.exists(os.path.join(HERE, "host_parity"))


def test_png_writer_roundtrip(tmp_path):
    """raytracer_amd/host/ImageIO.h: files written by the C++ host mirror decode with zlib to the same pixels."""
    import struct
    import zlib
    import numpy as np
    src = tmp_path / "png_test.cpp"
    exe = tmp_path / "png_test"
    src.write_text('''
#include "%s/raytracer_amd/host/ImageIO.h"
int main(int argc, char** argv) {
    const int W = 301, H = 233;   // > 65535 bytes of scanlines: several stored blocks
    std::vector<uint8_t> rgb(W * H * 3), grey(W * H);
    for (int i = 0; i < W * H; ++i) { rgb[3*i] = i * 7; rgb[3*i+1] = i / 3; rgb[3*i+2] = i ^ (i >> 5); grey[i] = i * 13; }
    return rtamd::png::save(argv[1], rgb.data(), W, H, 3) && rtamd::png::save(argv[2], grey.data(), W, H, 1) ? 0 : 1;
}
''' % os.path.dirname(os.path.dirname(HERE)))
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", str(exe), str(src)])
    a, b = tmp_path / "rgb.png", tmp_path / "grey.png"
    subprocess.check_call([str(exe), str(a), str(b)])

    def decode(path, channels):
        data = open(path, "rb").read()
        assert data[:8] == b"\x89PNG\r\n\x1a\n"
        pos, idat, hdr = 8, b"", None
        while pos < len(data):
            n, typ = struct.unpack(">I4s", data[pos:pos + 8])
            body = data[pos + 8:pos + 8 + n]
            crc, = struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])
            assert zlib.crc32(typ + body) == crc
            if typ == b"IHDR":
                hdr = struct.unpack(">IIBBBBB", body)
            if typ == b"IDAT":
                idat += body
            pos += 12 + n
        W, H, depth, ctype = hdr[:4]
        assert depth == 8 and ctype == (2 if channels == 3 else 0)
        raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(H, 1 + W * channels)
        assert (raw[:, 0] == 0).all()
        return raw[:, 1:].reshape(H, W, channels)

    i = np.arange(301 * 233, dtype=np.int64)
    rgb = decode(a, 3).reshape(-1, 3)
    assert np.array_equal(rgb[:, 0], (i * 7) % 256) and np.array_equal(rgb[:, 1], (i // 3) % 256) and np.array_equal(rgb[:, 2], (i ^ (i >> 5)) % 256)
    assert np.array_equal(decode(b, 1).reshape(-1), (i * 13) % 256)
