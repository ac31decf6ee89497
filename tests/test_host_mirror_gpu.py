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
