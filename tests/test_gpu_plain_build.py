"""The product library is built with LLVM-internal options per source file (bcftools_amd/csrc/Makefile: scheduling strategy, no loop
strength reduction, uniform regions left alone) that another toolchain may refuse -- the Makefile then leaves them out.  They reorder
instructions, not arithmetic (-ffp-contract=off stays): this test runs the same seeded inputs through both builds -- libbcfgpu.so
and libbcfgpu_plain.so (`make plain`) -- in a child process each and compares a digest of every output byte, stage by stage."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _digest(so):
    env = dict(os.environ)
    if so:
        env["BCFGPU_SO"] = so
    r = subprocess.run([sys.executable, "-m", "tests.helpers.digest"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stderr or r.stdout)[-2000:]
    d = dict(ln.split() for ln in r.stdout.strip().splitlines() if len(ln.split()) == 2)
    assert set(d) == {"snp", "indel", "baq"}, r.stdout
    return d


def test_plain_build_gives_the_same_bytes():
    plain = os.path.join(ROOT, "bcftools_amd", "libbcfgpu_plain.so")
    assert os.path.exists(plain), "run __graft_entry__.build() (make plain)"
    assert _digest(None) == _digest(plain)
