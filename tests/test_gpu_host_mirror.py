"""Runs the C++ counterparts of the reference's MODP example and tests (host mirror of the crate API over the
C ABI, GPU engine underneath).  BASELINE config C1: examples/mpvss_all."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "mpvss_rs_amd", "csrc"), "examples", "-s"])


def test_mpvss_all_example_prints_recovered_secret():
    _build()
    out = subprocess.run([os.path.join(ROOT, "examples", "mpvss_all"), "42"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    assert out.stdout.splitlines() == ["secret message: Hello MPVSS Example.", "r1 str: Hello MPVSS Example.",
                                       "r2 str: Hello MPVSS Example.", "r3 str: Hello MPVSS Example."]   # mpvss_all.rs:91-94


def test_reference_modp_tests_on_the_host_mirror():
    _build()
    out = subprocess.run([os.path.join(ROOT, "tests", "_build", "host_mirror_tests")], capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all passed" in out.stdout
