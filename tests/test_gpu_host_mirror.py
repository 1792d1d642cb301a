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


@pytest.mark.parametrize("binary,message", [("mpvss_all_secp256k1", "Hello MPVSS Example (secp256k1)."),
                                            ("mpvss_all_ristretto255", "Hello MPVSS Example (Ristretto255).")])
def test_curve_group_examples_print_recovered_secret(binary, message):
    """C++ counterparts of examples/mpvss_all_secp256k1.rs and examples/mpvss_all_ristretto255.rs (the reference's only
    end-to-end coverage of the ristretto255 Participant): distribute, verify, extract, verify_share, reconstruct."""
    _build()
    for seed in ("7", "1234567"):
        out = subprocess.run([os.path.join(ROOT, "examples", binary), seed], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr
        assert out.stdout.splitlines() == [f"secret message: {message}"] + [f"r{k} str: {message}" for k in (1, 2, 3)]


def test_compiled_callers_one_box_per_call_on_one_context():
    """examples/drop_in_threads.cpp: six std::threads each calling mpvss_modp_verify_distribution -- ONE box per call, the reference's
    call shape (participant.rs:399-455) -- on one context over five dealers' boxes: every call must give verdict 1 and its dealer's
    digest (the program exits non-zero otherwise), a tampered box is rejected.  The compiled twin of bench.py's `drop_in` leg."""
    _build()
    out = subprocess.run([os.path.join(ROOT, "examples", "drop_in_threads"), "4200", "16", "5", "6", "2"], capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.splitlines()[-1] == "ok" and "6 compiled callers" in out.stdout
