"""Host check of the reciprocal + fma-residual divisions of the quad search kernel
(muzero_amd/csrc/mzs_quad.h div_r1 / div_r2): they must equal correctly rounded division,
because the oracle (and the reference) divide for real."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_corrected_reciprocal_division_is_exact(tmp_path):
    exe = tmp_path / "division_check"
    subprocess.check_call(["gcc", "-O2", "-mfma", "-fopenmp", "-ffp-contract=off", "-o", str(exe),
                           os.path.join(HERE, "csrc", "division_check.c"), "-lm"])
    out = subprocess.run([str(exe), "20000000"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.split() == ["mismatches", "0", "0", "0"]
