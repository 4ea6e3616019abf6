"""CPU proof (exhaustive) of the exact-arithmetic shortcuts used by the HIP kernels; see
tests/native/exact_arith_check.c.  The GPU side of the atan2 proof is sl3d_create's self-check."""
import os
import subprocess

from conftest import ROOT


def test_exact_division_and_atan2_lattice(tmp_path):
    src = os.path.join(ROOT, "tests", "native", "exact_arith_check.c")
    exe = str(tmp_path / "exact_arith_check")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", src, "-o", exe, "-lm"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.strip().endswith("OK")
