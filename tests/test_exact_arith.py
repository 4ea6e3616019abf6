"""CPU proof (exhaustive) of the exact-arithmetic shortcuts used by the HIP kernels; see
tests/native/exact_arith_check.c.  The GPU side of the atan2 proof is sl3d_create's self-check."""
import os
import subprocess

from conftest import ROOT


def test_exact_division_and_atan2_lattice(tmp_path):
    src = os.path.join(ROOT, "tests", "native", "exact_arith_check.c")
    exe = str(tmp_path / "exact_arith_check")
    inc = "-I" + os.path.join(ROOT, "3dscan_amd", "csrc")  # sl3d_atan_coeffs.h: the constants the kernels are built with
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", inc, src, "-o", exe, "-lm"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.strip().endswith("OK")


def test_atan2_lattice_other_degrees(tmp_path):
    """The header documents which polynomial degrees are proven: check them all (the default one is covered above)."""
    src = os.path.join(ROOT, "tests", "native", "exact_arith_check.c")
    inc = "-I" + os.path.join(ROOT, "3dscan_amd", "csrc")
    for deg in (6, 7, 10):
        exe = str(tmp_path / f"atan_deg{deg}")
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-DATAN_ONLY", f"-DSL3D_ATAN_DEG={deg}", inc, src, "-o", exe, "-lm"])
        out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and out.stdout.strip().endswith("OK"), out.stdout + out.stderr
