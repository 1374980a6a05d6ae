"""The claim behind the fused shrink pass's exponential (shrinkblur.hip FASTEXP, devsleef.h xexpf_v_ldexp): sleef's scaling by 2^q as five
multiplications by powers of two (the reference, the oracle, the three-kernel form) and as ONE correctly rounded ldexp return the same bits
for every argument whose result is a normal number or zero.  scripts/exp_ldexp_check.c walks every float (27 s on eight cores; its output is
quoted in DESIGN.md section 14.6); here every 7th bit pattern."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ldexp_scaling_differs_only_in_subnormal_results(tmp_path):
    exe = str(tmp_path / "exp_ldexp_check")
    subprocess.check_call(["gcc", "-O2", "-fopenmp", "-ffp-contract=off", "-msse2", os.path.join(ROOT, "scripts", "exp_ldexp_check.c"), "-lm", "-o", exe])
    out = subprocess.run([exe, "7"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    m = re.search(r"tested (\d+) arguments \(stride 7\): (\d+) differ, all in d = \[([-0-9.e+]+), ([-0-9.e+]+)\], largest result among them ([-0-9.e+]+) .*; (\d+) of them with a normal result", out.stdout)
    assert m, out.stdout
    tested, ndiff, dmin, dmax, rmax, nnormal = int(m.group(1)), int(m.group(2)), float(m.group(3)), float(m.group(4)), float(m.group(5)), int(m.group(6))
    assert tested > 3.7e8 and nnormal == 0
    assert ndiff > 1000 and -99.4 < dmin and dmax < -89.0 and rmax < 2.2e-39       # (the set exists, and lies where the kernel's argument says it does)
    o = re.search(r"result overflows and where the two forms differ: (\d+), the smallest ([-0-9.e+]+)", out.stdout)
    assert o and float(o.group(2)) > 398.0                                          # (beyond it ldexpk's factors leave the exponent field)
