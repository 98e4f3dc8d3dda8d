"""rz_device_math.h: rcp_mid() -- v_rcp_f32 plus one Newton step -- stands in for the 11-instruction IEEE division
1.0f / x wherever rcp_mid_ok(x) holds.  That is only legitimate if it is the correctly rounded quotient for EVERY such
x on the GPU at hand: the sweep below tries all 2^32 bit patterns against the compiler's full division (about a second)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu


def test_short_reciprocal_is_the_ieee_quotient_for_every_admitted_input(tmp_path):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "rcp_exhaustive")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-I",
                           os.path.join(root, "rayzen_amd", "csrc", "hip"), "-I", os.path.join(root, "include"), "-o", exe,
                           os.path.join(root, "profiles", "scripts", "rcp_exhaustive.hip")], stderr=subprocess.DEVNULL)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    line = [l for l in out.stdout.splitlines() if l.startswith("PRODUCT rcp_mid")]
    assert line, out.stdout + out.stderr
    # every x with 2^-126 <= |x| <= 2^126: exponents 1 .. 253 with any mantissa, minus the patterns above 2^126 exactly
    assert line[0].endswith("mismatches 0"), out.stdout
    assert out.returncode == 0, out.stdout
