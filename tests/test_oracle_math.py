"""Pins the oracle's self-written cosf/sinf against this machine's libm (the function the
reference's `cos(float)`/`sin(float)` at ORBextractor.cpp:53 binds to) and checks fastAtan2."""
import os
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r"""
#include "orc_math.h"
#include <cstdio>
#include <cstdlib>
int main(int argc, char** argv) {
    unsigned step = (unsigned)atoi(argv[1]);
    float lim = 6.2831860f; uint32_t u1; memcpy(&u1, &lim, 4);
    long bad_c = 0, bad_s = 0, tot = 0;
    for (uint32_t u = 0; u <= u1; u += step) {
        float f; memcpy(&f, &u, 4); tot++;
        if (orc::orc_cosf(f) != cosf(f)) bad_c++;
        if (orc::orc_sinf(f) != sinf(f)) bad_s++;
    }
    double worst = 0;
    for (int i = -2000; i <= 2000; i++) for (int j = -2000; j <= 2000; j += 7) {
        if (!i && !j) continue;
        double ref = atan2((double)i, (double)j) * 180.0 / 3.14159265358979323846; if (ref < 0) ref += 360.0;
        double got = orc::fast_atan2((float)i, (float)j);
        double d = fabs(got - ref); if (d > 180) d = 360 - d; if (d > worst) worst = d;
    }
    printf("%ld %ld %ld %.6f\n", tot, bad_c, bad_s, worst);
    return 0;
}
"""


def _run(step):
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "m.cpp")
        exe = os.path.join(td, "m")
        with open(src, "w") as f:
            f.write(SRC)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "oracle"),
                               "-o", exe, src, "-lm"])
        tot, bad_c, bad_s, worst = subprocess.check_output([exe, str(step)]).split()
        return int(tot), int(bad_c), int(bad_s), float(worst)


def test_sincos_bit_exact_vs_libm_sampled():
    tot, bad_c, bad_s, worst = _run(127)
    assert tot > 8_000_000
    assert bad_c == 0 and bad_s == 0
    assert worst < 0.02  # fastAtan2 polynomial: ~0.01 degree


@pytest.mark.slow
def test_sincos_bit_exact_vs_libm_exhaustive():
    tot, bad_c, bad_s, _ = _run(1)
    assert bad_c == 0 and bad_s == 0
