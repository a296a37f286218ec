"""Writes tests/golden/xorwow_constants.json: the constants the XORWOW restatement assumes (SURVEY.md 8c asks
the fixtures to record them) and the first 16 curand() outputs for seeds 0, 1, 0x5EED as the oracle produces them.

    python tests/golden/gen_xorwow_golden.py

Provenance of each constant is recorded in the file: the xorshift/Weyl base state and increment are confirmed by a
third party present in this image (rocRAND's xorwow_engine, /opt/rocm/include/rocrand/rocrand_xorwow.h:104-110,165-175,
which uses the same values); the four curand_init scramble constants are restated from memory of CUDA 11.7's
curand_kernel.h and cannot be verified here (no CUDA toolkit, no CUDA device)."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import oracle  # noqa: E402

out = {
    "generator": "XORWOW (Marsaglia 2003, 'Xorshift RNGs' section 3.1) as cuRAND's curandStateXORWOW",
    "base_state": {"v": [123456789, 362436069, 521288629, 88675123, 5783321], "d": 6615241,
                   "provenance": "rocrand_xorwow.h:104-110 holds the same six values (third party, in this image)"},
    "weyl_increment": {"value": 362437, "provenance": "rocrand_xorwow.h:173 (third party, in this image)"},
    "recurrence": "t = v0 ^ (v0 >> 2); v0..v3 = v1..v4; v4 = (v4 ^ (v4 << 4)) ^ (t ^ (t << 1)); d += 362437; return v4 + d",
    "curand_init_scramble": {"salt_lo": 0xaad26b49, "salt_hi": 0xf7dcefdd, "mul_lo": 1099087573, "mul_hi": 2591861531,
                             "structure": "s0 = lo32(seed) ^ salt_lo; s1 = hi32(seed) ^ salt_hi; t0 = mul_lo * s0; "
                                          "t1 = mul_hi * s1; v0 += t0; v1 ^= t0; v2 += t1; v3 ^= t1; v4 += t0; d += t1 + t0",
                             "provenance": "structure confirmed by rocrand_xorwow.h:112-122 (same statements, other "
                                           "constants); the four constants are from memory of CUDA 11.7 curand_kernel.h "
                                           "-- UNVERIFIED"},
    "rocrand_init_scramble": {"salt_lo": 0x2c7f967f, "salt_hi": 0xa03697cb, "mul_lo": 1228688033, "mul_hi": 2073658381},
    "uniform_f32": "curand_uniform: float(x) * 2^-32 + 2^-33 (one rounding after the cvt.rn of x)",
    "uniform_f64": "curand_uniform_double (XORWOW): z = x ^ (y << 21) over two draws; z * 2^-53 + 2^-54",
    "first_outputs": {},
    "first_states": {},
}
for seed in (0, 1, 0x5EED):
    out["first_outputs"][str(seed)] = [int(x) for x in oracle.xorwow_stream(seed, 16)]
    st = oracle.random_states(1, seed)
    out["first_states"][str(seed)] = {"d": int(st["d"][0]), "v": [int(x) for x in st["v"][0]]}
json.dump(out, open(os.path.join(HERE, "xorwow_constants.json"), "w"), indent=1)
print("wrote xorwow_constants.json")
