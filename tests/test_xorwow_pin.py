"""Pins of the XORWOW restatement (oracle/ggms_oracle.c:22-63, xgnn_amd/csrc/ggms_device.h Xorwow) that this image allows.

cuRAND itself is absent (no CUDA toolkit), so:
  * the RECURRENCE is checked against a third-party implementation that is present: rocRAND's host-callable
    rocrand_device::xorwow_engine::next() (/opt/rocm/include/rocrand/rocrand_xorwow.h:165-175), loaded with identical
    state through oracle/rocrand_pin.cc -- 4M draws from several states, word for word;
  * the STRUCTURE of curand_init(seed, 0, 0) is checked against rocRAND's constructor (:104-122): the same statements
    with rocRAND's salts/multipliers reproduce rocRAND's state, with cuRAND's they reproduce the oracle's.  The four
    cuRAND constants themselves stay "from memory" (tests/golden/xorwow_constants.json records them);
  * curand_uniform / curand_uniform_double are checked against the closed forms of SURVEY.md 8c in exact arithmetic.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "xorwow_constants.json")))
M32 = 0xFFFFFFFF


@pytest.fixture(scope="module")
def pin():
    p = oracle.rocrand_pin()
    if p is None:
        pytest.skip("rocRAND headers not available: cannot build oracle/librocrand_pin.so")
    return p


def scramble(seed, salt_lo, salt_hi, mul_lo, mul_hi):
    """The init statements shared by rocRAND (rocrand_xorwow.h:104-122) and curand_init(seed, 0, 0)."""
    v = list(GOLD["base_state"]["v"])
    d = GOLD["base_state"]["d"]
    s0 = (seed & M32) ^ salt_lo
    s1 = ((seed >> 32) & M32) ^ salt_hi
    t0 = (mul_lo * s0) & M32
    t1 = (mul_hi * s1) & M32
    v[0] = (v[0] + t0) & M32
    v[1] ^= t0
    v[2] = (v[2] + t1) & M32
    v[3] ^= t1
    v[4] = (v[4] + t0) & M32
    d = (d + t1 + t0) & M32
    return d, v


def state_array(d, v):
    st = np.zeros(1, dtype=oracle.XORWOW_DTYPE)
    st["d"][0] = d
    st["v"][0] = v
    return st


SEEDS = [0, 1, 0x5EED, 0x5EED + 1000003 * 3 + 17, 2 ** 40 + 12345, 2 ** 64 - 1]


def test_golden_constants_match_the_oracle():
    for seed in (0, 1, 0x5EED):
        st = oracle.random_states(1, seed)
        want = GOLD["first_states"][str(seed)]
        assert int(st["d"][0]) == want["d"] and [int(x) for x in st["v"][0]] == want["v"]
        assert [int(x) for x in oracle.xorwow_stream(seed, 16)] == GOLD["first_outputs"][str(seed)]
        c = GOLD["curand_init_scramble"]
        d, v = scramble(seed, c["salt_lo"], c["salt_hi"], c["mul_lo"], c["mul_hi"])
        assert (d, v) == (want["d"], want["v"])


def test_marsaglia_reference_stream():
    """Unscrambled base state = the example generator printed in Marsaglia's paper; its first outputs follow from the
    recurrence alone (independent Python restatement, 64-bit ints masked to 32)."""
    v = list(GOLD["base_state"]["v"])
    d = GOLD["base_state"]["d"]
    st = state_array(d, v)
    got = oracle.xorwow_draws(st, 1000)
    for k in range(1000):
        t = v[0] ^ (v[0] >> 2)
        v = v[1:] + [((v[4] ^ (v[4] << 4)) ^ (t ^ (t << 1))) & M32]
        d = (d + GOLD["weyl_increment"]["value"]) & M32
        assert int(got[k]) == (v[4] + d) & M32


@pytest.mark.parametrize("seed", SEEDS)
def test_recurrence_against_rocrand_engine(pin, seed):
    """>= 4M draws per state: oracle orc_xorwow_next == rocRAND xorwow_engine::next on identical state."""
    st = oracle.random_states(1, seed)
    mine = st.copy()
    theirs = np.array([st["d"][0]] + list(st["v"][0]), dtype=np.uint32)
    n = 1 << 16
    a = oracle.xorwow_draws(mine, n)
    b = np.empty(n, np.uint32)
    pin.pin_xorwow_draws(theirs.ctypes.data_as(C.c_void_p), C.c_size_t(n), b.ctypes.data_as(C.c_void_p))
    np.testing.assert_array_equal(a, b)
    # long stream: compare a rotating XOR fold of 4M further draws and the state both sides end in
    big = 1 << 22
    fa = oracle.lib().orc_xorwow_fold(mine.ctypes.data_as(C.c_void_p), C.c_size_t(big))
    fb = pin.pin_xorwow_fold(theirs.ctypes.data_as(C.c_void_p), C.c_size_t(big))
    assert fa == fb
    assert int(mine["d"][0]) == int(theirs[0]) and [int(x) for x in mine["v"][0]] == [int(x) for x in theirs[1:]]


@pytest.mark.parametrize("seed", SEEDS)
def test_init_structure_against_rocrand(pin, seed):
    """rocRAND's constructor == the shared scramble statements with rocRAND's constants; the oracle's curand_init ==
    the same statements with the (unverifiable) cuRAND constants."""
    theirs = np.zeros(6, np.uint32)
    pin.pin_rocrand_init(C.c_uint64(seed), theirs.ctypes.data_as(C.c_void_p))
    r = GOLD["rocrand_init_scramble"]
    d, v = scramble(seed, r["salt_lo"], r["salt_hi"], r["mul_lo"], r["mul_hi"])
    assert [int(x) for x in theirs] == [d] + v
    c = GOLD["curand_init_scramble"]
    d, v = scramble(seed, c["salt_lo"], c["salt_hi"], c["mul_lo"], c["mul_hi"])
    st = oracle.random_states(1, seed)
    assert int(st["d"][0]) == d and [int(x) for x in st["v"][0]] == v


def test_uniform_conversions_closed_form():
    """curand_uniform(x) = float(x) * 2^-32 + 2^-33 (f32, one rounding after cvt.rn); curand_uniform_double: z = x ^
    (y << 21), z * 2^-53 + 2^-54.  Evaluated here in exact arithmetic (float64 holds every intermediate)."""
    n = 1 << 20
    st = oracle.random_states(1, 0x5EED)
    raw = oracle.xorwow_draws(st.copy(), 2 * n)
    got32 = oracle.xorwow_uniforms(st.copy(), n)
    xf = raw[:n].astype(np.float32).astype(np.float64)  # cvt.rn.f32.u32
    want32 = (xf * 2.0 ** -32 + 2.0 ** -33).astype(np.float32)  # exact in f64, then the single rounding
    assert got32.tobytes() == want32.tobytes()
    assert got32.min() > 0.0 and got32.max() <= 1.0
    got64 = oracle.xorwow_uniforms(st.copy(), n, double=True)
    x, y = raw[0::2].astype(np.uint64), raw[1::2].astype(np.uint64)
    z = x ^ (y << np.uint64(21))  # < 2^53: exact in float64
    want64 = z.astype(np.float64) * 2.0 ** -53 + 2.0 ** -54
    assert got64.tobytes() == want64.tobytes()
    assert got64.min() > 0.0 and got64.max() < 1.0
