"""The oracle's RNG and portable math (the definitions the GPU reproduces bit for bit)."""
from __future__ import annotations

import math

import numpy as np


def ulp_diff(a, b):
    ia = np.array(a, dtype=np.float64).view(np.int64)
    ib = np.array(b, dtype=np.float64).view(np.int64)
    return np.abs(ia - ib)


def test_philox4x32_10_known_answer_vectors(orc):
    """Random123 kat_vectors for philox4x32 with 10 rounds."""
    assert orc.philox((0, 0, 0, 0), (0, 0)) == (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)
    assert orc.philox((0xffffffff,) * 4, (0xffffffff,) * 2) == (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)
    assert orc.philox((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0)) == (
        0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)


def test_step_uniforms_are_strictly_inside_unit_interval_and_counter_based(orc):
    u = np.array([orc.step_uniforms(7, e, 0, t) for e in range(50) for t in range(50)])
    assert u.min() > 0.0 and u.max() < 1.0
    assert abs(u.mean() - 0.5) < 0.01
    assert orc.step_uniforms(7, 3, 0, 9) == orc.step_uniforms(7, 3, 0, 9)
    assert orc.step_uniforms(7, 3, 0, 9) != orc.step_uniforms(7, 3, 1, 9)
    w = orc.philox((9, 0, 3, 0), (7, 0))
    assert orc.step_uniforms(7, 3, 0, 9) == tuple((x + 0.5) * 2.0 ** -32 for x in w)


def test_polar_normal_moments(orc):
    z = np.array([orc.std_normal(1234, e, 0, t)[0] for e in range(200) for t in range(100)])
    assert abs(z.mean()) < 0.03 and abs(z.std() - 1.0) < 0.03
    assert abs(((z ** 3).mean())) < 0.1 and abs((z ** 4).mean() - 3.0) < 0.2


def test_portable_exp_log_within_one_ulp_of_libm(orc):
    L = orc.lib()
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.uniform(-700, 700, 20000), rng.uniform(-1, 1, 20000), [0.0, -0.0, 1e-10, 24.0, -500.0]])
    got = np.array([L.wedm_oracle_exp(float(x), orc.MATH_PORTABLE) for x in xs])
    want = np.exp(xs)
    assert ulp_diff(got, want).max() <= 1
    xs = np.concatenate([rng.uniform(0, 1, 20000)[1:], 2.0 ** rng.uniform(-104, 0, 20000)])
    got = np.array([L.wedm_oracle_log(float(x), orc.MATH_PORTABLE) for x in xs])
    assert ulp_diff(got, np.log(xs)).max() <= 1


def test_portable_cube_is_correctly_rounded(orc):
    from fractions import Fraction

    L = orc.lib()
    rng = np.random.default_rng(1)
    for x in rng.uniform(0, 4, 5000):
        assert L.wedm_oracle_cube(float(x), orc.MATH_PORTABLE) == float(Fraction(float(x)) ** 3)
    assert all(L.wedm_oracle_cube(float(x), orc.MATH_LIBM) == float(x) ** 3 for x in rng.uniform(0, 4, 2000))


def test_python_floor_division_semantics(orc):
    L = orc.lib()
    rng = np.random.default_rng(2)
    for a, b in zip(rng.uniform(0, 50, 5000), rng.choice([0.1, 0.2, 0.25, 0.3, 0.625], 5000)):
        assert L.wedm_oracle_py_floordiv(float(a), float(b)) == float(a) // float(b)
    assert int(L.wedm_oracle_py_floordiv(30.0, 0.2)) == 149 and int(L.wedm_oracle_py_floordiv(20.0, 0.2)) == 99
