"""XORWOW: the oracle's from-scratch implementation (GF(2) skip-ahead) against rocRAND 4.2.0's host
engine and against the known-answer vector recorded in SURVEY.md Appendix C."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))

# seed 12345, subsequence 7, offset 3
KAT_WORDS = [0x06EFECAB, 0x473F4D25, 0xFC0B38B5, 0x8E32EB99]
KAT_NEXT_UNIFORM = 0.146008611


def _orc_stream(ob, seed, sub, off, n=8):
    L = ob.lib()
    st = (C.c_uint32 * 6)()
    L.orc_xorwow_init(seed, sub, off, st)
    words = [L.orc_xorwow_next(st) for _ in range(n)]
    return words, L.orc_xorwow_uniform(st)


def test_known_answer_vector(ob):
    words, _ = _orc_stream(ob, 12345, 7, 3, 4)
    assert words == KAT_WORDS
    L = ob.lib()
    st = (C.c_uint32 * 6)()
    L.orc_xorwow_init(12345, 7, 3, st)
    for _ in range(4):
        L.orc_xorwow_next(st)
    assert abs(L.orc_xorwow_uniform(st) - KAT_NEXT_UNIFORM) < 5e-10


def test_uniform_range_and_formula(ob):
    L = ob.lib()
    st = (C.c_uint32 * 6)()
    L.orc_xorwow_init(1, 0, 0, st)
    for _ in range(2000):
        st2 = (C.c_uint32 * 6)(*st)
        w = L.orc_xorwow_next(st2)
        u = L.orc_xorwow_uniform(st)
        expect = np.float32(2.3283064e-10) + np.float32(w) * np.float32(2.3283064e-10)
        assert np.float32(u) == expect
        assert 0.0 < u <= 1.0


def test_offset_equals_sequential_steps(ob):
    """init(seed, y, x) == init(seed, y, 0) advanced x times: what K1 relies on."""
    L = ob.lib()
    a = (C.c_uint32 * 6)()
    L.orc_xorwow_init(99, 5, 0, a)
    for x in range(0, 300):
        b = (C.c_uint32 * 6)()
        L.orc_xorwow_init(99, 5, x, b)
        assert list(a) == list(b), x
        L.orc_xorwow_next(a)


@pytest.mark.skipif(shutil.which("g++") is None or not os.path.exists("/opt/rocm/include/rocrand/rocrand_xorwow.h"),
                    reason="needs g++ and the rocRAND headers")
def test_matches_rocrand_host_engine(ob):
    out_dir = os.path.join(HERE, "_build")
    os.makedirs(out_dir, exist_ok=True)
    exe = os.path.join(out_dir, "rocrand_kat")
    src = os.path.join(HERE, "helpers", "rocrand_kat.cpp")
    if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-I/opt/rocm/include", src, "-o", exe])
    triples = [(12345, 7, 3), (0, 0, 0), (1, 1, 1), (0xDEADBEEFCAFE, 4129, 6199), (42, 1 << 20, 1 << 33), (7, 3071, 4095)]
    args = [str(v) for t in triples for v in t]
    lines = subprocess.check_output([exe] + args, text=True).strip().splitlines()
    assert len(lines) == len(triples)
    for (seed, sub, off), line in zip(triples, lines):
        toks = line.split()
        ref_words = [int(t, 16) for t in toks[:8]]
        ref_uniform = float(toks[8])
        words, uni = _orc_stream(ob, seed, sub, off, 8)
        assert words == ref_words, (seed, sub, off)
        assert abs(uni - ref_uniform) < 1e-9
