"""The product's field / curve headers (csrc/ff.hpp, ec.hpp), compiled for the host with -DUG_CHECK_BOUNDS so every
lazy-reduction range stated in ec.hpp is asserted, against the oracle. The same headers are compiled for gfx950."""
import ctypes as C
import os
import random
import subprocess

import pytest

import oracle as O

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ultragroth_amd", "csrc")


@pytest.fixture(scope="module")
def hm():
    subprocess.check_call(["make", "-s", "-C", CSRC, os.path.join(CSRC, "libug_hostmath_test.so")])
    L = C.CDLL(os.path.join(CSRC, "libug_hostmath_test.so"))
    vp = C.c_void_p
    L.ugt_f_op.argtypes = [C.c_int, C.c_int, vp, vp, vp]
    L.ugt_fq_chain.argtypes = [vp, vp, vp, C.c_int]
    L.ugt_fq2_op.argtypes = [C.c_int, vp, vp, vp]
    for n in ("ugt_g1_sum", "ugt_g2_sum"):
        getattr(L, n).argtypes = [vp, vp, vp, C.c_size_t]
    for n in ("ugt_g1_tree_sum", "ugt_g2_tree_sum"):
        getattr(L, n).argtypes = [vp, vp, C.c_size_t]
    for n in ("ugt_g1_mul", "ugt_g2_mul", "ugt_g1_mul_w4", "ugt_g2_mul_w4"):
        getattr(L, n).argtypes = [vp, vp, vp]
    return L


def _sec(buf, sid):
    off, sz = O.section(buf, "zkey", sid)
    return buf[off:off + sz]


def _fop(L, which, op, a, b=None):
    out = C.create_string_buffer(32)
    L.ugt_f_op(which, op, out, O.to_le(a), O.to_le(b) if b is not None else None)
    return O.from_le(out.raw)


@pytest.mark.parametrize("which,mod", [(0, O.R_MOD), (1, O.Q_MOD)])
def test_f29_field_ops(hm, which, mod):
    rng = random.Random(40 + which)
    vals = [0, 1, 2, mod - 1, mod - 2, (1 << 256) % mod, (1 << 255) % mod, mod // 2] + [rng.randrange(mod) for _ in range(400)]
    names = {0: "mul", 1: "add", 2: "sub"}
    for i, a in enumerate(vals):
        b = vals[(i * 7 + 3) % len(vals)]
        for op in (0, 1, 2):
            assert _fop(hm, which, op, a, b) == O.f_op(names[op], which, a, b), (op, hex(a), hex(b))
        assert _fop(hm, which, 3, a) == O.f_op("neg", which, a)
        assert _fop(hm, which, 5, a) == O.f_op("mul", which, a, a)
        assert _fop(hm, which, 6, a) == a            # to_normal / from_normal round trip
        assert _fop(hm, which, 7, a) == a            # pack256 / unpack256 round trip
        if i < 40:
            assert _fop(hm, which, 4, a) == O.f_op("inv", which, a)


@pytest.mark.parametrize("which,mod", [(0, O.R_MOD), (1, O.Q_MOD)])
def test_f29_inversion_by_divsteps(hm, which, mod):
    """inv() is the constant-time divsteps ("safegcd") form since round 3: against the oracle's inverse, against the Fermat form it
    replaced, on a lazily reduced operand (a + 5q), and on the values where a gcd walk is shortest or longest (0, 1, powers of
    two, q - 1, q - 2^k, the Montgomery constants)"""
    rng = random.Random(400 + which)
    R = 1 << 256
    edge = [0, 1, 2, 3, mod - 1, mod - 2, mod // 2, mod // 2 + 1, R % mod, (R * R) % mod, pow(R, -1, mod)]
    edge += [(1 << k) % mod for k in range(1, 256, 15)] + [(mod - (1 << k)) % mod for k in range(1, 254, 17)]
    edge += [pow(3, k, mod) for k in (5, 77, 200)]
    for a in edge + [rng.randrange(mod) for _ in range(600)]:
        want = O.f_op("inv", which, a)
        assert _fop(hm, which, 4, a) == want, hex(a)
        assert _fop(hm, which, 9, a) == want, hex(a)
    for a in edge + [rng.randrange(mod) for _ in range(40)]:
        assert _fop(hm, which, 8, a) == O.f_op("inv", which, a), hex(a)


def test_f29_lazy_chain(hm):
    rng = random.Random(6)
    R = 1 << 256
    q = O.Q_MOD
    Rinv = pow(R, -1, q)
    for _ in range(20):
        a, b = rng.randrange(q), rng.randrange(q)
        out = C.create_string_buffer(32)
        hm.ugt_fq_chain(out, O.to_le(a), O.to_le(b), 25)
        # the same chain on plain residues (inputs/outputs are Montgomery R=2^256 values)
        x, y = a * Rinv % q, b * Rinv % q
        acc = x
        for _ in range(25):
            t = (acc * y + x - y) % q
            t = t * t % q
            t = (t * x - y) % q
            acc = (t + 2 * acc) % q
        assert O.from_le(out.raw) == acc * R % q


def test_fq2_ops(hm):
    rng = random.Random(8)
    q, R = O.Q_MOD, 1 << 256
    Rinv = pow(R, -1, q)
    enc = lambda v: O.to_le(v * R % q)
    for _ in range(50):
        a0, a1, b0, b1 = (rng.randrange(q) for _ in range(4))
        out = C.create_string_buffer(64)
        hm.ugt_fq2_op(0, out, enc(a0) + enc(a1), enc(b0) + enc(b1))
        assert (O.mont_decode(out.raw[:32]), O.mont_decode(out.raw[32:])) == ((a0 * b0 - a1 * b1) % q, (a0 * b1 + a1 * b0) % q)
        hm.ugt_fq2_op(1, out, enc(a0) + enc(a1), None)
        assert (O.mont_decode(out.raw[:32]), O.mont_decode(out.raw[32:])) == ((a0 * a0 - a1 * a1) % q, 2 * a0 * a1 % q)
        hm.ugt_fq2_op(2, out, enc(a0) + enc(a1), None)
        n = pow(a0 * a0 + a1 * a1, -1, q)
        assert (O.mont_decode(out.raw[:32]), O.mont_decode(out.raw[32:])) == (a0 * n % q, (-a1) * n % q)


def test_curve_formulas_and_exceptional_cases(hm, zkey):
    rng = random.Random(2)
    A, B2 = _sec(zkey, 5), _sec(zkey, 7)
    n = 200
    pts = A[:64 * n]
    signs = bytes(rng.randrange(2) for _ in range(n))
    sc = b"".join(O.to_le(O.R_MOD - 1 if s else 1) for s in signs)
    out = C.create_string_buffer(64)
    hm.ugt_g1_sum(out, pts, signs, n)
    assert out.raw == O.g1_msm(pts, sc, n)
    p = A[64 * 5:64 * 6]
    hm.ugt_g1_sum(out, p * 7, None, 7)                       # doubling branch of the mixed add
    assert out.raw == O.g1_mul(p, 7)
    hm.ugt_g1_sum(out, p * 2, bytes([0, 1]), 2)              # P - P = infinity
    assert out.raw == bytes(64)
    hm.ugt_g1_sum(out, p * 3, bytes([0, 1, 0]), 3)           # resume from infinity
    assert out.raw == p
    hm.ugt_g1_tree_sum(out, pts, n)                          # general XYZZ + XYZZ adds
    assert out.raw == O.g1_msm(pts, O.to_le(1) * n, n)
    hm.ugt_g1_tree_sum(out, p * 8, 8)                        # doubling branch of the general add
    assert out.raw == O.g1_mul(p, 8)
    for k in (0, 1, 2, 3, O.R_MOD - 1, O.R_MOD, rng.randrange(1 << 256)):
        hm.ugt_g1_mul(out, p, O.to_le(k))
        assert out.raw == O.g1_mul(p, k)
    # the signed 4-bit window form of the provers' host parts: every digit value, carries through every window, the top carry
    for k in [0, 1, 7, 8, 9, 15, 16, 0x88888888, (1 << 256) - 1, (1 << 255) + 9, int("9" * 64, 16), int("8" * 64, 16), O.R_MOD - 1] + \
             [rng.randrange(1 << 256) for _ in range(6)]:
        hm.ugt_g1_mul_w4(out, p, O.to_le(k))
        assert out.raw == O.g1_mul(p, k), hex(k)
    n2 = 100
    pts2 = B2[:128 * n2]
    signs = bytes(rng.randrange(2) for _ in range(n2))
    sc = b"".join(O.to_le(O.R_MOD - 1 if s else 1) for s in signs)
    out2 = C.create_string_buffer(128)
    hm.ugt_g2_sum(out2, pts2, signs, n2)
    assert out2.raw == O.g2_msm(pts2, sc, n2)
    p2 = B2[128 * 4:128 * 5]
    hm.ugt_g2_sum(out2, p2 * 5, None, 5)
    assert out2.raw == O.g2_mul(p2, 5)
    hm.ugt_g2_sum(out2, p2 * 2, bytes([0, 1]), 2)
    assert out2.raw == bytes(128)
    hm.ugt_g2_tree_sum(out2, pts2, n2)
    assert out2.raw == O.g2_msm(pts2, O.to_le(1) * n2, n2)
    for k in (1, 2, O.R_MOD - 1, rng.randrange(1 << 256)):
        hm.ugt_g2_mul(out2, p2, O.to_le(k))
        assert out2.raw == O.g2_mul(p2, k)
    for k in (0, 8, 9, (1 << 256) - 1, int("8" * 64, 16), rng.randrange(1 << 256)):
        hm.ugt_g2_mul_w4(out2, p2, O.to_le(k))
        assert out2.raw == O.g2_mul(p2, k), hex(k)


def test_shoup_product_of_the_ntt(hm):
    """ff.hpp mul_shoup / shoup_quotient: x * w mod r for a table constant w with w' = floor(w 2^261 / r) -- the product the NTT
    butterflies make since round 5 -- on un-normalised limb sums (four addends, limbs up to 2^31) and edge constants; the raw
    result stays below 3r (the bound the kernel's subtraction tables and contraction rely on)"""
    hm.ugt_fr_mul_shoup.argtypes = [C.c_void_p] * 6
    rng = random.Random(2029)
    R = O.R_MOD
    out = C.create_string_buffer(32)
    ws = [1, 2, R - 1, R - 2, (R - 1) // 2, pow(5, (R - 1) >> 20, R), pow(5, (R - 1) >> 28, R)] + [rng.randrange(1, R) for _ in range(40)]
    for w in ws:
        for trial in range(6):
            if trial == 0:
                xs = [(1 << 256) - 1] * 4                           # the largest limbs the unpacked form can hold, four times
            elif trial == 1:
                xs = [0, 0, 0, 0]
            elif trial == 2:
                xs = [R - 1, R - 1, 2 * R - 1, (1 << 256) - 1]
            else:
                xs = [rng.randrange(1 << 256) for _ in range(4)]
            ok = hm.ugt_fr_mul_shoup(out, *[O.to_le(x) for x in xs], O.to_le(w))
            assert int.from_bytes(out.raw, "little") == sum(xs) * w % R, (hex(w), trial)
            assert ok == 1, (hex(w), trial)


def test_segment_map_invariants(hm):
    """csrc/segmap.hpp: the cut of an MSM schedule's sorted entries into long segments and, for the last eighth, short
    ones. Every kernel of msm.hip derives positions from this map (on the device, from the number of valid entries), so
    its invariants are checked here for sizes around every boundary: segments tile the entries in order, seg_of inverts
    first_entry, the lane-transposed position stays inside the wave's tile, and the host's upper bound of the segment
    count (grid and slot sizes) holds for every n_valid <= total."""
    hm.ugt_segmap_check.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_int]
    rng = random.Random(77)
    for log_a, log_b in ((7, 5), (7, 7), (6, 6), (5, 5), (7, 6)):
        tile = 64 << log_a
        sizes = {0, 1, 63, 64, 65, tile - 1, tile, tile + 1, 8 * tile - 1, 8 * tile, 8 * tile + 1, 9 * tile + 77, 64 * tile + 5}
        sizes |= {rng.randrange(1, 3_000_000) for _ in range(40)}
        for n in sorted(sizes):
            for total in (n, n + n // 3 + 1, 2 * n + tile):
                assert hm.ugt_segmap_check(n, total, log_a, log_b) == 0, (n, total, log_a, log_b)
