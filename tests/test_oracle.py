"""CPU tests of the oracle itself: pinned to the reference's fixture, SURVEY.md Appendix A and the reference's own
field layer (oracle/_ref, built from /root/reference/build when that tree is present)."""
import hashlib
import json
import random

import pytest

import oracle as O
from oracle import pairing
from conftest import fixed_rs


def _sec(buf, ftype, sid):
    off, sz = O.section(buf, ftype, sid)
    return buf[off:off + sz]


def test_fixture_header(zkey, wtns):
    info = O.zkey_info(zkey)
    assert info == dict(nVars=1003, nPublic=1, domainSize=1024, ultra=False, nCoefs=2002)
    # section layout recorded in SURVEY.md Appendix A
    assert O.section(zkey, "zkey", 4) == (852, 88092)
    assert O.section(zkey, "zkey", 7) == (217364, 128384)
    assert O.section(zkey, "zkey", 9) == (409836, 65536)
    w = _sec(wtns, "wtns", 2)
    assert [O.from_le(w[32 * i:32 * i + 32]) for i in (0, 2, 3, 4, 5)] == [1, 3, 11, 20, 411]


def test_known_answer_proof(zkey, wtns, vkey):
    r, s = fixed_rs()
    proof, pub, raw = O.groth16_prove(zkey, wtns, int.from_bytes(r, "little"), int.from_bytes(s, "little"), want_raw=True)
    assert hashlib.sha256(proof.encode()).hexdigest() == "11767e6c2a13edf5c87e282bc652346b4e62e0faa4156275edbef96c25762592"
    assert pub == '["7713112592372404476342535432037683616424591277138491596200192981572885523208"]'
    dec = lambda o: O.mont_decode(raw[o:o + 32])
    assert (dec(0), dec(32)) == (21344626350637401086172020791957193447896122895487902080665101052945368184094,
                                 6713982217719299172625616378045403143338774554356668389789068804132537304884)     # MSM_A
    assert dec(64) == 8193048668315265143422292719263587730195930046948232152095219516163543912381               # MSM_B1.x
    assert dec(128) == 7062321476789293854033358658370016416373968873586528291130727575191400769869              # MSM_B2.x.a
    assert dec(256) == 16457540699400983235344720173412980693113658479595934480322634826421860999224             # MSM_C.x
    assert dec(320) == 12494683413857414348299179237330527902580898532379008589172168355592525264032             # MSM_H.x
    # the reference's acceptance test: verifier accepts, and rejects after public[0] -= 1 (build.yml:69-81)
    assert pairing.groth16_verify(vkey, pub, proof)
    bad = json.loads(pub)
    bad[0] = str(int(bad[0]) - 1)
    assert not pairing.groth16_verify(vkey, bad, proof)


def test_h_polynomial_known_answers(zkey, wtns):
    info = O.zkey_info(zkey)
    h, abc = O.hpoly(_sec(zkey, "zkey", 4)[4:], info["nCoefs"], _sec(wtns, "wtns", 2), info["nVars"], 1024, want_abc=True)
    assert hashlib.sha256(h).hexdigest() == "44ca2358066ca82cffac3cc7f58f163f8cd806059850dea84e12cde01ff3b872"
    assert O.from_le(h[:32]) == 19249825615842551751830010770668855053805061129563596815467143610091464650162
    assert O.from_le(h[32 * 1023:]) == 8238469065222963538316179134130322574598295596105653466880219272298245475871
    assert O.mont_decode(abc[:32], O.R_MOD) == 12883692537473569801090764782517160015411006304666400525877627806088311585420
    assert O.root_of_unity(11) * pow(1 << 256, -1, O.R_MOD) % O.R_MOD == \
        1120550406532664055539694724667294622065367841900378087843176726913374367458


@pytest.mark.parametrize("which,mod,name", [(O.FR, O.R_MOD, "fr"), (O.FQ, O.Q_MOD, "fq")])
def test_field_against_python_and_reference(which, mod, name):
    rng = random.Random(which)
    R = 1 << 256
    Rinv = pow(R, -1, mod)
    vals = [0, 1, 2, mod - 1, mod - 2, R % mod, (R * R) % mod] + [rng.randrange(mod) for _ in range(200)]
    for i, a in enumerate(vals):
        b = vals[(7 * i + 3) % len(vals)]
        assert O.f_op("mul", which, a, b) == a * b * Rinv % mod
        assert O.f_op("add", which, a, b) == (a + b) % mod
        assert O.f_op("sub", which, a, b) == (a - b) % mod
        assert O.f_op("neg", which, a) == (-a) % mod
        assert O.f_op("to_mont", which, a) == a * R % mod
        assert O.f_op("from_mont", which, a) == a * Rinv % mod
        if O.ref is not None:                       # the reference's own code (build/f{r,q}*.cpp + GMP)
            assert O.ref_op(name, "mul", a, b) == O.f_op("mul", which, a, b)
            assert O.ref_op(name, "add", a, b) == O.f_op("add", which, a, b)
            assert O.ref_op(name, "sub", a, b) == O.f_op("sub", which, a, b)
            assert O.ref_op(name, "neg", a) == O.f_op("neg", which, a)
            assert O.ref_op(name, "to_mont", a) == O.f_op("to_mont", which, a)
            assert O.ref_op(name, "from_mont", a) == O.f_op("from_mont", which, a)
            if a and i < 40:
                assert O.ref_op(name, "inv", a) == O.f_op("inv", which, a)
    # operands >= q in the top limb (unreduced inputs, as fromMpz can hand to toMontgomery)
    if O.ref is not None:
        for a in (mod, mod + 5, (1 << 256) - 1, (1 << 255) + 12345):
            assert O.ref_op(name, "to_mont", a) == O.f_op("to_mont", which, a)


@pytest.mark.parametrize("which,mod", [(O.FR, O.R_MOD), (O.FQ, O.Q_MOD)])
def test_field_product_asm_form_equals_portable_form(which, mod):
    """field.h holds the Montgomery product twice: the portable CIOS restatement (fe_mul_c) and, on x86_64 with BMI2 + ADX, the
    mulx / adcx / adox form the CPU baseline runs on; same bits on random, edge and unreduced operands"""
    import ctypes as C
    rng = random.Random(17 + which)
    edge = [0, 1, mod - 1, mod, mod + 1, (1 << 256) - 1, (1 << 255), (1 << 64) - 1, 1 << 64, (1 << 192) - 1]
    vals = edge + [rng.randrange(1 << 256) for _ in range(300)] + [rng.randrange(mod) for _ in range(300)]
    for i, a in enumerate(vals):
        b = vals[(11 * i + 5) % len(vals)]                # (either operand may be unreduced: zkey coefficients come as they are)
        x, y = C.create_string_buffer(O.to_le(a), 32), C.create_string_buffer(O.to_le(b), 32)
        r1, r2 = C.create_string_buffer(32), C.create_string_buffer(32)
        O.lib.ugo_f_mul(which, r1, x, y)
        O.lib.ugo_f_mul_portable(which, r2, x, y)
        assert r1.raw == r2.raw, (hex(a), hex(b))
        want = a * b * pow(1 << 256, -1, mod) % mod
        # one conditional subtraction, as the reference: canonical when one operand is reduced, congruent otherwise
        assert O.from_le(r1.raw) % mod == want and (b >= mod and a >= mod or O.from_le(r1.raw) == want)


def test_pippenger_against_double_and_add(zkey):
    rng = random.Random(3)
    for n in (0, 1, 2, 33, 300):
        sc = b"".join(O.to_le(rng.choice([0, 1, rng.randrange(O.R_MOD), O.R_MOD - 1, rng.randrange(1 << 256)])) for _ in range(n))
        pts = _sec(zkey, "zkey", 5)[:64 * n]
        assert O.g1_msm(pts, sc, n) == O.g1_msm(pts, sc, n, naive=True)
        pts2 = _sec(zkey, "zkey", 7)[:128 * min(n, 40)]
        m = min(n, 40)
        assert O.g2_msm(pts2, sc[:32 * m], m) == O.g2_msm(pts2, sc[:32 * m], m, naive=True)


def test_fixture_points_on_curve(zkey):
    import ctypes as C
    a = _sec(zkey, "zkey", 5)
    b2 = _sec(zkey, "zkey", 7)
    for i in range(0, 1003, 17):
        assert O.lib.ugo_g1_on_curve(a[64 * i:64 * i + 64]) == 1
        assert O.lib.ugo_g2_on_curve(b2[128 * i:128 * i + 128]) == 1


def test_ntt_against_direct_dft():
    rng = random.Random(9)
    logn = 4
    n = 1 << logn
    R = 1 << 256
    xs = [rng.randrange(O.R_MOD) for _ in range(n)]
    data = b"".join(O.to_le(x * R % O.R_MOD) for x in xs)
    w = O.root_of_unity(logn) * pow(R, -1, O.R_MOD) % O.R_MOD
    exp = [sum(xs[j] * pow(w, j * k, O.R_MOD) for j in range(n)) % O.R_MOD for k in range(n)]
    out = O.ntt(data, logn)
    assert [O.mont_decode(out[32 * k:32 * k + 32], O.R_MOD) for k in range(n)] == exp
    assert O.ntt(out, logn, inverse=True) == data


def test_keccak256_vectors():
    assert O.keccak256(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
    assert O.keccak256(b"abc").hex() == "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"


needs_ref = pytest.mark.skipif(O.ref is None or not hasattr(O.ref, "ref_lookup_row"), reason="oracle/_ref (reference field layer) not built")


@needs_ref
def test_lookup_row_against_the_reference_overloads():
    """compute_lookup's table row (src/ultra_groth.cpp:72-79) against the reference's own RawFr: `frequencies[i]` is a
    uint32_t that binds to mul(int, Element) (build/fr.hpp:251), so values >= 2^31 enter as freq - 2^32 + r"""
    rng = random.Random(77)
    R = 1 << 256
    for t in range(60):
        rand = rng.randrange(O.R_MOD)
        i = rng.choice([0, 1, 2, 255, 65535, rng.randrange(1 << 20)])
        f = rng.choice([0, 1, 7, (1 << 31) - 1, 1 << 31, (1 << 31) + 5, (1 << 32) - 1, rng.randrange(1 << 32)])
        got = O.lookup_row(i, f, rand * R % O.R_MOD)
        assert got == O.ref_lookup_row(i, f, rand * R % O.R_MOD)
        s = (i + rand) % O.R_MOD
        inv = pow(s, -1, O.R_MOD) if s else 0
        signed = f - (1 << 32) if f >= (1 << 31) else f
        assert got == (inv, signed * inv % O.R_MOD)
    # i + rand = 0: mpz_invert leaves 0, and so do we
    assert O.lookup_row(5, 9, (O.R_MOD - 5) * R % O.R_MOD) == O.ref_lookup_row(5, 9, (O.R_MOD - 5) * R % O.R_MOD) == (0, 0)


@needs_ref
def test_derive_challenge_against_the_reference_code(zkey):
    """derive_challenge (src/ultra_groth.cpp:33-58): big-endian x || y, the reference's Keccak-256, reduction mod r"""
    a = _sec(zkey, "zkey", 5)
    for k in range(0, 400, 7):
        rec = a[64 * k:64 * k + 64]
        if rec == bytes(64):
            continue
        x, y = O.mont_decode(rec[:32]), O.mont_decode(rec[32:])
        exp = int.from_bytes(O.keccak256(x.to_bytes(32, "big") + y.to_bytes(32, "big")), "big") % O.R_MOD
        assert O.derive_challenge(rec) == O.ref_derive_challenge(rec) == exp
