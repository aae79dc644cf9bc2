"""Expected Groth16 proof of a synthetic generator-walk circuit WITHOUT a CPU MSM -- TEST INFRASTRUCTURE ONLY.

The synthetic circuits of ``ultragroth_amd.synth`` use base points P_i = (seed + i) * G, so each of the five
multi-scalar multiplications of the prover (src/groth16.cpp:55,58,61,64,154) has a closed form in the exponent:

    sum_i s_i * P_i  =  (sum_i s_i * (seed + i) mod r) * G .

The H-polynomial block (S5-S9, src/groth16.cpp:66-148) is computed by the CPU oracle at full size (ugo_hpoly), the
five sums by one dot product mod r each (ugo_fr_dot_walk) and one scalar multiplication of the generator, and the
blinding and JSON by the oracle's restatement of S11-S13 (ugo_groth16_finish). This makes a byte-for-byte expected
proof.json affordable at 2^24 and 2^26 constraints, where a CPU Pippenger would take minutes to hours.
"""
import ctypes as C

from . import lib, section, zkey_info, g1_mul, g2_mul, groth16_finish, header_only_zkey, from_le


def _addr(buf):
    """address of the first byte of a bytes object or a ctypes array (no copy)"""
    if isinstance(buf, (bytes, bytearray)):
        return C.cast(C.c_char_p(bytes(buf) if isinstance(buf, bytearray) else buf), C.c_void_p).value
    return C.addressof(buf)


def _dot_walk(ptr, n, seed):
    out = C.create_string_buffer(32)
    lib.ugo_fr_dot_walk(out, C.c_void_p(ptr), n, seed)
    return from_le(out.raw)


def groth16_expected(zkey, wtns, seeds, g1_generator, g2_generator, r, s, g1_only=False, progress=None, b_zero_mask=None):
    """(proof_json, public_json) the prover must produce for (zkey, wtns) with blinding scalars r, s (ints < 2^248).

    seeds: {"A","B1","B2","C","H"} -> walk seed of each section; g1_only: B1, B2, C sections are all-infinity;
    b_zero_mask (numpy bool per signal): the signals whose B1 and B2 points are the point at infinity (their scalars drop out)."""
    say = progress or (lambda msg: None)
    info = zkey_info(zkey)
    nv, npub, dom, ncoefs = info["nVars"], info["nPublic"], info["domainSize"], info["nCoefs"]
    zbase, wbase = _addr(zkey), _addr(wtns)
    off4, _ = section(zkey, "zkey", 4)
    offw, szw = section(wtns, "wtns", 2)
    assert szw >= nv * 32
    w_ptr = wbase + offw
    say("oracle H polynomial, domain %d" % dom)
    h = C.create_string_buffer(dom * 32)
    if lib.ugo_hpoly(h, C.c_void_p(zbase + off4 + 4), ncoefs, C.c_void_p(w_ptr), nv, dom, None):
        raise ValueError("coefficient index out of range")
    say("closed-form sums")
    zero1, zero2 = bytes(64), bytes(128)
    ka = _dot_walk(w_ptr, nv, seeds["A"])
    sum_a = g1_mul(g1_generator, ka) if ka else zero1
    if g1_only:
        sum_b1, sum_b2, sum_c = zero1, zero2, zero1
    else:
        wb_ptr, keep = w_ptr, None
        if b_zero_mask is not None:
            import numpy as np
            keep = np.frombuffer(C.string_at(w_ptr, nv * 32), dtype=np.uint8).reshape(nv, 32).copy()
            keep[np.asarray(b_zero_mask, dtype=bool)] = 0
            wb_ptr = keep.ctypes.data
        kb1 = _dot_walk(wb_ptr, nv, seeds["B1"])
        kb2 = _dot_walk(wb_ptr, nv, seeds["B2"])
        del keep
        kc = _dot_walk(w_ptr + (npub + 1) * 32, nv - npub - 1, seeds["C"])
        sum_b1 = g1_mul(g1_generator, kb1) if kb1 else zero1
        sum_b2 = g2_mul(g2_generator, kb2) if kb2 else zero2
        sum_c = g1_mul(g1_generator, kc) if kc else zero1
    kh = _dot_walk(C.addressof(h), dom, seeds["H"])
    sum_h = g1_mul(g1_generator, kh) if kh else zero1
    public_w = C.string_at(w_ptr, (npub + 1) * 32)
    say("blinding + JSON")
    return groth16_finish(header_only_zkey(zkey), sum_a + sum_b1 + sum_b2 + sum_c + sum_h, public_w, r, s)
