"""CPU oracle for the Groth16 / UltraGroth hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package, and only as the checker / reported baseline. The product
(``ultragroth_amd``) never imports it.

``oracle.lib``      ctypes binding of ``libug_oracle.so`` (plain-C restatement, ug_oracle.c)
``oracle.ref``      ctypes binding of ``_ref/libref_field.so`` (the reference's own field layer,
                    built from /root/reference/build by oracle/Makefile) or None when absent
``oracle.pairing``  pure-Python BN254 pairing check = the reference's acceptance test
                    (verifier, src/groth16.cpp:314-364; CI .github/workflows/build.yml:69-81)
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
FR, FQ = 0, 1

R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617
Q_MOD = 21888242871839275222246405745257275088696311157297823662689037894645226208583
MONT_R = 1 << 256


def build(force=False):
    """Compile libug_oracle.so (and _ref when the reference tree is present)."""
    so = os.path.join(_HERE, "libug_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("ug_oracle.c", "field.h", "curve_tmpl.inc", "ug_oracle.h")]
    stale = force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, os.path.join(_HERE, "libug_oracle.so")])
    if os.path.isdir("/root/reference/build") and (force or not os.path.exists(os.path.join(_HERE, "_ref", "libref_field.so"))):
        subprocess.call(["make", "-s", "-C", _HERE, "ref"])
    return so


def _load():
    so = build()
    L = C.CDLL(so)
    u64p, u8p, vp = C.POINTER(C.c_uint64), C.c_char_p, C.c_void_p
    L.ugo_zkey_info.argtypes = [vp, C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
    L.ugo_section.argtypes = [vp, C.c_uint64, C.c_char_p, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.ugo_groth16_prove.argtypes = [vp, C.c_uint64, vp, C.c_uint64, vp, vp, vp, C.c_uint64, vp, C.c_uint64,
                                    vp, C.POINTER(C.c_double), vp, C.c_uint64]
    L.ugo_ultra_groth_prove.argtypes = [vp, C.c_uint64, vp, C.c_uint64, vp, vp, vp, vp, C.c_uint64, vp,
                                        C.c_uint64, vp, C.c_uint64]
    L.ugo_groth16_finish.argtypes = [vp, C.c_uint64, vp, vp, vp, vp, vp, C.c_uint64, vp, C.c_uint64, vp, C.c_uint64]
    L.ugo_hpoly.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint32, C.c_uint32, vp]
    L.ugo_fr_ntt.argtypes = [vp, C.c_int, C.c_int]
    L.ugo_fr_root_of_unity.argtypes = [vp, C.c_int]
    for name in ("ugo_g1_msm", "ugo_g1_msm_naive", "ugo_g2_msm", "ugo_g2_msm_naive"):
        getattr(L, name).argtypes = [vp, vp, vp, C.c_size_t]
    for name in ("ugo_g1_mul", "ugo_g2_mul", "ugo_g1_add", "ugo_g2_add"):
        getattr(L, name).argtypes = [vp, vp, vp]
    L.ugo_g1_on_curve.argtypes = [vp]
    L.ugo_g2_on_curve.argtypes = [vp]
    for name in ("ugo_f_mul", "ugo_f_mul_portable", "ugo_f_add", "ugo_f_sub"):
        getattr(L, name).argtypes = [C.c_int, vp, vp, vp]
    for name in ("ugo_f_neg", "ugo_f_to_mont", "ugo_f_from_mont", "ugo_f_inv"):
        getattr(L, name).argtypes = [C.c_int, vp, vp]
    L.ugo_f_mul_vec.argtypes = [C.c_int, vp, vp, vp, C.c_size_t]
    L.ugo_keccak256.argtypes = [vp, vp, C.c_uint64]
    L.ugo_derive_challenge.argtypes = [vp, vp]
    L.ugo_fr_dot_walk.argtypes = [vp, vp, C.c_uint64, C.c_uint64]
    L.ugo_lookup_row.argtypes = [vp, vp, C.c_uint32, C.c_uint32, vp]
    return L


lib = _load()


def _load_ref():
    p = os.path.join(_HERE, "_ref", "libref_field.so")
    if not os.path.exists(p):
        return None
    try:
        R = C.CDLL(p)
    except OSError:
        return None
    if hasattr(R, "ref_lookup_row"):
        R.ref_lookup_row.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_void_p]
        R.ref_derive_challenge.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    return R


ref = _load_ref()


# ------------------------------------------------------------------ small helpers

def to_le(x, n=32):
    return int(x).to_bytes(n, "little")


def from_le(b):
    return int.from_bytes(bytes(b), "little")


def f_op(name, which, *args):
    """Run ugo_f_<name> on python ints / 32-byte values; returns int."""
    out = C.create_string_buffer(32)
    bufs = [C.create_string_buffer(to_le(a) if isinstance(a, int) else bytes(a), 32) for a in args]
    getattr(lib, "ugo_f_" + name)(which, out, *bufs)
    return from_le(out.raw)


def ref_op(field, name, *args):
    out = C.create_string_buffer(32)
    bufs = [C.create_string_buffer(to_le(a) if isinstance(a, int) else bytes(a), 32) for a in args]
    getattr(ref, "ref_%s_%s" % (field, name))(out, *bufs)
    return from_le(out.raw)


def zkey_info(zkey):
    out = (C.c_uint32 * 4)()
    nc = C.c_uint64()
    if lib.ugo_zkey_info(zkey, len(zkey), out, C.byref(nc)):
        raise ValueError("bad zkey")
    return dict(nVars=out[0], nPublic=out[1], domainSize=out[2], ultra=bool(out[3]), nCoefs=nc.value)


def section(buf, ftype, sid):
    off, sz = C.c_uint64(), C.c_uint64()
    if lib.ugo_section(buf, len(buf), ftype.encode(), sid, C.byref(off), C.byref(sz)):
        raise KeyError("section %d" % sid)
    return off.value, sz.value


def groth16_prove(zkey, wtns, r, s, want_raw=False, want_timings=False):
    """r, s: ints < 2^248. Returns (proof_json_str, public_json_str[, raw][, (msm_s, fft_s)])."""
    info = zkey_info(zkey)
    proof = C.create_string_buffer(1024)
    pub = C.create_string_buffer(info["nPublic"] * 82 + 16)
    raw = C.create_string_buffer(384)
    tim = (C.c_double * 2)()
    err = C.create_string_buffer(256)
    rc = lib.ugo_groth16_prove(zkey, len(zkey), wtns, len(wtns), to_le(r), to_le(s), proof, len(proof),
                               pub, len(pub), raw, tim, err, len(err))
    if rc:
        raise RuntimeError("oracle prove failed (%d): %s" % (rc, err.value.decode()))
    res = [proof.value.decode(), pub.value.decode()]
    if want_raw:
        res.append(raw.raw)
    if want_timings:
        res.append((tim[0], tim[1]))
    return tuple(res)


def groth16_finish(zkey_header_only, sums, public_w, r, s):
    """Blinding + JSON (S11-S13) from the five MSM results as affine records (A | B1 | B2 | C | H, 384 bytes).
    zkey_header_only: a zkey holding at least sections 1, 2, 4 (see header_only_zkey); public_w: w[0..nPublic] bytes."""
    info = zkey_info(zkey_header_only)
    proof = C.create_string_buffer(1024)
    pub = C.create_string_buffer(info["nPublic"] * 82 + 16)
    err = C.create_string_buffer(256)
    rc = lib.ugo_groth16_finish(zkey_header_only, len(zkey_header_only), bytes(sums), bytes(public_w), to_le(r), to_le(s),
                                proof, len(proof), pub, len(pub), err, len(err))
    if rc:
        raise RuntimeError("oracle finish failed (%d): %s" % (rc, err.value.decode()))
    return proof.value.decode(), pub.value.decode()


def header_only_zkey(zkey):
    """sections 1 and 2 of a (possibly huge) zkey plus an empty coefficient section, as a small zkey of its own"""
    import struct
    view = memoryview(zkey).cast("B")
    out = b"zkey" + struct.pack("<II", 1, 3)
    for sid in (1, 2):
        off, sz = section(zkey, "zkey", sid)
        out += struct.pack("<IQ", sid, sz) + bytes(view[off:off + sz])
    return out + struct.pack("<IQ", 4, 4) + bytes(4)


def ultra_groth_prove(zkey, wtns, rk, r, s):
    info = zkey_info(zkey)
    proof = C.create_string_buffer(1600)
    pub = C.create_string_buffer(info["nPublic"] * 82 + 16)
    err = C.create_string_buffer(256)
    rc = lib.ugo_ultra_groth_prove(zkey, len(zkey), wtns, len(wtns), to_le(rk), to_le(r), to_le(s),
                                   proof, len(proof), pub, len(pub), err, len(err))
    if rc:
        raise RuntimeError("oracle ultragroth prove failed (%d): %s" % (rc, err.value.decode()))
    return proof.value.decode(), pub.value.decode()


def hpoly(coefs, ncoefs, wtns, nvars, domain, want_abc=False):
    h = C.create_string_buffer(domain * 32)
    abc = C.create_string_buffer(domain * 96) if want_abc else None
    if lib.ugo_hpoly(h, coefs, ncoefs, wtns, nvars, domain, abc):
        raise ValueError("coefficient index out of range")
    return (h.raw, abc.raw) if want_abc else h.raw


def ntt(data, logn, inverse=False):
    buf = C.create_string_buffer(bytes(data), len(data))
    lib.ugo_fr_ntt(buf, logn, 1 if inverse else 0)
    return buf.raw


def root_of_unity(s):
    out = C.create_string_buffer(32)
    lib.ugo_fr_root_of_unity(out, s)
    return from_le(out.raw)


def g1_msm(bases, scalars, n, naive=False):
    out = C.create_string_buffer(64)
    (lib.ugo_g1_msm_naive if naive else lib.ugo_g1_msm)(out, bytes(bases), bytes(scalars), n)
    return out.raw


def g2_msm(bases, scalars, n, naive=False):
    out = C.create_string_buffer(128)
    (lib.ugo_g2_msm_naive if naive else lib.ugo_g2_msm)(out, bytes(bases), bytes(scalars), n)
    return out.raw


def g1_mul(base, k):
    out = C.create_string_buffer(64)
    lib.ugo_g1_mul(out, bytes(base), to_le(k))
    return out.raw


def g2_mul(base, k):
    out = C.create_string_buffer(128)
    lib.ugo_g2_mul(out, bytes(base), to_le(k))
    return out.raw


def g1_add(p, q):
    out = C.create_string_buffer(64)
    lib.ugo_g1_add(out, bytes(p), bytes(q))
    return out.raw


def g2_add(p, q):
    out = C.create_string_buffer(128)
    lib.ugo_g2_add(out, bytes(p), bytes(q))
    return out.raw


def fr_dot_walk(scalars, n, seed):
    """sum scalars[i] * (seed + i) mod r  (scalars: n x 32-byte plain integers, any buffer)"""
    out = C.create_string_buffer(32)
    buf = scalars if isinstance(scalars, (bytes, bytearray)) else (C.c_char * (n * 32)).from_buffer(scalars)
    lib.ugo_fr_dot_walk(out, buf, n, seed)
    return from_le(out.raw)


def lookup_row(i, frequency, rand_mont):
    """(inv2[i], prod[i]) of compute_lookup's table (src/ultra_groth.cpp:72-79) as plain integers; rand in Montgomery form"""
    a, b = C.create_string_buffer(32), C.create_string_buffer(32)
    lib.ugo_lookup_row(a, b, i, frequency, to_le(rand_mont))
    return from_le(a.raw), from_le(b.raw)


def ref_lookup_row(i, frequency, rand_mont):
    """the same row computed by the reference's own RawFr calls (oracle/ref_shim.cpp)"""
    a, b = C.create_string_buffer(32), C.create_string_buffer(32)
    ref.ref_lookup_row(a, b, i, frequency, to_le(rand_mont))
    return from_le(a.raw), from_le(b.raw)


def derive_challenge(commit_record):
    """derive_challenge (src/ultra_groth.cpp:33-58) of a 64-byte affine G1 record; plain integer < r"""
    out = C.create_string_buffer(32)
    lib.ugo_derive_challenge(out, bytes(commit_record))
    return from_le(out.raw)


def ref_derive_challenge(commit_record):
    """the same through the reference's RawFq::toMpz / FIPS202_KECCAK_256 / RawFr::fromMpz; Montgomery -> plain here"""
    out = C.create_string_buffer(32)
    rec = bytes(commit_record)
    ref.ref_derive_challenge(out, rec[:32], rec[32:64])
    return mont_decode(out.raw, R_MOD)


def keccak256(data):
    out = C.create_string_buffer(32)
    lib.ugo_keccak256(out, bytes(data), len(data))
    return out.raw


def mont_decode(b32, mod=Q_MOD):
    """32-byte Montgomery value -> int"""
    return from_le(b32) * pow(MONT_R, -1, mod) % mod


def mont_encode(x, mod=Q_MOD):
    return to_le(x * MONT_R % mod)
