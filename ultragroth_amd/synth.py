"""Seeded synthetic Groth16 circuits in the reference's zkey / wtns formats (bench and test tooling).

Shapes follow SURVEY.md section 8(d): domainSize N, nVars = N - 1, nPublic = 1, nCoefs = 4 N (two A and
two B entries per row, signal ids from the PRNG). Base points are valid curve points
P_i = (seed + i) * G produced on the GPU by ``ug_synth_points`` (the generator walk of the survey), so an
MSM over them has a closed form in the exponent:  sum s_i P_i = (sum s_i (seed + i) mod r) * G, which the
full-size tests check. Scalar mixes: "U" uniform in [0, r); "C" circom-like (40 % in {0,1}, 20 % below
2^32, 40 % uniform).

The proofs of such circuits are not valid Groth16 proofs (there is no trapdoor behind the points); every
arithmetic step is nevertheless exactly the prover's, and results are compared bit for bit with the oracle.
"""
import struct

import numpy as np

R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617
Q_MOD = 21888242871839275222246405745257275088696311157297823662689037894645226208583

G1_GEN = (1, 2)
G2_GEN = ((10857046999023057135944570762232829481370756359578518086990519993285655852781,
           11559732032986387107991004021392285783925812861821192530917403151452391805634),
          (8495653923123431417604973247489272438418190587263600148770280649306958101930,
           4082367875863433681332203403145435568316851327593401208105741076214120093531))


def _mont(x):
    return int(x * (1 << 256) % Q_MOD).to_bytes(32, "little")


def g1_generator_record():
    return _mont(G1_GEN[0]) + _mont(G1_GEN[1])


def g2_generator_record():
    return _mont(G2_GEN[0][0]) + _mont(G2_GEN[0][1]) + _mont(G2_GEN[1][0]) + _mont(G2_GEN[1][1])


# section seeds: point i of a set is (seed + i) * G
SEEDS = {"A": 0x1000_0001, "B1": 0x2000_0003, "B2": 0x3000_0005, "C": 0x4000_0007, "H": 0x5000_0009, "VK": 0x77}


def synth_points(dev, n, seed, g2=False, out=None):
    """n zkey-format records (seed + i) * G, computed on the GPU; written into ``out`` (a writable buffer of exactly
    n records) when given"""
    import ctypes as C
    size = n * (128 if g2 else 64)
    if n == 0:
        return (C.c_char * 0)()
    if out is None:
        out = (C.c_char * size)()
    else:
        assert len(out) == size
        out = (C.c_char * size).from_buffer(out)
    gen = g2_generator_record() if g2 else g1_generator_record()
    rc = dev._L.ug_synth_points(dev._h, 1 if g2 else 0, gen, seed, n, out)
    if rc != 0:
        raise RuntimeError(dev._L.ug_last_error().decode())
    return out


def scalars(n, mix, seed):
    """n plain 32-byte scalars < r as a uint64 array [n, 4]; mix 'U' or 'C'"""
    rng = np.random.Generator(np.random.PCG64(seed))
    v = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(n, 4), dtype=np.uint64)
    # top limb uniform below r's top limb: uniform over [0, r) up to a 2^-62 sliver, always canonical
    v[:, 3] = rng.integers(0, R_MOD >> 192, size=n, dtype=np.uint64)
    if mix == "C":
        kind = rng.random(n)
        small = kind < 0.4
        v[small, 1:] = 0
        v[small, 0] = rng.integers(0, 2, size=int(small.sum()), dtype=np.uint64)
        mid = (kind >= 0.4) & (kind < 0.6)
        v[mid, 1:] = 0
        v[mid, 0] &= np.uint64(0xFFFFFFFF)
    elif mix != "U":
        raise ValueError("mix must be 'U' or 'C'")
    return v


COEF_DTYPE = np.dtype([("m", "<u4"), ("c", "<u4"), ("s", "<u4"), ("v", "<u8", (4,))], align=False)
COEF_CHUNK_ROWS = 1 << 18


def coefficients(domain, nvars, seed, out=None):
    """4 * domain packed 44-byte records: rows c = 0..domain-1, two entries in A (m=0) and two in B (m=1), generated in
    chunks of 2^18 rows (each chunk from its own PRNG stream, shuffled within itself: the reference assumes no order --
    it takes locks), so that the peak host footprint is the output plus one chunk, whatever the domain. Written into
    `out` (a writable buffer of 4 * domain * 44 bytes) when given."""
    assert COEF_DTYPE.itemsize == 44
    k = 4 * domain
    rec = np.zeros(k, dtype=COEF_DTYPE) if out is None else np.frombuffer(out, dtype=COEF_DTYPE)
    assert len(rec) == k
    def chunk(ci):
        r0 = ci * COEF_CHUNK_ROWS
        r1 = min(domain, r0 + COEF_CHUNK_ROWS)
        n = 4 * (r1 - r0)
        rng = np.random.Generator(np.random.PCG64([seed, ci]))
        part = np.zeros(n, dtype=COEF_DTYPE)
        part["m"] = np.tile(np.array([0, 0, 1, 1], dtype=np.uint32), r1 - r0)
        part["c"] = np.repeat(np.arange(r0, r1, dtype=np.uint32), 4)
        part["s"] = rng.integers(0, nvars, size=n, dtype=np.uint32)
        val = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
        val[:, 3] &= np.uint64((1 << 60) - 1)
        part["v"] = val
        rec[4 * r0:4 * r1] = part[rng.permutation(n)]
    n_chunks = (domain + COEF_CHUNK_ROWS - 1) // COEF_CHUNK_ROWS
    if n_chunks <= 2:
        for ci in range(n_chunks):
            chunk(ci)
    else:                                          # chunks are independent (own PRNG stream, own slice): numpy drops the GIL
        import os
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=max(1, min(8, len(os.sched_getaffinity(0))))) as pool:
            list(pool.map(chunk, range(n_chunks)))
    return rec


def _section(sid, payload):
    return struct.pack("<IQ", sid, len(payload)) + payload


def b_zero_mask(nvars, b_zero, seed=0x5EED0000):
    """which signals have NO B-side point (B1 and B2 at infinity) in build_circuit(.., b_zero=fraction): a real circuit's B-side
    polynomials vanish for the signals that never appear on the B side of a constraint"""
    return np.random.Generator(np.random.PCG64(seed + 7)).random(nvars) < b_zero


def build_circuit(dev, log_domain, mix="U", seed=0x5EED0000, n_public=1, g1_only=False, coefs=None, b_zero=0.0):
    """Returns (zkey, wtns_bytes, info). Sections are laid out as snarkjs does (1..10). The zkey comes back as a
    ctypes char array (buffer protocol, ``len``, accepted wherever the bindings take ``bytes``): every section is
    generated in place, so the peak host footprint is one zkey (9.4 GB at 2^24), not two.
    b_zero: fraction of the signals whose B1 and B2 points are the point at infinity (b_zero_mask; info["b_zero"])."""
    import ctypes as C
    domain = 1 << log_domain
    nvars = domain - 1
    n_c = nvars - n_public - 1
    r_le = R_MOD.to_bytes(32, "little")
    q_le = Q_MOD.to_bytes(32, "little")
    vk_g1 = bytes(synth_points(dev, 3, SEEDS["VK"]))                    # alpha1, beta1, delta1
    vk_g2 = bytes(synth_points(dev, 3, SEEDS["VK"], g2=True))           # beta2, gamma2, delta2
    header = struct.pack("<I", 32) + q_le + struct.pack("<I", 32) + r_le + struct.pack("<III", nvars, n_public, domain)
    header += vk_g1[0:64] + vk_g1[64:128] + vk_g2[0:128] + vk_g2[128:256] + vk_g1[128:192] + vk_g2[256:384]
    n_coefs = 4 * domain if coefs is None else len(coefs)               # (a caller may pass its own record array)
    sizes = [(1, 4), (2, len(header)), (3, 64 * (n_public + 1)), (4, 4 + 44 * n_coefs), (5, 64 * nvars), (6, 64 * nvars),
             (7, 128 * nvars), (8, 64 * n_c), (9, 64 * domain), (10, 0)]
    total = 12 + sum(12 + sz for _, sz in sizes)
    zkey = (C.c_char * total)()                                         # zero-filled
    view = memoryview(zkey).cast("B")
    view[0:12] = b"zkey" + struct.pack("<II", 1, len(sizes))
    off, at = 12, {}
    for sid, sz in sizes:
        view[off:off + 12] = struct.pack("<IQ", sid, sz)
        at[sid] = (off + 12, off + 12 + sz)
        off += 12 + sz
    view[at[1][0]:at[1][1]] = struct.pack("<I", 1)
    view[at[2][0]:at[2][1]] = header
    lo, hi = at[4]
    view[lo:lo + 4] = struct.pack("<I", n_coefs)
    if coefs is not None:
        if n_coefs:
            np.frombuffer(view[lo + 4:hi], dtype=coefs.dtype)[:] = coefs
        del coefs
    else:
        coefficients(domain, nvars, seed + 1, out=view[lo + 4:hi])       # generated in place, chunk by chunk
    synth_points(dev, nvars, SEEDS["A"], out=view[at[5][0]:at[5][1]])
    if not g1_only:
        # (g1_only = config 2 of BASELINE.json, G1 MSM + NTT only: B1, B2, C stay all-infinity sets of the right size)
        synth_points(dev, nvars, SEEDS["B1"], out=view[at[6][0]:at[6][1]])
        synth_points(dev, nvars, SEEDS["B2"], g2=True, out=view[at[7][0]:at[7][1]])
        synth_points(dev, n_c, SEEDS["C"], out=view[at[8][0]:at[8][1]])
        if b_zero:
            mask = b_zero_mask(nvars, b_zero, seed)
            np.frombuffer(view[at[6][0]:at[6][1]], dtype=np.uint8).reshape(nvars, 64)[mask] = 0
            np.frombuffer(view[at[7][0]:at[7][1]], dtype=np.uint8).reshape(nvars, 128)[mask] = 0
    synth_points(dev, domain, SEEDS["H"], out=view[at[9][0]:at[9][1]])
    del view
    wtns = build_witness(log_domain, mix, seed)
    info = dict(domainSize=domain, nVars=nvars, nPublic=n_public, nCoefs=n_coefs, mix=mix, seed=seed, g1_only=g1_only, b_zero=b_zero)
    return zkey, wtns, info


def zkey_header_section(dev, log_domain, n_public=1):
    """the bytes of zkey section 2 of build_circuit(log_domain)"""
    domain = 1 << log_domain
    vk_g1 = bytes(synth_points(dev, 3, SEEDS["VK"]))
    vk_g2 = bytes(synth_points(dev, 3, SEEDS["VK"], g2=True))
    header = struct.pack("<I", 32) + Q_MOD.to_bytes(32, "little") + struct.pack("<I", 32) + R_MOD.to_bytes(32, "little")
    header += struct.pack("<III", domain - 1, n_public, domain)
    return header + vk_g1[0:64] + vk_g1[64:128] + vk_g2[0:128] + vk_g2[128:256] + vk_g1[128:192] + vk_g2[256:384]


def build_circuit_slices(dev, log_domain, ranges, with_coefs=True, seed=0x5EED0000, n_public=1, g1_only=False, b_zero=0.0):
    """What ONE rank of a sharded prover needs of build_circuit(log_domain, seed=seed), without ever materialising the rest:
    (header section, coefficient records or None, (A, B1, B2, C, H) slices) for ranges = ((w0, w1), (c0, c1), (h0, h1))
    as ShardedGroth16Prover.shard_ranges reports them. The points are the same generator walk, entered at the slice."""
    import ctypes as C
    domain = 1 << log_domain
    nvars = domain - 1
    (w0, w1), (c0, c1), (h0, h1) = ranges
    coefs = None
    if with_coefs:
        coefs = (C.c_char * (4 * domain * 44))()
        coefficients(domain, nvars, seed + 1, out=coefs)

    def pts(n, name, first, g2=False):
        if g1_only and name in ("B1", "B2", "C"):
            return (C.c_char * (n * (128 if g2 else 64)))()
        return synth_points(dev, n, SEEDS[name] + first, g2=g2)
    slices = (pts(w1 - w0, "A", w0), pts(w1 - w0, "B1", w0), pts(w1 - w0, "B2", w0, g2=True), pts(c1 - c0, "C", c0), pts(h1 - h0, "H", h0))
    if b_zero and not g1_only and w1 > w0:       # (the same signals as build_circuit(.., b_zero): the mask is cut to the slice)
        mask = b_zero_mask(nvars, b_zero, seed)[w0:w1]
        np.frombuffer(memoryview(slices[1]).cast("B"), dtype=np.uint8).reshape(w1 - w0, 64)[mask] = 0
        np.frombuffer(memoryview(slices[2]).cast("B"), dtype=np.uint8).reshape(w1 - w0, 128)[mask] = 0
    return zkey_header_section(dev, log_domain, n_public), coefs, slices


def build_witness(log_domain, mix="U", seed=0x5EED0000):
    """the .wtns of build_circuit(log_domain, mix, seed) alone (the zkey does not depend on the mix)"""
    nvars = (1 << log_domain) - 1
    r_le = R_MOD.to_bytes(32, "little")
    w = scalars(nvars, mix, seed + 2)
    w[0] = (1, 0, 0, 0)
    wtns = b"wtns" + struct.pack("<II", 2, 2) + _section(1, struct.pack("<I", 32) + r_le + struct.pack("<I", nvars))
    return wtns + struct.pack("<IQ", 2, w.nbytes) + w.tobytes()


def _ultra_index_lists(log_domain, seed):
    """the generator's split of the private signals into the round set (a quarter) and the final set, and the generator
    state behind it (the .uwtns draws from the same stream)"""
    nvars = (1 << log_domain) - 1
    n_public = 2
    rng = np.random.Generator(np.random.PCG64(seed))
    priv = rng.permutation(np.arange(n_public + 1, nvars, dtype=np.uint32))
    n1 = len(priv) // 4
    return np.sort(priv[:n1]).astype("<u4"), np.sort(priv[n1:]).astype("<u4"), rng


def _ultra_header(dev, log_domain, n1, n2, n_public=2, rand_indx=2):
    domain = 1 << log_domain
    r_le = R_MOD.to_bytes(32, "little")
    q_le = Q_MOD.to_bytes(32, "little")
    vk_g1 = bytes(synth_points(dev, 4, SEEDS["VK"]))                    # alpha1, beta1, round_delta1, final_delta1
    vk_g2 = bytes(synth_points(dev, 4, SEEDS["VK"], g2=True))           # beta2, gamma2, round_delta2, final_delta2
    header = struct.pack("<I", 32) + q_le + struct.pack("<I", 32) + r_le
    header += struct.pack("<IIIIII", domain - 1, n_public, domain, n1, n2, rand_indx)
    header += vk_g1[0:64] + vk_g1[64:128] + vk_g2[0:128] + vk_g2[128:256]
    header += vk_g1[128:192] + vk_g2[256:384] + vk_g1[192:256] + vk_g2[384:512]
    return header


def ultra_info(log_domain, seed=0x5EED0005):
    """sizes of build_ultra_circuit(log_domain, seed) without building it"""
    idx1, idx2, _ = _ultra_index_lists(log_domain, seed)
    domain = 1 << log_domain
    return dict(domainSize=domain, nVars=domain - 1, nPublic=2, randIndx=2, nC1=len(idx1), nC2=len(idx2), nCoefs=4 * domain)


def build_ultra_circuit_slices(dev, log_domain, ranges, with_coefs=True, seed=0x5EED0005):
    """What ONE rank of a sharded UltraGroth prover needs of build_ultra_circuit(log_domain, seed=seed), without the rest:
    (header section, coefficient records or None, (A, B1, B2, C1, C2, H, round_indexes, final_round_indexes) slices) for
    ranges = ((w0, w1), (r0, r1), (f0, f1), (h0, h1)) as ShardedUltraGrothProver.shard_ranges reports them."""
    import ctypes as C
    domain = 1 << log_domain
    idx1, idx2, _ = _ultra_index_lists(log_domain, seed)
    (w0, w1), (r0, r1), (f0, f1), (h0, h1) = ranges
    coefs = None
    if with_coefs:
        coefs = (C.c_char * (4 * domain * 44))()
        coefficients(domain, domain - 1, seed + 1, out=coefs)
    slices = (synth_points(dev, w1 - w0, SEEDS["A"] + w0), synth_points(dev, w1 - w0, SEEDS["B1"] + w0),
              synth_points(dev, w1 - w0, SEEDS["B2"] + w0, g2=True), synth_points(dev, r1 - r0, SEEDS["C"] + r0),
              synth_points(dev, f1 - f0, SEEDS["C"] + (1 << 28) + f0), synth_points(dev, h1 - h0, SEEDS["H"] + h0),
              idx1[r0:r1].tobytes(), idx2[f0:f1].tobytes())
    return _ultra_header(dev, log_domain, len(idx1), len(idx2)), coefs, slices


def _ultra_witness(log_domain, mix, seed, lookup_log, idx2, rng, n_public=2, rand_indx=2):
    """the .uwtns of build_ultra_circuit: signals, chunks, frequencies and the two lookup index lists (src/prover.cpp:287-292);
    rng = the generator state _ultra_index_lists left"""
    nvars = (1 << log_domain) - 1
    r_le = R_MOD.to_bytes(32, "little")
    w = scalars(nvars, mix, seed + 2)
    w[0] = (1, 0, 0, 0)
    lookup = 1 << lookup_log
    n_chunks = max(nvars // 8, 1)
    chunks = rng.integers(0, lookup, size=n_chunks, dtype=np.uint32).astype("<u4")
    freq = rng.integers(0, 1 << 20, size=lookup, dtype=np.uint32).astype("<u4")
    n_push = 2 * lookup + n_chunks + 1
    # lookup signals live in the final set; the challenge itself lands on the public signal rand_indx
    targets = rng.permutation(idx2)[:min(n_push - 1, len(idx2))]
    w_idx = np.concatenate([np.array([rand_indx], dtype=np.uint32), targets]).astype("<u4")
    p_idx = np.concatenate([np.array([0], dtype=np.uint32), 1 + rng.permutation(n_push - 1)[:len(targets)].astype(np.uint32)]).astype("<u4")
    uw = [(1, struct.pack("<I", 32) + r_le + struct.pack("<I", nvars)), (2, w.tobytes()), (3, chunks.tobytes()),
          (4, freq.tobytes()), (5, w_idx.tobytes()), (6, p_idx.tobytes())]
    return b"wtns" + struct.pack("<II", 2, len(uw)) + b"".join(_section(i, p) for i, p in uw), lookup, n_chunks


def build_ultra_witness(log_domain, mix="C", seed=0x5EED0005, lookup_log=8):
    """the .uwtns of build_ultra_circuit(log_domain, mix, seed, lookup_log) alone"""
    _, idx2, rng = _ultra_index_lists(log_domain, seed)
    return _ultra_witness(log_domain, mix, seed, lookup_log, idx2, rng)[0]


def build_ultra_circuit(dev, log_domain, mix="C", seed=0x5EED0005, lookup_log=8, b_zero=0.0):
    """UltraGroth (protocol 1337) zkey + .uwtns of the same shapes: SURVEY.md section 8(d) config 5.

    nPublic = 2 with rand_indx = 2; the private signals are split into a round set C1 (a quarter of them) and a
    final set C2 (the rest); lookup table of 2^lookup_log rows, nVars/8 chunks. Section map as the reference reads
    it (src/prover.cpp:242-259, src/zkey_utils.cpp:123-163; uwtns sections 3..6 src/prover.cpp:287-292)."""
    domain = 1 << log_domain
    nvars = domain - 1
    n_public, rand_indx = 2, 2
    idx1, idx2, rng = _ultra_index_lists(log_domain, seed)
    header = _ultra_header(dev, log_domain, len(idx1), len(idx2), n_public, rand_indx)
    coefs = coefficients(domain, nvars, seed + 1)
    b1, b2 = bytes(synth_points(dev, nvars, SEEDS["B1"])), bytes(synth_points(dev, nvars, SEEDS["B2"], g2=True))
    if b_zero:                                   # (signals without a B-side point: B1 and B2 at infinity, as in build_circuit)
        mask = b_zero_mask(nvars, b_zero, seed)
        a1 = np.frombuffer(b1, dtype=np.uint8).reshape(nvars, 64).copy(); a1[mask] = 0
        a2 = np.frombuffer(b2, dtype=np.uint8).reshape(nvars, 128).copy(); a2[mask] = 0
        b1, b2 = a1.tobytes(), a2.tobytes()
    secs = [
        (1, struct.pack("<I", 1337)), (2, header), (3, bytes(64 * (n_public + 1))),
        (4, struct.pack("<I", len(coefs)) + coefs.tobytes()),
        (5, bytes(synth_points(dev, nvars, SEEDS["A"]))), (6, b1),
        (7, b2),
        (8, bytes(synth_points(dev, len(idx1), SEEDS["C"]))),                 # round points C1
        (9, bytes(synth_points(dev, len(idx2), SEEDS["C"] + (1 << 28)))),     # final points C2
        (10, idx1.tobytes()), (11, idx2.tobytes()),
        (12, bytes(synth_points(dev, domain, SEEDS["H"]))), (13, b""),
    ]
    zkey = b"zkey" + struct.pack("<II", 1, len(secs)) + b"".join(_section(i, p) for i, p in secs)
    wtns, lookup, n_chunks = _ultra_witness(log_domain, mix, seed, lookup_log, idx2, rng, n_public, rand_indx)
    info = dict(domainSize=domain, nVars=nvars, nPublic=n_public, randIndx=rand_indx, nC1=len(idx1), nC2=len(idx2),
                lookup=lookup, chunks=n_chunks)
    return zkey, wtns, info
