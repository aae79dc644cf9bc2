"""GPU parity tests proper: the HIP path, called through the C-ABI, against the CPU oracle on the same inputs."""
import hashlib
import json
import random

import pytest

import oracle as O
from oracle import pairing
from conftest import fixed_rs

pytestmark = pytest.mark.gpu


def _sec(buf, ftype, sid):
    off, sz = O.section(buf, ftype, sid)
    return buf[off:off + sz]


def _rand_elems(rng, mod, n):
    edge = [0, 1, 2, mod - 1, mod - 2, (1 << 256) % mod, (1 << 255) % mod]
    vals = edge + [rng.randrange(mod) for _ in range(n - len(edge))]
    return vals


@pytest.mark.parametrize("field,mod", [(O.FR, O.R_MOD), (O.FQ, O.Q_MOD)])
def test_field_ops_bit_exact(device, field, mod):
    import ultragroth_amd as ug
    rng = random.Random(11 + field)
    n = 4096
    a = _rand_elems(rng, mod, n)
    b = list(reversed(_rand_elems(rng, mod, n)))
    ab = b"".join(O.to_le(x) for x in a)
    bb = b"".join(O.to_le(x) for x in b)
    for op, name in ((ug.OP_MUL, "mul"), (ug.OP_ADD, "add"), (ug.OP_SUB, "sub")):
        got = device.field_op(field, op, ab, bb)
        for i in range(n):
            exp = O.f_op(name, field, a[i], b[i])
            assert O.from_le(got[32 * i:32 * i + 32]) == exp, (name, i)


@pytest.mark.parametrize("logn", [0, 1, 2, 5, 10, 11, 12, 14, 17])
def test_ntt_matches_oracle(device, logn):
    rng = random.Random(100 + logn)
    n = 1 << logn
    data = b"".join(O.to_le(rng.randrange(O.R_MOD)) for _ in range(n))
    fwd = device.ntt(data, logn, inverse=False)
    assert fwd == O.ntt(data, logn, inverse=False)
    inv = device.ntt(data, logn, inverse=True)
    assert inv == O.ntt(data, logn, inverse=True)
    assert device.ntt(fwd, logn, inverse=True) == data            # round trip


def test_ntt_root_convention(device):
    """omega_2048 = 5^((r-1)/2^11): the transform of the delta at index 1 is [omega^k]."""
    logn = 11
    one_m = O.to_le((1 << 256) % O.R_MOD)
    data = bytes(32) + one_m + bytes(32 * ((1 << logn) - 2))
    out = device.ntt(data, logn)
    w = O.mont_decode(out[32:64], O.R_MOD)
    assert w == 1120550406532664055539694724667294622065367841900378087843176726913374367458


def test_msm_g1_fixture_sections(device, zkey, wtns):
    """Raw MSMs over the reference fixture's own sections: SURVEY.md Appendix A known answers."""
    info = O.zkey_info(zkey)
    w = _sec(wtns, "wtns", 2)
    n = info["nVars"]
    got = device.msm_g1(_sec(zkey, "zkey", 5), w, n)
    assert got == O.g1_msm(_sec(zkey, "zkey", 5), w, n)
    assert O.mont_decode(got[:32]) == 21344626350637401086172020791957193447896122895487902080665101052945368184094
    assert O.mont_decode(got[32:]) == 6713982217719299172625616378045403143338774554356668389789068804132537304884
    got = device.msm_g1(_sec(zkey, "zkey", 6), w, n)                      # B1 has 3 points at infinity
    assert O.mont_decode(got[:32]) == 8193048668315265143422292719263587730195930046948232152095219516163543912381


def test_msm_g2_fixture_section(device, zkey, wtns):
    info = O.zkey_info(zkey)
    w = _sec(wtns, "wtns", 2)
    got = device.msm_g2(_sec(zkey, "zkey", 7), w, info["nVars"])
    assert got == O.g2_msm(_sec(zkey, "zkey", 7), w, info["nVars"])
    assert O.mont_decode(got[:32]) == 7062321476789293854033358658370016416373968873586528291130727575191400769869


def _scalar_mix(rng, n):
    """edge-heavy scalars: zeros, ones, small, r-1, values >= r, all-ones windows, uniform"""
    out = []
    for i in range(n):
        k = rng.randrange(10)
        if k == 0: v = 0
        elif k == 1: v = 1
        elif k == 2: v = rng.randrange(1 << 32)
        elif k == 3: v = O.R_MOD - 1 - rng.randrange(3)
        elif k == 4: v = O.R_MOD + rng.randrange(1 << 200)        # malformed witness value >= r
        elif k == 5: v = (1 << 254) - 1
        else: v = rng.randrange(O.R_MOD)
        out.append(v)
    return out


@pytest.mark.parametrize("n", [0, 1, 2, 63, 500, 1000])
def test_msm_g1_edge_scalars(device, zkey, n):
    rng = random.Random(7 * n + 1)
    pts = _sec(zkey, "zkey", 9)[:64 * n]                     # H points
    sc = b"".join(O.to_le(v) for v in _scalar_mix(rng, n))
    assert device.msm_g1(pts, sc, n) == O.g1_msm(pts, sc, n)


def test_msm_all_zero_and_all_one_scalars(device, zkey):
    """no entry at all (every digit is zero: the sort's drop mode leaves an empty schedule) and one entry per scalar, all in one bucket"""
    n = 700
    pts = _sec(zkey, "zkey", 5)[:64 * n]
    assert device.msm_g1(pts, bytes(32 * n), n) == bytes(64)
    ones = O.to_le(1) * n
    assert device.msm_g1(pts, ones, n) == O.g1_msm(pts, ones, n)
    pts2 = _sec(zkey, "zkey", 7)[:128 * 200]
    assert device.msm_g2(pts2, bytes(32 * 200), 200) == bytes(128)


def test_msm_heavy_bucket_and_duplicates(device, zkey):
    """many equal scalars on equal points: one giant bucket (block-parallel path) and the doubling branch"""
    rng = random.Random(5)
    base = _sec(zkey, "zkey", 5)
    n = 6000
    pts = b"".join(base[64 * (i % 7 + 2):64 * (i % 7 + 3)] for i in range(n))     # 7 distinct points repeated
    sc = b"".join(O.to_le(1 if i % 3 else rng.randrange(4)) for i in range(n))
    assert device.msm_g1(pts, sc, n) == O.g1_msm(pts, sc, n)
    # P and -P with the same scalar cancel
    p = base[64 * 4:64 * 5]
    negp = p[:32] + O.to_le((O.Q_MOD - O.from_le(p[32:])) % O.Q_MOD)
    assert device.msm_g1(p + negp, O.to_le(12345) * 2, 2) == bytes(64)


@pytest.mark.parametrize("n", [1, 300])
def test_msm_g2_edge_scalars(device, zkey, n):
    rng = random.Random(3 * n + 2)
    pts = _sec(zkey, "zkey", 7)[:128 * n]
    sc = b"".join(O.to_le(v) for v in _scalar_mix(rng, n))
    assert device.msm_g2(pts, sc, n) == O.g2_msm(pts, sc, n)


@pytest.mark.parametrize("n,c", [(1, 16), (5, 17), (1000, 16), (1000, 24)])
def test_msm_window_tables_g1(device, zkey, n, c):
    """fixed-base window tables (ug_bases_precompute + ug_schedule_build_tables): same sums as the oracle for edge
    scalars, with points at infinity in the set (B1 of the fixture has three) and a classic schedule on the same bases"""
    rng = random.Random(11 * n + c)
    pts = _sec(zkey, "zkey", 6)[:64 * n]
    sc = b"".join(O.to_le(v) for v in _scalar_mix(rng, n))
    exp = O.g1_msm(pts, sc, n)
    assert device.msm_g1(pts, sc, n, table_c=c) == exp
    b = device.bases(pts, n, table_c=c)
    v = device.dvec(n, sc)
    assert device.msm(b, device.schedule(v, 0, n)) == exp                  # classic schedule reads table 0 only
    # a sub-range of the scalars against the table set: indices line up through the schedule's first index
    if n >= 1000:
        lo, cnt = 137, 700
        exp_sub = O.g1_msm(pts[64 * lo:64 * (lo + cnt)], sc[32 * lo:32 * (lo + cnt)], cnt)
        assert device.msm(b, device.schedule(v, lo, cnt, table_c=c)) == exp_sub
        # index shift, as the C section uses it: scalar i multiplies point i - 3
        exp_shift = O.g1_msm(pts[:64 * (n - 3)], sc[32 * 3:], n - 3)
        assert device.msm(b, device.schedule(v, 0, n, table_c=c), index_shift=3) == exp_shift


@pytest.mark.parametrize("n,c", [(1, 16), (300, 18)])
def test_msm_window_tables_g2(device, zkey, n, c):
    rng = random.Random(13 * n + c)
    pts = _sec(zkey, "zkey", 7)[:128 * n]
    sc = b"".join(O.to_le(v) for v in _scalar_mix(rng, n))
    assert device.msm_g2(pts, sc, n, table_c=c) == O.g2_msm(pts, sc, n)


def _class_tilings(rng, q_log, parts, n):
    """`parts` residue ranges that tile [0, 2^q_log) and as many scalar ranges that tile [0, n), both uneven"""
    Q = 1 << q_log
    sizes = [Q // parts + (1 if k < Q % parts else 0) for k in range(parts)]
    for _ in range(4 * parts):                      # uneven, every range within 1 .. 31 residues (a result block holds 63 points)
        a, b = rng.randrange(parts), rng.randrange(parts)
        d = rng.randrange(0, 8)
        if a != b and sizes[a] - d >= 1 and sizes[b] + d <= 31:
            sizes[a] -= d; sizes[b] += d
    cuts = [sum(sizes[:k]) for k in range(1, parts)]
    res = list(zip([0] + cuts, cuts + [Q]))
    scuts = sorted(rng.randrange(0, n + 1) for _ in range(parts - 1))
    sp = list(zip([0] + scuts, scuts + [n]))
    return res, sp


@pytest.mark.parametrize("table_c,q_log,parts,specials", [(16, 3, 8, 16), (16, 7, 5, 16), (18, 2, 3, 0), (17, 4, 16, 64), (20, 6, 4, 1)])
def test_msm_bucket_classes_add_up(device, zkey, table_c, q_log, parts, specials):
    """ug_schedule_set_classes (bucket-range sharding of a many-device prover): schedules over the SAME
    scalars and bases that own disjoint residue ranges of the bucket ids -- and disjoint scalar ranges of the special (lowest)
    buckets -- give products that add up to the plain product == the oracle: G1, G2 and the interleaved A | B1 | C group,
    several table widths, edge scalars, and the skewed cases the specials exist for (every scalar 1; every scalar
    r - 1, which puts ALL entries of a window into one bucket of one class)"""
    info = O.zkey_info(zkey)
    n, shift = 1000, info["nPublic"] + 1
    rng = random.Random(1000 * q_log + parts + table_c)
    A, B1, B2, Cs = (_sec(zkey, "zkey", k) for k in (5, 6, 7, 8))
    mixes = [_scalar_mix(rng, n), [1] * n, [O.R_MOD - 1] * n, [rng.choice([0, 1, 2, 3, 255, 2**32 - 1, O.R_MOD - 2]) for _ in range(n)]]
    g1 = device.bases(A[:64 * n], n, table_c=table_c)
    g2 = device.bases(B2[:128 * n], n, g2=True, table_c=table_c)
    grp = device.bases_group([(A[:64 * n], n, 0), (B1[:64 * n], n, 0), (Cs[:64 * (n - shift)], n - shift, shift)], 0, n, table_c=table_c)
    for vals in mixes:
        sc = b"".join(O.to_le(v) for v in vals)
        v = device.dvec(n, sc)
        exp1, exp2 = O.g1_msm(A[:64 * n], sc, n), O.g2_msm(B2[:128 * n], sc, n)
        expg = [exp1, O.g1_msm(B1[:64 * n], sc, n), O.g1_msm(Cs[:64 * (n - shift)], sc[32 * shift:], n - shift)]
        res, sp = _class_tilings(rng, q_log, parts, n)
        sum1, sum2, sumg = bytes(64), bytes(128), [bytes(64)] * 3
        for (r0, r1), (s0, s1) in zip(res, sp):
            sch = device.schedule(v, 0, n, table_c=table_c, classes=(q_log, r0, r1 - r0, specials, s0, s1 - s0))
            sum1 = O.g1_add(sum1, device.msm(g1, sch))
            sum2 = O.g2_add(sum2, device.msm(g2, sch, g2=True))
            sumg = [O.g1_add(a, b) for a, b in zip(sumg, device.msm_group(grp, sch))]
        assert sum1 == exp1 and sum2 == exp2 and sumg == expg, (vals[:4], res, sp)
    # a sub-range of the scalars, as a rank of a P x B layout holds it (the special range given in the vector's indices)
    lo, cnt = 100, 800
    sc = b"".join(O.to_le(x) for x in mixes[0])
    v = device.dvec(n, sc)
    b = device.bases(A[64 * lo:64 * (lo + cnt)], cnt, global_first=lo, table_c=table_c)
    exp = O.g1_msm(A[64 * lo:64 * (lo + cnt)], sc[32 * lo:32 * (lo + cnt)], cnt)
    res, sp = _class_tilings(rng, q_log, parts, cnt)
    acc = bytes(64)
    for (r0, r1), (s0, s1) in zip(res, sp):
        acc = O.g1_add(acc, device.msm(b, device.schedule(v, lo, cnt, table_c=table_c, classes=(q_log, r0, r1 - r0, specials, lo + s0, s1 - s0))))
    assert acc == exp


def test_msm_bucket_classes_with_classic_windows(device):
    """bucket classes on a CLASSIC schedule (no window tables: one bucket set per window and owned residue, the special buckets
    per window, Horner over the windows on the host) -- possible when the windows are few enough for the result block: 2^16
    scalars take ~20 windows; two ranks with one residue each, and four with the residues of Q = 4, G1 and G2"""
    import ultragroth_amd as ug
    from ultragroth_amd import synth
    zk, wt, info = synth.build_circuit(device, 16, mix="C")
    n = info["nVars"]
    A, B2, wb = bytes(_sec(zk, "zkey", 5)), bytes(_sec(zk, "zkey", 7)), bytes(_sec(wt, "wtns", 2))
    g1, g2 = device.bases(A, n), device.bases(B2, n, g2=True)
    v = device.dvec(n, wb)
    plain = device.schedule(v, 0, n)
    exp1, exp2 = device.msm(g1, plain), device.msm(g2, plain, g2=True)
    assert exp1 == O.g1_msm(A, wb, n)
    ran = 0
    for q_log, specials in ((1, 0), (2, 0), (1, 16), (2, 3)):
        Q = 1 << q_log
        s1, s2 = bytes(64), bytes(128)
        try:
            for r in range(Q):
                sch = device.schedule(v, 0, n, classes=(q_log, r, 1, specials, n * r // Q, n * (r + 1) // Q - n * r // Q))
                s1 = O.g1_add(s1, device.msm(g1, sch))
                s2 = O.g2_add(s2, device.msm(g2, sch, g2=True))
        except ug.DeviceError as e:                     # (the cost-model window brings more sets than a result block holds: refused)
            assert "too many bucket sets" in str(e) or "window too narrow" in str(e)
            continue
        assert s1 == exp1 and s2 == exp2, (q_log, specials)
        ran += 1
    assert ran >= 1


def test_msm_bucket_classes_errors(device, zkey):
    import ultragroth_amd as ug
    v = device.dvec(10, b"".join(O.to_le(i + 1) for i in range(10)))
    with pytest.raises(ug.DeviceError, match="residues outside"):
        device.schedule(v, 0, 10, table_c=16, classes=(3, 6, 3, 0, 0, 10))
    with pytest.raises(ug.DeviceError, match="more than 64 special"):
        device.schedule(v, 0, 10, table_c=16, classes=(3, 0, 1, 65, 0, 10))
    with pytest.raises(ug.DeviceError, match="window too narrow"):
        device.schedule(v, 0, 10, classes=(8, 0, 1, 0, 0, 10))          # the classic window of ten scalars is a few bits wide
    with pytest.raises(ug.DeviceError, match="too many bucket sets"):
        device.schedule(v, 0, 10, table_c=24, classes=(7, 0, 100, 0, 0, 10))
    # (classic windows bring one set per window and owned residue: the classes are made for the one-set table form)
    w = device.dvec(4000, b"".join(O.to_le(3 * i + 1) for i in range(4000)))
    with pytest.raises(ug.DeviceError, match="too many bucket sets|window too narrow"):
        device.schedule(w, 0, 4000, classes=(2, 0, 2, 0, 0, 4000))
    # classes switched off again: the plain product
    pts = _sec(zkey, "zkey", 5)[:640]
    b = device.bases(pts, 10, table_c=16)
    assert device.msm(b, device.schedule(v, 0, 10, table_c=16, classes=(0, 0, 0, 0, 0, 0))) == O.g1_msm(pts, b"".join(O.to_le(i + 1) for i in range(10)), 10)


def test_schedule_refuses_more_than_2_30_entries(device):
    """a look-back status word of the radix partition carries a 30-bit pair count (one bin may hold every pair: equal scalars), so a
    schedule of more than 2^30 (scalar, window) entries is refused before any kernel runs -- the provers stay below it (ranges of
    at most 2^26 scalars x 16 windows; larger ranges are proved in pieces)"""
    import ultragroth_amd as ug
    n = (1 << 30) // 12 + 4096                          # 12 windows of 22 bits: just above the limit
    v = device.dvec(n)                                  # (2.9 GB of device memory, never read)
    with pytest.raises(ug.DeviceError, match="exceeds 2\\^30 entries"):
        device.schedule(v, 0, n, table_c=22)
    device.schedule(v, 0, 1 << 10, table_c=22)          # the same vector, a legal range: fine


def test_msm_window_tables_errors(device, zkey):
    pts = _sec(zkey, "zkey", 5)[:64 * 10]
    sc = b"".join(O.to_le(i + 1) for i in range(10))
    import ultragroth_amd as ug
    with pytest.raises(ug.DeviceError, match="outside \\[16, 24\\]"):
        device.bases(pts, 10, table_c=15)
    b = device.bases(pts, 10)
    v = device.dvec(10, sc)
    with pytest.raises(ug.DeviceError, match="hold no tables"):
        device.msm(b, device.schedule(v, 0, 10, table_c=16))
    b16 = device.bases(pts, 10, table_c=16)
    with pytest.raises(ug.DeviceError, match="tables of width 16"):
        device.msm(b16, device.schedule(v, 0, 10, table_c=17))
    assert 16 <= device.table_window(1 << 24) <= 24


def test_lookup_completion_last_write_wins(device):
    """ug_dvec_apply_lookup == the reference's loop `wtns[w_idx[i]] = push[p_idx[i]]` over the explicit push vector
    [rand | inv2[chunks[j]] | inv2 | prod], with repeated targets (the last write must win) and every region of push"""
    import numpy as np
    import ultragroth_amd as ug
    rng = np.random.Generator(np.random.PCG64(21))
    n_dst, L, n_chunks, n = 5000, 16, 300, 4000
    table = rng.integers(0, 256, size=(1 + 2 * L, 32), dtype=np.uint8)
    chunks = rng.integers(0, L, size=n_chunks, dtype=np.uint32)
    push = np.concatenate([table[0:1], table[1 + chunks], table[1:]])
    assert len(push) == 1 + n_chunks + 2 * L
    w_idx = rng.integers(0, 600, size=n, dtype=np.uint32)                 # heavy repetition
    w_idx[::7] = rng.integers(0, n_dst, size=len(w_idx[::7]), dtype=np.uint32)
    p_idx = rng.integers(0, len(push), size=n, dtype=np.uint32)
    p_idx[:3] = (0, n_chunks, len(push) - 1)
    start = rng.integers(0, 256, size=(n_dst, 32), dtype=np.uint8)
    exp = start.copy()
    for i in range(n):
        exp[w_idx[i]] = push[p_idx[i]]
    v = device.dvec(n_dst, start.tobytes())
    device.apply_lookup(v, w_idx, p_idx, chunks, table.tobytes(), L)
    assert device.download(v, 0, n_dst) == exp.tobytes()
    device.apply_lookup(v, w_idx, p_idx, chunks, table.tobytes(), L)       # scratch left clean: same result again
    assert device.download(v, 0, n_dst) == exp.tobytes()
    bad = p_idx.copy(); bad[5] = len(push)
    with pytest.raises(ug.DeviceError, match="lookup index out of range"):
        device.apply_lookup(v, w_idx, bad, chunks, table.tobytes(), L)
    badc = chunks.copy(); badc[0] = L
    with pytest.raises(ug.DeviceError, match="chunk index outside"):
        device.apply_lookup(v, w_idx, p_idx, badc, table.tobytes(), L)
    assert device.download(v, 0, n_dst) == exp.tobytes()


def test_msm_batch_equals_single_products(device, zkey, wtns):
    """ug_msm_batch queues A, B1, B2, C (G1 and G2 mixed, C with its index shift) back to back and synchronises once:
    the same records as one call each, with and without window tables, and for an empty batch"""
    info = O.zkey_info(zkey)
    n, shift = info["nVars"], info["nPublic"] + 1
    w = device.dvec(n, _sec(wtns, "wtns", 2))
    for c in (0, 16):
        a = device.bases(_sec(zkey, "zkey", 5), n, table_c=c)
        b1 = device.bases(_sec(zkey, "zkey", 6), n, table_c=c)
        b2 = device.bases(_sec(zkey, "zkey", 7), n, g2=True, table_c=c)
        cc = device.bases(_sec(zkey, "zkey", 8), n - shift, table_c=c)
        s = device.schedule(w, 0, n, table_c=c)
        got = device.msm_batch([(a, False), (b1, False), (b2, True), (cc, False)], s, [0, 0, 0, shift])
        assert got == [device.msm(a, s), device.msm(b1, s), device.msm(b2, s, g2=True), device.msm(cc, s, index_shift=shift)]
        assert got[0] == O.g1_msm(_sec(zkey, "zkey", 5), _sec(wtns, "wtns", 2), n)
        assert device.msm_batch([], s) == []
        assert device.msm_batch([(b2, True)], s) == [got[2]]


def test_msm_group_equals_single_products(device, zkey, wtns):
    """ug_bases_create_group_g1 + ug_msm_group_enqueue: A, B1 and C of the fixture as ONE interleaved array of three-point
    records (C with its index shift folded into the slots: the first nPublic + 1 slots hold infinity for it), one
    accumulation launch with three accumulators per lane == the three separate products == the oracle; with and without
    window tables, on sub-ranges (a sharded rank's slice, C cut by the slice), and the two-member form (UltraGroth's A, B1)"""
    import ultragroth_amd as ug
    info = O.zkey_info(zkey)
    n, shift = info["nVars"], info["nPublic"] + 1
    A, B1, Cs = _sec(zkey, "zkey", 5), _sec(zkey, "zkey", 6), _sec(zkey, "zkey", 8)
    wb = _sec(wtns, "wtns", 2)
    w = device.dvec(n, wb)
    exp = [O.g1_msm(A, wb, n), O.g1_msm(B1, wb, n), O.g1_msm(Cs, wb[shift * 32:], n - shift)]
    for c in (0, 16, 18):
        g = device.bases_group([(A, n, 0), (B1, n, 0), (Cs, n - shift, shift)], 0, n, table_c=c)
        s = device.schedule(w, 0, n, table_c=c)
        assert device.msm_group(g, s) == exp, c
        assert device.msm_group(g, s) == exp                               # the workspace is reused
        g2 = device.bases_group([(A, n, 0), (B1, n, 0)], 0, n, table_c=c)
        assert device.msm_group(g2, s) == exp[:2]
        with pytest.raises(ug.DeviceError, match="ug_msm_group_enqueue"):
            device.msm(g, s)
    # a slice [lo, hi) of the scalars, as a sharded rank holds it: the group starts at lo, C's slice is cut by the shift
    for lo, hi in ((0, 1), (1, 500), (500, n), (0, 2), (3, 3)):
        c_lo, c_hi = max(lo - shift, 0), max(hi - shift, 0)
        g = device.bases_group([(A[lo * 64:hi * 64], hi - lo, lo), (B1[lo * 64:hi * 64], hi - lo, lo),
                                (Cs[c_lo * 64:c_hi * 64], c_hi - c_lo, c_lo + shift)], lo, hi - lo)
        s = device.schedule(w, lo, hi - lo)
        sl = wb[lo * 32:hi * 32]
        assert device.msm_group(g, s) == [O.g1_msm(A[lo * 64:hi * 64], sl, hi - lo), O.g1_msm(B1[lo * 64:hi * 64], sl, hi - lo),
                                          O.g1_msm(Cs[c_lo * 64:c_hi * 64], wb[(c_lo + shift) * 32:(c_hi + shift) * 32], c_hi - c_lo)], (lo, hi)
    with pytest.raises(ug.DeviceError, match="outside the group"):
        device.bases_group([(A, n, 0), (B1, n, 1)], 0, n)
    with pytest.raises(ug.DeviceError, match="2 or 3 members"):
        device.bases_group([(A, n, 0)], 0, n)


def test_hpoly_matches_oracle_and_known_answers(device, zkey, wtns):
    info = O.zkey_info(zkey)
    coefs = _sec(zkey, "zkey", 4)[4:]
    w = _sec(wtns, "wtns", 2)
    hp = device.hpoly(coefs, info["nCoefs"], info["domainSize"], info["nVars"])
    hv = hp.run(device.dvec(info["nVars"], w))
    h = device.download(hv, 0, info["domainSize"])
    h_exp, abc = O.hpoly(coefs, info["nCoefs"], w, info["nVars"], info["domainSize"], want_abc=True)
    assert h == h_exp
    assert hashlib.sha256(h).hexdigest() == "44ca2358066ca82cffac3cc7f58f163f8cd806059850dea84e12cde01ff3b872"
    a, b, c = hp.debug_abc()
    n = info["domainSize"] * 32
    assert a == abc[:n] and b == abc[n:2 * n] and c == abc[2 * n:]
    # the block is queued without a host wait: a caller that never asks for timings may queue it many times over
    # (the event pairs behind the kernel statistics are recycled as launches finish), and results stay right
    wv = device.dvec(info["nVars"], w)
    device.kernel_stats(which=2, reset=True)              # (the context is shared with other tests' transforms)
    outs = [hp.run(wv) for _ in range(40)]
    assert device.download(outs[-1], 0, info["domainSize"]) == h_exp and device.download(outs[0], 0, info["domainSize"]) == h_exp
    ms, launches, points = device.kernel_stats(which=2)
    assert launches > 0 and points == launches * info["domainSize"]


@pytest.mark.parametrize("domain", [1, 2, 64])
def test_hpoly_degenerate_matrices(device, zkey, domain):
    """no coefficients at all (h = 0), a single row, records that repeat the same (matrix, row, signal) -- the reference
    adds every record it meets -- and a matrix index that is neither 0 nor 1 rejected at create"""
    import struct
    import ultragroth_amd as ug
    nvars = 5
    w = b"".join(O.to_le(v) for v in (1, 7, O.R_MOD - 2, 12345678901234567890, 3))
    wv = device.dvec(nvars, w)
    hp = device.hpoly(b"", 0, domain, nvars)
    assert device.download(hp.run(wv), 0, domain) == bytes(32 * domain)
    val = O.to_le(5 * pow(2, 512, O.R_MOD) % O.R_MOD)                          # the zkey stores coefficients times R^2
    recs = []
    for m, c, s_ in ((0, 0, 1), (0, 0, 1), (1, 0, 2), (1, domain - 1, 4), (0, domain - 1, 3), (1, 0, 2)):
        recs.append(struct.pack("<III", m, c, s_) + val)
    coefs = b"".join(recs)
    hp = device.hpoly(coefs, len(recs), domain, nvars)
    assert device.download(hp.run(wv), 0, domain) == O.hpoly(coefs, len(recs), w, nvars, domain)
    with pytest.raises(ug.DeviceError, match="coefficient record out of range"):
        device.hpoly(struct.pack("<III", 2, 0, 0) + val, 1, domain, nvars)
    with pytest.raises(ug.DeviceError, match="coefficient record out of range"):
        device.hpoly(struct.pack("<III", 0, domain, 0) + val, 1, domain, nvars)
    with pytest.raises(ug.DeviceError, match="coefficient record out of range"):
        device.hpoly(struct.pack("<III", 1, 0, nvars) + val, 1, domain, nvars)


def test_groth16_proof_bit_exact_and_valid(zkey, wtns, vkey):
    """Full prove through the reference's C API with the Appendix A blinding: byte-identical to the oracle and to
    the committed known answer, and accepted by the reference's acceptance test (pairing check)."""
    import ultragroth_amd as ug
    r, s = fixed_rs()
    ug.set_test_blinding(r + s)
    try:
        with ug.Groth16Prover(zkey) as p:
            proof, pub = p.prove(wtns)
            proof2, pub2 = p.prove(wtns)                  # the handle is reusable
    finally:
        ug.set_test_blinding(b"")
    exp_proof, exp_pub = O.groth16_prove(zkey, wtns, int.from_bytes(r, "little"), int.from_bytes(s, "little"))
    assert proof == exp_proof and pub == exp_pub
    assert proof2 == proof and pub2 == pub
    assert hashlib.sha256(proof.encode()).hexdigest() == "11767e6c2a13edf5c87e282bc652346b4e62e0faa4156275edbef96c25762592"
    assert pub == '["7713112592372404476342535432037683616424591277138491596200192981572885523208"]'
    assert pairing.groth16_verify(vkey, pub, proof)
    tampered = json.loads(pub)
    tampered[0] = str(int(tampered[0]) - 1)
    assert not pairing.groth16_verify(vkey, tampered, proof)


def test_groth16_random_blinding_still_verifies(zkey, wtns, vkey):
    import ultragroth_amd as ug
    proof, pub = ug.groth16_prover(zkey, wtns)
    assert pairing.groth16_verify(vkey, pub, proof)


def test_one_prover_object_from_several_threads(zkey, wtns, vkey):
    """the reference's prover object keeps no per-proof state, so callers may share it between threads; here the calls
    take turns on the object's device buffers: every proof (fresh OS-entropy blinding each) verifies, all differ"""
    import threading
    import ultragroth_amd as ug
    results, errors = [], []
    with ug.Groth16Prover(zkey) as p:
        def work():
            try:
                for _ in range(2):
                    results.append(p.prove(wtns))
            except Exception as e:             # noqa: BLE001 - reported below
                errors.append(e)
        threads = [threading.Thread(target=work) for _ in range(3)]
        for t in threads: t.start()
        for t in threads: t.join()
    assert not errors
    assert len(results) == 6 and len({pr for pr, _ in results}) == 6
    assert all(pub == results[0][1] for _, pub in results)
    for proof, pub in results[:3]:
        assert pairing.groth16_verify(vkey, pub, proof)
