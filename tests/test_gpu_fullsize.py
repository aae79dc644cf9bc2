"""Whole-proof bit-exactness at BASELINE.json's full sizes (configs[2] = 2^24, configs[3] = 2^26 constraints).

A CPU Pippenger at these sizes takes minutes to hours, so the expected proof is assembled without one
(oracle/closed_form.py): the H polynomial by the oracle at full size, the five MSMs in the exponent (the synthetic
base points are the generator walk P_i = (seed + i) G, so sum s_i P_i = (sum s_i (seed + i) mod r) G), blinding and
JSON by the oracle's restatement of src/groth16.cpp:158-250. The GPU side is the product path end to end: a CREATED
prover (window tables at 2^24; classic windows at 2^26, where the tables do not fit one GPU), the witness handed over in
host memory, groth16_prover_prove. Everything the bench line times is therefore compared byte for byte: CSR sort and
mat-vec with 2^26 / 2^28 coefficient records, the NTT chains with the fused twist, h = a.b - c, the batched
A/B1/B2/C products (G2 at the full size) and the H product.

Sizes: UG_FULL_LOG (default 24) and UG_HUGE_LOG (default 26; 0 skips that test). Progress goes to
gpurun_out/fullsize_progress.log (pytest captures stdout; a long silent test would look hung).
"""
import os
import time

import pytest

import oracle as O
from oracle import closed_form
from conftest import fixed_rs, ROOT

pytestmark = pytest.mark.gpu

FULL_LOG = int(os.environ.get("UG_FULL_LOG", "24"))
HUGE_LOG = int(os.environ.get("UG_HUGE_LOG", "26"))


def _progress(msg):
    d = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "fullsize_progress.log"), "a") as f:
            f.write("%s %s\n" % (time.strftime("%H:%M:%S"), msg))
    except OSError:
        pass


_EXPECTED = {}


def _expected(zkey, wtns, log_domain, mix=None):
    """mix: cache key -- the 2^24 tests share a circuit, so tests that prove the same witness share the expected proof
    (each costs an oracle H polynomial at full size)"""
    from ultragroth_amd import synth
    if mix is not None and (log_domain, mix) in _EXPECTED:
        return _EXPECTED[(log_domain, mix)]
    r, s = fixed_rs()
    exp = closed_form.groth16_expected(zkey, wtns, synth.SEEDS, synth.g1_generator_record(), synth.g2_generator_record(),
                                        int.from_bytes(r, "little"), int.from_bytes(s, "little"),
                                        progress=lambda m: _progress("2^%d: %s" % (log_domain, m)))
    if mix is not None:
        _EXPECTED[(log_domain, mix)] = exp
    return exp


def _prove(prover, wtns):
    import ultragroth_amd as ug
    r, s = fixed_rs()
    ug.set_test_blinding(r + s)
    try:
        return prover.prove(wtns)
    finally:
        ug.set_test_blinding(b"")


@pytest.fixture(scope="module")
def full_zkey(device):
    """the 2^FULL_LOG circuit of the bench (the zkey does not depend on the scalar mix)"""
    from ultragroth_amd import synth
    _progress("2^%d: building the circuit" % FULL_LOG)
    zkey, _, info = synth.build_circuit(device, FULL_LOG, mix="C")
    return zkey, info


@pytest.fixture(scope="module")
def full_prover(full_zkey):
    import ultragroth_amd as ug
    _progress("2^%d: creating the prover (window tables)" % FULL_LOG)
    p = ug.Groth16Prover(full_zkey[0])
    yield p
    p.close()


def test_closed_form_equals_the_oracle_prover(device):
    """the closed-form assembly itself, against the oracle's real Pippenger where that is cheap (2^13)"""
    from ultragroth_amd import synth
    r, s = fixed_rs()
    ri, si = int.from_bytes(r, "little"), int.from_bytes(s, "little")
    for mix, g1_only in (("U", False), ("C", False), ("U", True)):
        zkey, wtns, info = synth.build_circuit(device, 13, mix=mix, g1_only=g1_only)
        exp = O.groth16_prove(zkey, wtns, ri, si)
        got = closed_form.groth16_expected(zkey, wtns, synth.SEEDS, synth.g1_generator_record(), synth.g2_generator_record(),
                                           ri, si, g1_only=g1_only)
        assert got == (exp[0], exp[1])


def test_whole_proof_at_configs2_size_uniform(full_zkey, request):
    """BASELINE.json configs[2]: 2^24 constraints, uniform scalars, created prover with window tables -- the bench's
    default workload (a second witness on the same prover object follows in the circom-like test below; two different
    witnesses through one prover are also compared at 2^20, test_created_prover_at_2_20_bit_exact).
    COLD START at the full size (round 5): groth16_prover_create returns once the zkey is resident; the first proof runs at
    once -- on the classic windows, BESIDE the table kernels -- and the one after the tables are finished on the tables: both
    byte for byte the expected proof."""
    from ultragroth_amd import synth
    zkey, info = full_zkey
    wtns = synth.build_witness(FULL_LOG, "U")
    exp = _expected(zkey, wtns, FULL_LOG, "U")                  # (first: the oracle's H polynomial takes longer than the table build)
    t0 = time.perf_counter()
    full_prover = request.getfixturevalue("full_prover")        # groth16_prover_create
    created = time.perf_counter() - t0
    ready_at_create = full_prover.tables_ready()
    _progress("2^%d U: proving at once (create %.2f s, tables ready at create: %s)" % (FULL_LOG, created, ready_at_create))
    first = _prove(full_prover, wtns)
    t_first = time.perf_counter() - t0
    assert first == exp
    full_prover.tables_ready(wait=True)
    t_tables = time.perf_counter() - t0
    _progress("2^%d U: first proof %.2f s after the start of create, tables in use after %.2f s; proving on the tables" % (FULL_LOG, t_first, t_tables))
    assert full_prover.tables_ready()
    assert _prove(full_prover, wtns) == exp
    if os.environ.get("ULTRAGROTH_TABLES_BG", "1") != "0" and FULL_LOG >= 22:
        assert not ready_at_create                              # (the build takes seconds at this size: create did not wait for it)


def test_whole_proof_at_configs2_size_circom_like(full_zkey, full_prover):
    """the SAME prover object with another witness, the circom-like one (40 % zeros and ones: million-entry buckets, the
    heavy-bucket path with window tables): nothing may leak from the previous proof"""
    from ultragroth_amd import synth
    zkey, info = full_zkey
    wtns = synth.build_witness(FULL_LOG, "C")
    _progress("2^%d C: proving" % FULL_LOG)
    assert _prove(full_prover, wtns) == _expected(zkey, wtns, FULL_LOG, "C")


def test_piecewise_ranges_at_full_size(full_zkey, full_prover, monkeypatch):
    """ranges above ULTRAGROTH_MAX_RANGE scalars are proved in pieces whose partial sums are added (the path the
    reference's largest legal domain, 2^27, takes): forced here at the full size with 2^22-scalar pieces, which also
    run the classic windows (no tables) at this size"""
    import ultragroth_amd as ug
    from ultragroth_amd import synth
    zkey, info = full_zkey
    full_prover.close()                                         # one 2^24 prover with tables at a time is enough
    monkeypatch.setenv("ULTRAGROTH_MAX_RANGE", str(1 << (FULL_LOG - 2)))
    wtns = synth.build_witness(FULL_LOG, "C")
    _progress("2^%d C piecewise: creating + proving" % FULL_LOG)
    with ug.Groth16Prover(zkey) as p:
        assert _prove(p, wtns) == _expected(zkey, wtns, FULL_LOG, "C")


def test_reference_api_on_eight_ranks_at_configs2_size(full_zkey, monkeypatch):
    """The path the reference's own CLI / API takes on a node (src/main_prover.cpp:55, groth16_prover_create / _prove with
    ULTRAGROTH_DEVICES): eight ranks -- here on the one device -- behind groth16_prover_prove at the full 2^24 size: one sharded
    rank object per listed device, the witness in parts, chains on ranks 0-2 first, peer copies of the evaluation slices
    (ug_dvec_copy: the same-device branch here), partial blocks added on the host; byte for byte the expected proof, twice
    (circom-like, then uniform scalars) on one object. The step time goes to gpurun_out/fullsize_progress.log (eight ranks share
    one GPU: a correctness rehearsal, not a speed)."""
    import ultragroth_amd as ug
    from ultragroth_amd import synth
    zkey, info = full_zkey
    monkeypatch.setenv("ULTRAGROTH_DEVICES", "0,0,0,0,0,0,0,0")
    _progress("2^%d x8 behind the reference API: creating" % FULL_LOG)
    t0 = time.perf_counter()
    with ug.Groth16Prover(zkey) as p:
        _progress("2^%d x8 behind the reference API: created in %.1f s" % (FULL_LOG, time.perf_counter() - t0))
        for mix in ("C", "U"):
            wtns = synth.build_witness(FULL_LOG, mix)
            exp = _expected(zkey, wtns, FULL_LOG, mix)
            t0 = time.perf_counter()
            got = _prove(p, wtns)
            _progress("2^%d x8 behind the reference API, mix %s: groth16_prover_prove took %.1f ms" % (FULL_LOG, mix, 1e3 * (time.perf_counter() - t0)))
            assert got == exp, mix


def test_eight_bucket_class_ranks_at_configs2_size(device, full_zkey):
    """The bucket-class layout (ULTRAGROTH_SHARD=1x8, DESIGN.md section 7: every rank the WHOLE window tables and its residues of the
    bucket ids; an option, not the library's choice) rehearsed at the full 2^24 size, eight ranks one after the other on the one
    device (each holds 60 GiB of tables), circom-like scalars -- the million-entry bucket of the ones is split by scalar range over
    the eight ranks: every rank from its slices only, chain ranks 0-2 run their chain and take no part of h, ranks 3-7 a fifth of
    the H product each; the 384-byte blocks add up to the expected proof, byte for byte."""
    import ctypes as C
    import torch
    import ultragroth_amd as ug
    from ultragroth_amd import synth
    zkey, info = full_zkey
    world, n_dom, nv = 8, info["domainSize"], info["nVars"]
    wtns = synth.build_witness(FULL_LOG, "C")
    exp = _expected(zkey, wtns, FULL_LOG, "C")
    view = memoryview(zkey).cast("B")

    def sec(sid, skip=0):
        off, sz = O.section(zkey, "zkey", sid)
        return off + skip, sz - skip
    header = bytes(view[sec(2)[0]:sec(2)[0] + sec(2)[1]])
    c_off, c_sz = sec(4, 4)
    coefs = (C.c_char * c_sz).from_buffer(view[c_off:c_off + c_sz])

    def part(sid, rec, lo, hi):
        off, _ = sec(sid)
        n = (hi - lo) * rec
        return (C.c_char * n).from_buffer(view[off + lo * rec:off + lo * rec + n]) if n else (C.c_char * 0)()

    lays = [ug.ShardedGroth16Prover.shard_layout(nv, info["nPublic"], n_dom, k, world, 1) for k in range(world)]
    assert all(L.witness == (0, nv) and L.q_log == 7 for L in lays) and [L.chains for L in lays[:4]] == [[0], [1], [2], []]
    full = torch.empty((3, n_dom, 32), dtype=torch.uint8, device="cuda")
    total, first_rank = None, None
    r, s = fixed_rs()
    try:
        for k, L in enumerate(lays):
            (w0, w1), (c0, c1), (h0, h1) = L.ranges
            slices = (part(5, 64, w0, w1), part(6, 64, w0, w1), part(7, 128, w0, w1), part(8, 64, c0, c1), part(9, 64, h0, h1))
            _progress("2^%d classes x8: rank %d (%d residues, h %d..%d)" % (FULL_LOG, k, L.residues, h0, h1))
            p = ug.ShardedGroth16Prover.from_slices(header, coefs if L.chains else None, info["nCoefs"], slices, 0, k, world,
                                                    public_size=82 * info["nPublic"] + 4, layout=L)
            try:
                p.load_witness_part(wtns, 0)
                if L.chains:
                    p.load_witness_part(wtns, 1)
                    for c in L.chains:
                        p.hpoly_chain(c, full[c].data_ptr())
                if k == 0:
                    ug.set_test_blinding(r + s)             # rank 0 draws r and s when its products are queued
                try:
                    p.witness_msm_begin()
                finally:
                    ug.set_test_blinding(b"")
                assert p.h_range() == (h0, h1 - h0, n_dom)
                bufs = [full[c, h0:max(h1, h0 + 1)] for c in range(3)]
                torch.cuda.current_stream().synchronize()
                p.hpoly_combine(*(b.data_ptr() for b in bufs))
                hblk = p.run_h_msm()
                blk = p.witness_msm_end()[:320] + hblk[320:384]
            except BaseException:
                p.close()
                raise
            total = blk if total is None else ug.ShardedGroth16Prover.add_partials(total, blk)
            if k == 0:
                first_rank = p                             # (it finishes at the end -- it holds the blinding; one more rank fits beside it)
            else:
                p.close()
        got = first_rank.finish(total)
    finally:
        if first_rank is not None:
            first_rank.close()
        del coefs, view
    assert got == exp
    _progress("2^%d classes x8: done" % FULL_LOG)


@pytest.fixture(scope="module")
def huge(device):
    """BASELINE.json configs[3]'s circuit (2^HUGE_LOG constraints) with its expected proof, made once for the single-GPU
    and the 8-rank tests below (the oracle's H polynomial at this size is most of the cost of either)"""
    from ultragroth_amd import synth
    if HUGE_LOG == 0:
        pytest.skip("UG_HUGE_LOG=0")
    free, total = device.mem_info()
    if total < (200 << 30):
        pytest.skip("needs a 288 GB device")
    _progress("2^%d: building the circuit" % HUGE_LOG)
    zkey, wtns, info = synth.build_circuit(device, HUGE_LOG, mix="U")
    _progress("2^%d: expected proof" % HUGE_LOG)
    return zkey, wtns, info, _expected(zkey, wtns, HUGE_LOG)


def test_whole_proof_at_configs3_size(huge):
    """BASELINE.json configs[3]'s circuit, 2^26 constraints, on ONE GPU (classic windows; 2^28 coefficient records)"""
    import ultragroth_amd as ug
    zkey, wtns, info, exp = huge
    _progress("2^%d: creating the prover" % HUGE_LOG)
    with ug.Groth16Prover(zkey) as p:
        _progress("2^%d: proving" % HUGE_LOG)
        got = _prove(p, wtns)
    assert got == exp
    _progress("2^%d: done" % HUGE_LOG)


def test_eight_sliced_ranks_at_configs3_size(device, huge, monkeypatch):
    """BASELINE.json configs[3] as its text has it -- 2^26 constraints, base points sharded over EIGHT ranks -- rehearsed on
    one device at full size, byte for byte against the same expected proof: every rank is created from ITS slices only
    (ug_groth16_prover_create_sharded_slices, the witness ranges bench.py chooses: chain ranks get fewer points), the
    witness travels in two parts, ranks 0-2 run the three iFFT/twist/FFT chains, every rank combines its slice of h
    (ug_groth16_prover_hpoly_combine) and runs its H-MSM shard, the 384-byte partials are added and rank 0 finishes.
    This puts under test what the small sharded tests cannot: slice offsets above 2^24, the C section's index shift inside
    a slice, h ranges of 2^23 and the per-rank memory. Rank 0 runs the classic windows, the others fixed-base window
    tables (the per-rank choice a many-GPU node makes from its free memory). Ranks 3-7 live one after the other: the
    device holds the three chain ranks (26 GB of H-polynomial state each) and one more at a time."""
    import ctypes as C
    import torch
    import ultragroth_amd as ug
    import bench
    zkey, wtns, info, exp = huge
    world, n_dom, nv = 8, info["domainSize"], info["nVars"]
    # The rehearsal keeps the THREE chain ranks resident on the one device at once (each has a device of its own on a node).
    # A chain rank's H-polynomial state is 5 vectors + the h vector of the domain (6 x 32 N), the CSR matrix of 4 N coefficients
    # (~40 N), the twiddle and twist tables (~192 N) and the whole witness (32 N): ~460 bytes per constraint -- 29 GiB at 2^26,
    # 57.5 GiB at 2^27. Beside it: the point slices (320 bytes per witness point and 64 per H point, x 13 with window tables;
    # rank 0 runs without tables, chain ranks own ~0.6 of an eighth) of the three chain ranks and of the one plain rank that is
    # alive at a time, and 24 bytes per (scalar, window) entry of schedules. 2^26: ~195 GiB, fits; 2^27: ~390 GiB on a 288 GiB
    # device -- seen in round 4 as `hipMalloc: out of memory` while rank 3 was created (gpurun_out/r4_2p27.log). A limit of
    # rehearsing eight ranks on ONE device, not of the ranks.
    n8 = n_dom // 8
    need = (3 * 460 * n_dom + int(0.6 * n8) * 320 + 2 * int(0.6 * n8) * 320 * 13 + n8 * 320 * 13 + 4 * n8 * 64 * 13 + 4 * n8 * 13 * 24)
    have = torch.cuda.get_device_properties(0).total_memory
    if need > 0.97 * have:
        pytest.skip("eight ranks of a 2^%d domain rehearsed on one device need ~%.0f GiB at once (three resident chain ranks: 3 x %.1f GiB of "
                    "H-polynomial state, their point slices, the tables of the fourth rank) -- this device has %.0f GiB; on a node every "
                    "rank has a device of its own" % (HUGE_LOG, need / 2**30, 460 * n_dom / 2**30, have / 2**30))
    sl = n_dom // world
    view = memoryview(zkey).cast("B")

    def sec(sid, skip=0):
        off, sz = O.section(zkey, "zkey", sid)
        return off + skip, sz - skip
    header = bytes(view[sec(2)[0]:sec(2)[0] + sec(2)[1]])
    c_off, c_sz = sec(4, 4)
    coefs = (C.c_char * c_sz).from_buffer(view[c_off:c_off + c_sz])

    def part(sid, rec, lo, hi):
        off, _ = sec(sid)
        n = (hi - lo) * rec
        return (C.c_char * n).from_buffer(view[off + lo * rec:off + lo * rec + n]) if n else (C.c_char * 0)()

    def create(k):
        wr = bench.witness_slice(info, k, world)
        rg = ug.ShardedGroth16Prover.shard_ranges(nv, info["nPublic"], n_dom, k, world, wr)
        assert rg[0] == wr and rg[2] == (k * sl, (k + 1) * sl)
        (w0, w1), (c0, c1), (h0, h1) = rg
        slices = (part(5, 64, w0, w1), part(6, 64, w0, w1), part(7, 128, w0, w1), part(8, 64, c0, c1), part(9, 64, h0, h1))
        monkeypatch.setenv("ULTRAGROTH_TABLES", "0" if k == 0 else "1")
        _progress("2^%d x8: creating rank %d (witness %d..%d)" % (HUGE_LOG, k, w0, w1))
        return ug.ShardedGroth16Prover.from_slices(header, coefs if k < 3 else None, info["nCoefs"], slices, 0, k, world,
                                                   witness_range=wr, public_size=82 * info["nPublic"] + 4)

    def h_part(p, k, full):
        assert p.h_range() == (k * sl, sl, n_dom)
        bufs = [full[c, k * sl:(k + 1) * sl] for c in range(3)]          # contiguous views: no copy
        p.hpoly_combine(*(b.data_ptr() for b in bufs))
        return p.run_h_msm()[320:384]

    full = torch.empty((3, n_dom, 32), dtype=torch.uint8, device="cuda")
    total = None
    chain_ranks = []
    r, s = fixed_rs()
    try:
        # the order of bench.py's step at eight ranks: a chain rank runs its chain first and queues its witness products behind
        # it (witness_msm_begin: rank 0 draws r and s there); every rank drives its H branch while its products run and
        # collects them afterwards (witness_msm_end)
        for k in range(3):
            p = create(k)
            chain_ranks.append(p)
            p.load_witness_part(wtns, 0)
            p.load_witness_part(wtns, 1)
            p.hpoly_chain(k, full[k].data_ptr())
            if k == 0:
                ug.set_test_blinding(r + s)
            try:
                p.witness_msm_begin()
            finally:
                ug.set_test_blinding(b"")
        for k, p in enumerate(chain_ranks):
            hblk = h_part(p, k, full)
            blk = p.witness_msm_end()[:320] + hblk
            total = blk if total is None else ug.ShardedGroth16Prover.add_partials(total, blk)
            if k:
                p.close()
        for k in range(3, world):
            p = create(k)
            try:
                p.load_witness_part(wtns, 0)                 # a rank without a chain never sees the rest of the witness
                p.witness_msm_begin()
                hblk = h_part(p, k, full)
                blk = p.witness_msm_end()[:320] + hblk
            finally:
                p.close()
            total = ug.ShardedGroth16Prover.add_partials(total, blk)
        _progress("2^%d x8: finishing on rank 0" % HUGE_LOG)
        got = chain_ranks[0].finish(total)                   # (with the blinding rank 0 drew when its products were queued)
    finally:
        for p in chain_ranks:
            p.close()
        del coefs, view
    assert got == exp
    _progress("2^%d x8: done" % HUGE_LOG)


ULTRA_LOG = int(os.environ.get("UG_ULTRA_LOG", "22"))
ULTRA_BLINDING = (bytes(range(1, 32)), bytes(range(40, 71)), bytes(range(80, 111)))          # r_k, r, s


@pytest.fixture(scope="module")
def ultra(device):
    """BASELINE.json configs[4]'s circuit (UltraGroth, 2^22 constraints, lookup table 2^16) with the oracle's proof, made once for
    the single-GPU and the 8-rank tests below"""
    from ultragroth_amd import synth
    _progress("ultragroth 2^%d: building the circuit" % ULTRA_LOG)
    zkey, uwtns, info = synth.build_ultra_circuit(device, ULTRA_LOG, mix="C", lookup_log=16)
    _progress("ultragroth 2^%d: oracle" % ULTRA_LOG)
    exp = O.ultra_groth_prove(zkey, uwtns, *(int.from_bytes(b, "little") for b in ULTRA_BLINDING))
    return zkey, uwtns, info, exp


def test_eight_sliced_ultragroth_ranks_at_configs4_size(device, ultra):
    """BASELINE.json configs[4] as its text has it -- UltraGroth on a 2^22 circuit over EIGHT ranks -- rehearsed on one device at
    full size (src/ultra_groth.cpp:401-462 sharded; src/prover.cpp:226-300 is the single-process caller): every rank is created
    from the header section and ITS slices only (ug_ultra_groth_prover_create_sharded_slices: the witness-indexed sets, the round
    and final sets with their slices of the two index lists, H; ranks 3-7 without a coefficient matrix), every rank commits to its
    part of the round set, rank 0 closes the round, every rank derives the challenge and completes its witness (lookup 2^16), the
    final round runs queued -- chain ranks their chain first, then the witness products with the H branch beside them -- the
    384-byte blocks are added and rank 0 finishes: byte for byte the oracle's proof of the single-GPU test below."""
    import torch
    import ultragroth_amd as ug
    from ultragroth_amd import synth
    zkey, uwtns, info, exp = ultra
    world, n_dom, nv = 8, info["domainSize"], info["nVars"]
    assert synth.build_ultra_witness(ULTRA_LOG, "C", lookup_log=16) == uwtns       # what bench.py's ranks make for themselves
    ranks = []
    try:
        for k in range(world):
            rg = ug.ShardedUltraGrothProver.shard_ranges(nv, n_dom, info["nC1"], info["nC2"], k, world)
            _progress("ultragroth 2^%d x8: creating rank %d (witness %d..%d)" % (ULTRA_LOG, k, rg[0][0], rg[0][1]))
            header, coefs, slices = synth.build_ultra_circuit_slices(device, ULTRA_LOG, rg, with_coefs=(k < 3))
            ranks.append(ug.ShardedUltraGrothProver.from_slices(header, coefs, 4 * n_dom, slices, 0, k, world,
                                                                public_size=82 * (info["nPublic"] - 1) + 4))
            del coefs, slices
        full = torch.empty((3, n_dom, 32), dtype=torch.uint8, device="cuda")
        ug.set_test_blinding(b"".join(ULTRA_BLINDING))
        try:
            total = bytes(64)
            for p in ranks:
                p.load_witness(uwtns)
                total = ug.ShardedUltraGrothProver.add_records(total, p.round_commit())
            commitment = ranks[0].round_finish(total)
            for p in ranks:
                p.apply_commitment(commitment)
            _progress("ultragroth 2^%d x8: final round" % ULTRA_LOG)
            for k in range(3):                                 # eight ranks: a chain rank runs its chain first, alone
                ranks[k].hpoly_chain(k, full[k].data_ptr())
            for p in ranks:
                p.witness_msm_begin()
            acc = None
            for p in ranks:
                first, cnt, _ = p.h_range()
                bufs = [full[k, first:first + cnt] for k in range(3)]
                torch.cuda.current_stream().synchronize()
                p.hpoly_combine(*(b.data_ptr() for b in bufs))
                hpart = p.run_h_msm()
                part = p.witness_msm_end()[:320] + hpart[320:384]
                acc = part if acc is None else ug.ShardedGroth16Prover.add_partials(acc, part)
            got = ranks[0].finish(acc)
        finally:
            ug.set_test_blinding(b"")
    finally:
        for p in ranks:
            p.close()
    assert got == exp
    _progress("ultragroth 2^%d x8: done" % ULTRA_LOG)


def test_ultragroth_at_configs4_size(device, ultra):
    """BASELINE.json configs[4]'s circuit on one GPU inside the suite: UltraGroth, 2^22 constraints, lookup table 2^16
    (SURVEY.md section 8d cfg 5), a CREATED prover (window tables), two proofs on the one object; == the oracle (parity of the
    UltraGroth whole proof is unpinned upstream: no fixture exists there; the oracle's pieces are pinned)"""
    import ultragroth_amd as ug
    zkey, uwtns, info, exp = ultra
    log_domain = ULTRA_LOG
    rk, r, s = ULTRA_BLINDING
    _progress("ultragroth 2^%d: proving" % log_domain)
    with ug.UltraGrothProver(zkey) as p:
        for _ in range(2):
            ug.set_test_blinding(rk + r + s)
            try:
                assert p.prove(uwtns) == exp
            finally:
                ug.set_test_blinding(b"")


def test_g1_only_created_prover_at_configs1_size(device):
    """BASELINE.json configs[1] as written: 2^20 constraints, G1 MSM + Fr NTT only (the B1, B2 and C sections are all-infinity
    sets of the right size), through a created prover; == the oracle's whole proof and == the closed-form assembly"""
    import ultragroth_amd as ug
    from ultragroth_amd import synth
    zkey, wtns, info = synth.build_circuit(device, 20, mix="U", g1_only=True)
    r, s = fixed_rs()
    ri, si = int.from_bytes(r, "little"), int.from_bytes(s, "little")
    with ug.Groth16Prover(zkey) as p:
        got = _prove(p, wtns)
    exp = O.groth16_prove(zkey, wtns, ri, si)
    assert got == (exp[0], exp[1])
    assert got == closed_form.groth16_expected(zkey, wtns, synth.SEEDS, synth.g1_generator_record(), synth.g2_generator_record(),
                                               ri, si, g1_only=True)
