"""GPU tests beyond the fixture: synthetic circuits against the oracle at sizes it finishes in seconds, the sharded
(multi-GPU) path, UltraGroth, and size-independent properties at BASELINE.json's full sizes."""
import ctypes as C
import json
import os
import random

import numpy as np
import pytest

import oracle as O
from conftest import fixed_rs

pytestmark = pytest.mark.gpu

FULL_LOG = int(os.environ.get("UG_FULL_LOG", "24"))          # BASELINE.json configs[2]


def _sec(buf, ftype, sid):
    off, sz = O.section(buf, ftype, sid)
    return buf[off:off + sz]


@pytest.mark.parametrize("log_domain,mix", [(12, "U"), (14, "C"), (16, "U")])
def test_synthetic_circuit_proof_bit_exact(device, log_domain, mix):
    """whole prove through the reference's C API == oracle, on seeded synthetic circuits (unsorted coefficients,
    circom-like and uniform witnesses)"""
    import ultragroth_amd as ug
    from ultragroth_amd import synth
    zkey, wtns, info = synth.build_circuit(device, log_domain, mix=mix)
    r, s = fixed_rs()
    ug.set_test_blinding(r + s)
    try:
        proof, pub = ug.groth16_prover(zkey, wtns)
    finally:
        ug.set_test_blinding(b"")
    exp = O.groth16_prove(zkey, wtns, int.from_bytes(r, "little"), int.from_bytes(s, "little"))
    assert (proof, pub) == (exp[0], exp[1])


def test_synthetic_points_are_the_generator_walk(device):
    from ultragroth_amd import synth
    g1 = synth.g1_generator_record()
    g2 = synth.g2_generator_record()
    assert O.lib.ugo_g1_on_curve(g1) == 1 and O.lib.ugo_g2_on_curve(g2) == 1
    pts = bytes(synth.synth_points(device, 40, 1000))
    for i in (0, 1, 17, 39):
        assert pts[64 * i:64 * i + 64] == O.g1_mul(g1, 1000 + i)
    pts2 = bytes(synth.synth_points(device, 9, 5, g2=True))
    for i in (0, 3, 8):
        assert pts2[128 * i:128 * i + 128] == O.g2_mul(g2, 5 + i)


def test_hpoly_2_18_matches_oracle(device):
    from ultragroth_amd import synth
    zkey, wtns, info = synth.build_circuit(device, 18, mix="C", g1_only=True)
    coefs = _sec(zkey, "zkey", 4)[4:]
    w = _sec(wtns, "wtns", 2)
    hp = device.hpoly(coefs, info["nCoefs"], info["domainSize"], info["nVars"])
    h = device.download(hp.run(device.dvec(info["nVars"], w)), 0, info["domainSize"])
    assert h == O.hpoly(coefs, info["nCoefs"], w, info["nVars"], info["domainSize"])


def test_long_coefficient_rows(device):
    """rows with thousands of entries (range accumulation in the mat-vec) and empty rows"""
    rng = random.Random(1)
    logn, nvars = 10, 900
    n = 1 << logn
    recs = []
    for i in range(5000):
        recs.append((0, 3, rng.randrange(nvars)))
    for i in range(3000):
        recs.append((1, 3, rng.randrange(nvars)))
    for c in range(0, n, 7):
        recs.append((rng.randrange(2), c, rng.randrange(nvars)))
    rng.shuffle(recs)
    raw = b"".join(int(m).to_bytes(4, "little") + int(c).to_bytes(4, "little") + int(s).to_bytes(4, "little") +
                   O.to_le(rng.randrange(O.R_MOD)) for m, c, s in recs)
    w = b"".join(O.to_le(rng.randrange(O.R_MOD)) for _ in range(nvars))
    hp = device.hpoly(raw, len(recs), n, nvars)
    h = device.download(hp.run(device.dvec(nvars, w)), 0, n)
    assert h == O.hpoly(raw, len(recs), w, nvars, n)


def test_bad_coefficient_record_is_rejected(device):
    import ultragroth_amd as ug
    raw = (0).to_bytes(4, "little") + (99999).to_bytes(4, "little") + (0).to_bytes(4, "little") + bytes(32)
    with pytest.raises(ug.DeviceError):
        device.hpoly(raw, 1, 1024, 10)


def test_sharded_prover_equals_unsharded(device, zkey, wtns):
    """the N > 1 path on one GPU: 3 ranks' partial sums, added, give the unsharded proof byte for byte"""
    import ultragroth_amd as ug
    r, s = fixed_rs()
    world = 3
    ranks = [ug.ShardedGroth16Prover(zkey, 0, k, world) for k in range(world)]
    total = None
    for p in ranks:
        p.load_witness(wtns)
        part = p.run()
        total = part if total is None else ug.ShardedGroth16Prover.add_partials(total, part)
    ug.set_test_blinding(r + s)
    try:
        proof, pub = ranks[0].finish(total)
    finally:
        ug.set_test_blinding(b"")
    exp = O.groth16_prove(zkey, wtns, int.from_bytes(r, "little"), int.from_bytes(s, "little"))
    assert (proof, pub) == (exp[0], exp[1])
    for p in ranks:
        p.close()


def test_sharded_prover_with_caller_chosen_ranges(device, zkey, wtns):
    """uneven witness slices (bench.py gives ranks that also run an NTT chain fewer points), one of them empty and one
    cutting through the public signals, so the C section's shifted range starts mid-slice"""
    import ultragroth_amd as ug
    r, s = fixed_rs()
    n = O.zkey_info(zkey)["nVars"]
    cuts = [0, 1, 1, n // 3, n]
    ranks = [ug.ShardedGroth16Prover(zkey, 0, k, 4, witness_range=(cuts[k], cuts[k + 1])) for k in range(4)]
    total = None
    for p in ranks:
        p.load_witness(wtns)
        part = p.run()
        total = part if total is None else ug.ShardedGroth16Prover.add_partials(total, part)
    ug.set_test_blinding(r + s)
    try:
        proof, pub = ranks[0].finish(total)
    finally:
        ug.set_test_blinding(b"")
    exp = O.groth16_prove(zkey, wtns, int.from_bytes(r, "little"), int.from_bytes(s, "little"))
    assert (proof, pub) == (exp[0], exp[1])
    with pytest.raises(ug.ProverError, match="witness range outside"):
        ug.ShardedGroth16Prover(zkey, 0, 0, 2, witness_range=(0, n + 1))
    for p in ranks:
        p.close()


def test_sharded_prover_with_split_hpoly(device):
    """the N > 1 path with the H-polynomial chains split over ranks (as bench.py does over RCCL), on one GPU:
    rank k mod 2 computes chain k into a device buffer, each rank combines its slices, proofs match the oracle"""
    import torch
    import ultragroth_amd as ug
    from ultragroth_amd import synth
    zkey, wtns, info = synth.build_circuit(device, 13, mix="C")
    world, n_dom = 2, info["domainSize"]
    sl = n_dom // world
    ranks = [ug.ShardedGroth16Prover(zkey, 0, k, world) for k in range(world)]
    for p in ranks:
        p.load_witness(wtns)
    parts = [p.run_witness_msm() for p in ranks]
    full = torch.empty((3, n_dom, 32), dtype=torch.uint8, device="cuda")
    for k in range(3):
        ranks[k % world].hpoly_chain(k, full[k].data_ptr())
    torch.cuda.synchronize()
    total = None
    for r, p in enumerate(ranks):
        assert p.h_range() == (r * sl, sl, n_dom)
        sl_bufs = [full[k, r * sl:(r + 1) * sl].contiguous() for k in range(3)]
        torch.cuda.synchronize()
        p.hpoly_combine(*(b.data_ptr() for b in sl_bufs))
        part = parts[r][:320] + p.run_h_msm()[320:384]
        total = part if total is None else ug.ShardedGroth16Prover.add_partials(total, part)
    r_, s_ = fixed_rs()
    ug.set_test_blinding(r_ + s_)
    try:
        proof, pub = ranks[0].finish(total)
    finally:
        ug.set_test_blinding(b"")
    exp = O.groth16_prove(zkey, wtns, int.from_bytes(r_, "little"), int.from_bytes(s_, "little"))
    assert (proof, pub) == (exp[0], exp[1])
    for p in ranks:
        p.close()


def test_witness_products_queued_beside_the_h_branch(device):
    """ug_groth16_prover_witness_msm_begin / _end (what bench.py and the many-device prover do since round 3): the witness
    products of every rank are queued and left to run while the rank's H branch -- chain, combine, H product, on its second
    stream -- is driven from the same host thread; rank 0 draws r and s in _begin. Byte for byte the oracle's proof; the call
    order is checked (begin twice, end without begin, a witness load between the two)."""
    import torch
    import ultragroth_amd as ug
    from ultragroth_amd import synth
    zkey, wtns, info = synth.build_circuit(device, 13, mix="U", seed=0x5EED0077)
    world, n_dom = 3, info["domainSize"]
    ranks = [ug.ShardedGroth16Prover(zkey, 0, k, world) for k in range(world)]
    with pytest.raises(ug.ProverError, match="no witness loaded"):
        ranks[0].witness_msm_begin()
    for p in ranks:
        p.load_witness(wtns)
    with pytest.raises(ug.ProverError, match="no witness products queued"):
        ranks[1].witness_msm_end()
    r_, s_ = fixed_rs()
    full = torch.empty((3, n_dom, 32), dtype=torch.uint8, device="cuda")
    for turn in range(2):                                    # twice on the same objects
        ug.set_test_blinding(r_ + s_)
        try:
            for p in ranks:
                p.witness_msm_begin()                        # rank 0 draws here
            with pytest.raises(ug.ProverError, match="already queued"):
                ranks[2].witness_msm_begin()
            with pytest.raises(ug.ProverError, match="still read the witness"):
                ranks[1].load_witness_part(wtns, 0)
            for k in range(3):
                ranks[k].hpoly_chain(k, full[k].data_ptr())
            total = None
            for r, p in enumerate(ranks):
                first, cnt, _ = p.h_range()
                bufs = [full[k, first:first + cnt].contiguous() for k in range(3)]
                torch.cuda.synchronize()
                p.hpoly_combine(*(b.data_ptr() for b in bufs))
                hpart = p.run_h_msm()
                part = p.witness_msm_end()[:320] + hpart[320:384]
                total = part if total is None else ug.ShardedGroth16Prover.add_partials(total, part)
            got = ranks[0].finish(total)
        finally:
            ug.set_test_blinding(b"")
        exp = O.groth16_prove(zkey, wtns, int.from_bytes(r_, "little"), int.from_bytes(s_, "little"))
        assert got == (exp[0], exp[1])
    # the blocking form on the same objects still gives the same sums
    assert ranks[1].run_witness_msm()[:320] == (lambda p: (p.witness_msm_begin(), p.witness_msm_end())[1])(ranks[1])[:320]
    for p in ranks:
        p.close()


def test_sharded_prover_from_slices_and_witness_parts(device):
    """what bench.py does with N > 1 ranks: every rank is created from ITS slices of the point sections only
    (ug_groth16_prover_create_sharded_slices; the slices are the same generator walk entered at the slice), ranks beyond
    the three chain ranks hold no coefficient matrix and upload only their slice of the witness; the chain ranks upload
    the rest of the witness on the H branch's stream. 4 ranks on one GPU, caller-chosen uneven witness ranges."""
    import torch
    import ultragroth_amd as ug
    from ultragroth_amd import synth
    log_domain, world = 13, 4
    zkey, wtns, info = synth.build_circuit(device, log_domain, mix="U", seed=0x5EED0000)
    n_dom, nv = info["domainSize"], info["nVars"]
    cuts = [0, 1500, 3700, 6000, nv]
    sl = n_dom // world
    ranks = []
    for k in range(world):
        wr = (cuts[k], cuts[k + 1])
        rg = ug.ShardedGroth16Prover.shard_ranges(nv, 1, n_dom, k, world, wr)
        assert rg[0] == wr and rg[2] == (k * sl, (k + 1) * sl)
        assert rg[1] == (max(wr[0] - 2, 0), max(wr[1] - 2, 0))                 # C follows the witness slice, shifted by nPublic + 1
        header, coefs, slices = synth.build_circuit_slices(device, log_domain, rg, with_coefs=(k < 3))
        ranks.append(ug.ShardedGroth16Prover.from_slices(header, coefs, info["nCoefs"], slices, 0, k, world, witness_range=wr,
                                                         public_size=86))
    for p in ranks:
        p.load_witness_part(wtns, 0)
    parts = [p.run_witness_msm() for p in ranks]
    with pytest.raises(ug.ProverError, match="whole witness has not been loaded"):
        ranks[0].hpoly_chain(0, 0x1000)
    full = torch.empty((3, n_dom, 32), dtype=torch.uint8, device="cuda")
    for k in range(3):
        ranks[k].load_witness_part(wtns, 1)
        ranks[k].hpoly_chain(k, full[k].data_ptr())
    ranks[3].load_witness_part(wtns, 1)
    with pytest.raises(ug.ProverError, match="created without the coefficient matrix"):
        ranks[3].hpoly_chain(0, full[0].data_ptr())
    torch.cuda.synchronize()
    total = None
    for r, p in enumerate(ranks):
        bufs = [full[k, r * sl:(r + 1) * sl].contiguous() for k in range(3)]
        torch.cuda.synchronize()
        p.hpoly_combine(*(b.data_ptr() for b in bufs))
        part = parts[r][:320] + p.run_h_msm()[320:384]
        total = part if total is None else ug.ShardedGroth16Prover.add_partials(total, part)
    r_, s_ = fixed_rs()
    ug.set_test_blinding(r_ + s_)
    try:
        got = ranks[0].finish(total)
    finally:
        ug.set_test_blinding(b"")
    exp = O.groth16_prove(zkey, wtns, int.from_bytes(r_, "little"), int.from_bytes(s_, "little"))
    assert got == (exp[0], exp[1])
    for p in ranks:
        p.close()


@pytest.mark.parametrize("world,point_ranges,mix", [(4, 1, "C"), (4, 2, "U"), (8, 2, "C"), (5, 1, "U")])
def test_sharded_prover_with_bucket_class_layouts(device, world, point_ranges, mix):
    """ug_groth16_shard_layout + ug_groth16_prover_create_sharded_layout: the witness products cut over the
    ranks by BUCKET CLASS -- the ranks of a group hold the same base-point range with its whole window tables and take the
    bucket ids of their residues; the lowest buckets, where a circom-like witness piles up its ones, go by scalar range -- in
    P groups of world / P ranks; chain ranks of five ranks or more take no part of the H product (empty h range). Every rank
    from its slices only, the proof byte for byte the oracle's. (Not the layout the library chooses by itself: base-point
    ranges measured faster, profiles/r04_rank_phases_classes.txt.)"""
    import torch
    import ultragroth_amd as ug
    from ultragroth_amd import synth
    log_domain = 16
    zkey, wtns, info = synth.build_circuit(device, log_domain, mix=mix, seed=0x5EED0000)
    n_dom, nv = info["domainSize"], info["nVars"]
    lays = [ug.ShardedGroth16Prover.shard_layout(nv, 1, n_dom, k, world, point_ranges) for k in range(world)]
    # the layouts tile everything: h ranges [0, N), per group the residues [0, Q) and the special ranges the group's scalars
    assert lays[0].h[0] == 0 and lays[-1].h[1] == n_dom and all(a.h[1] == b.h[0] for a, b in zip(lays, lays[1:]))
    B = world // point_ranges
    for g in range(point_ranges):
        grp = lays[g * B:(g + 1) * B]
        assert all(L.witness == grp[0].witness and L.q_log == grp[0].q_log and L.q_log > 0 for L in grp)
        assert grp[0].first_residue == 0 and grp[-1].first_residue + grp[-1].residues == 1 << grp[0].q_log
        assert all(a.first_residue + a.residues == b.first_residue and a.special[1] == b.special[0] for a, b in zip(grp, grp[1:]))
        assert (grp[0].special[0], grp[-1].special[1]) == grp[0].witness
    assert lays[0].witness[0] == 0 and lays[-1].witness[1] == nv
    if world >= 5:
        assert all(L.h[0] == L.h[1] for L in lays[:3]) and all(L.h[1] > L.h[0] for L in lays[3:])
    full = torch.empty((3, n_dom, 32), dtype=torch.uint8, device="cuda")
    total = None
    # (ranks one after the other: each holds whole tables of its group's range)
    chains_done = False
    provers = []
    for k, L in enumerate(lays):
        header, coefs, slices = synth.build_circuit_slices(device, log_domain, L.ranges, with_coefs=bool(L.chains))
        p = ug.ShardedGroth16Prover.from_slices(header, coefs, info["nCoefs"], slices, 0, k, world, public_size=86, layout=L)
        provers.append(p)
        p.load_witness_part(wtns, 0)
        if L.chains:
            p.load_witness_part(wtns, 1)
            for c in L.chains:
                p.hpoly_chain(c, full[c].data_ptr())
    torch.cuda.synchronize()
    r_, s_ = fixed_rs()
    ug.set_test_blinding(r_ + s_)                  # (rank 0 draws r and s when its products are queued: witness_msm_begin)
    try:
        for k, (L, p) in enumerate(zip(lays, provers)):
            p.witness_msm_begin()                                        # the queued form, the H branch beside it
            first, cnt, _ = p.h_range()
            assert (first, first + cnt) == L.h
            bufs = [full[c, first:first + max(cnt, 1)].contiguous() for c in range(3)]
            torch.cuda.synchronize()
            p.hpoly_combine(*(b.data_ptr() for b in bufs))
            hpart = p.run_h_msm()
            part = p.witness_msm_end()[:320] + hpart[320:384]
            assert part[:320] == p.run_witness_msm()[:320]               # the blocking call gives the same sums
            total = part if total is None else ug.ShardedGroth16Prover.add_partials(total, part)
        got = provers[0].finish(total)
    finally:
        ug.set_test_blinding(b"")
    exp = O.groth16_prove(zkey, wtns, int.from_bytes(r_, "little"), int.from_bytes(s_, "little"))
    assert got == (exp[0], exp[1])
    for p in provers:
        p.close()


def test_bucket_class_layout_needs_the_window_tables(device, monkeypatch):
    """a rank of a bucket-class layout that cannot have its tables (here: switched off) refuses to be created -- with the
    reason -- instead of proving with classic windows, whose result blocks could not hold one bucket set per window and residue"""
    import ultragroth_amd as ug
    from ultragroth_amd import synth
    monkeypatch.setenv("ULTRAGROTH_TABLES", "0")
    L = ug.ShardedGroth16Prover.shard_layout((1 << 15) - 1, 1, 1 << 15, 1, 4, 1)
    header, coefs, slices = synth.build_circuit_slices(device, 15, L.ranges, with_coefs=False)
    with pytest.raises(ug.ProverError, match="needs the fixed-base window tables"):
        ug.ShardedGroth16Prover.from_slices(header, coefs, 4 << 15, slices, 0, 1, 4, public_size=86, layout=L)


@pytest.mark.parametrize("shard", ["1x4", "2x2", "auto"])
def test_api_on_several_devices_with_bucket_classes(device, monkeypatch, shard):
    """ULTRAGROTH_DEVICES with ULTRAGROTH_SHARD=PxB: the object the reference's groth16_prover_create returns shards its witness
    products by bucket class (four ranks on the one device, created one after the other: each holds whole tables); two proofs on
    one object == the single-device prover's"""
    import ultragroth_amd as ug
    from ultragroth_amd import synth
    zkey, wtns, info = synth.build_circuit(device, 16, mix="C", seed=0x5EED0000)
    r_, s_ = fixed_rs()
    exp = O.groth16_prove(zkey, wtns, int.from_bytes(r_, "little"), int.from_bytes(s_, "little"))
    monkeypatch.setenv("ULTRAGROTH_DEVICES", "0,0,0,0")
    monkeypatch.setenv("ULTRAGROTH_SHARD", shard)
    with ug.Groth16Prover(zkey) as p:
        for _ in range(2):
            ug.set_test_blinding(r_ + s_)
            try:
                got = p.prove(wtns)
            finally:
                ug.set_test_blinding(b"")
            assert got == (exp[0], exp[1])


def test_bench_two_ranks_over_rccl():
    """the real N = 2 launch line of the driver (one process per GPU, backend nccl = RCCL) with --check; needs two GPUs --
    on a one-GPU box the same control flow is rehearsed over gloo (tools/run_multi.sh)"""
    import subprocess
    import sys
    import ultragroth_amd as ug
    if ug.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for extra in ([], ["--ultra"]):
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                            "--master-port", "29631", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                            "--log-domain", "16", "--no-cpu-baseline", "--check"] + extra, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        line = json.loads(r.stdout.strip().splitlines()[-1])
        assert line["n_gpus"] == 2 and ("bit-exact" in (line.get("check") or line["config"]["workload"]))


def test_bench_driver_line_with_four_ranks_over_gloo():
    """The driver's own launch line for a many-GPU run -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W`, no other flag -- with N = 4 ranks sharing
    the one device over gloo (RCCL refuses two ranks on one device, and this pool allows six processes on a GPU: the test
    runner, the launcher and four ranks; N = 8 itself is the driver's to launch), at 2^16 so that it takes seconds. Four ranks
    means a rank without a chain and slices of different length; the chain-first order of a large node is forced. The whole
    JSON contract of the line is asserted; `cpu_baseline` is null at N > 1 (the run contract measures it at N = 1 only) and
    says so; every rank has left its process group before rank 0 prints."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, UG_BENCH_BACKEND="gloo", UG_BENCH_ONE_DEVICE="1", OMP_NUM_THREADS="4", UG_BENCH_CHAIN_ORDER="first")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
                        "--master-port", "29633", os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "2", "--warmup", "1",
                        "--log-domain", "16"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "comm"):
        assert key in d, key
    assert d["metric"] == "proofs/s" and d["n_gpus"] == 4 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "strong"
    assert abs(d["value"] * d["ms_per_step"] - 1e3) < 1e-3
    assert d["comm"] == {"backend": "gloo", "world_size": 4, "rccl_version": None, "devices_visible": d["comm"]["devices_visible"]}
    assert d["cpu_baseline"] is None and "N = 1" in d["cpu_baseline_note"]
    assert "base-range shard x4, H-poly chains split over ranks" in d["config"]["parallelism"] and "workload" in d["config"]
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "valu-issue") and rf["launches"] > 0 and 0 < rf["frac"] < 1
    # the same launch with --check: the four ranks' proof is the single-device proof, byte for byte
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
                        "--master-port", "29634", os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "1",
                        "--log-domain", "16", "--check"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    assert json.loads([ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")][-1])["check"] == "bit-exact"


def test_bench_sharded_path_over_rccl_with_one_rank():
    """A one-GPU box cannot host two RCCL ranks, but it can run the sharded code path of bench.py over a REAL RCCL process group
    of one rank (UG_BENCH_FORCE_DIST=1): slices-only creation, the witness in two parts, the chains on their own thread, RCCL
    scatter of the evaluation slices, all_gather_into_tensor of the partial blocks, all_reduce of the time, barrier -- the very
    calls the driver's N = 2, 4, 8 runs make -- with --check bit-exact; `comm` shows what the communicator was"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, UG_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29641")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--log-domain", "16",
                        "--no-cpu-baseline", "--check"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["check"] == "bit-exact" and d["n_gpus"] == 1
    assert d["comm"]["backend"] == "nccl" and d["comm"]["world_size"] == 1 and d["comm"]["rccl_version"]
    assert "base-range shard x1, H-poly chains split over ranks" in d["config"]["parallelism"]
    assert d["prove_call_ms_per_step"] is None and d["pipelined_proofs_per_s"] is None


def test_bench_ultragroth_sharded_path_over_rccl_with_one_rank():
    """the same for `bench.py --ultra`: the sharded UltraGroth step -- all_gather_into_tensor of the commitment parts, broadcast of
    the commitment, the queued final round, RCCL scatter of the evaluation slices, all_gather_into_tensor of the partial blocks --
    over a real RCCL process group of one rank, --check bit-exact"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, UG_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29642")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--ultra", "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--log-domain", "14", "--check"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["config"]["protocol"] == "ultragroth" and "bit-exact" in d["config"]["workload"]
    assert "section-range shard x1, H-poly chains split over ranks" in d["config"]["parallelism"]


def test_bench_line_contract_one_gpu():
    """`python bench.py` as the driver runs it at N = 1 (here at 2^16 so that it takes seconds): exactly one JSON line
    with the contract's keys, the `roofline` and `cpu_baseline` objects; the K timed steps run one after the other on the
    resident witness, the PCIe-inclusive call and the two-thread figure come as extra keys; `--check` bit-exact (a mismatch
    would make the exit code 3); the CPU baseline is measured at the benchmarked size, not extrapolated"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--log-domain", "16",
                        "--check"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["metric"] == "proofs/s" and d["unit"] == "proofs/s" and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    assert abs(d["value"] * d["ms_per_step"] - 1e3) < 1e-6 * 1e3
    assert "resident in HBM" in d["config"]["workload"] and d["comm"] is None
    assert d["prove_call_ms_per_step"] > d["ms_per_step"] * 0.5 and d["witness_upload_ms_per_proof"] > 0
    assert d["pipelined_proofs_per_s"] > 0 and d["pipelined_host_threads"] == 2
    # the headline runs in the library's fastest honest configuration (H branch beside the witness products); the MSM | FFT split
    # and the launch times of the roofline come from K un-overlapped steps right after the region
    assert d["config"]["overlap"] == 1 and "ULTRAGROTH_OVERLAP=1" in d["value_definition"] and "resident" in d["value_definition"]
    assert d["unoverlapped_ms_per_step"] > 0
    assert d["msm_ms_per_proof"] > 0 and d["fft_ms_per_proof"] > 0 and d["msm_ms_per_proof"] + d["fft_ms_per_proof"] < d["unoverlapped_ms_per_step"] * 1.05
    # SURVEY 8(d)'s ms/proof -- groth16_prover_prove with the .wtns in host memory -- is a first-class figure of the line
    assert abs(d["api_value"] * d["api_ms_per_step"] - 1e3) < 1e-6 * 1e3 and d["api_ms_per_step"] == d["prove_call_ms_per_step"]
    assert "groth16_prover_prove" in d["api_definition"]
    rf = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernels", "board", "launch_times_from"):
        assert key in rf, key
    # achieved / peak / frac are the HBM figures (algorithmic bytes against 8 TB/s); `bound` names the nearer roof
    assert rf["bound"] in ("hbm", "valu-issue") and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and rf["launches"] > 0
    assert rf["bound_frac"] == max(rf["frac"], rf["issue_bound"]["frac"])
    # the counter traffic is measured by this very run (two rocprofv3 --pmc children), not read from a committed file
    assert rf["traffic"] and rf["traffic"] > 0 and rf["traffic_source"].startswith("measured in this run"), rf["traffic_source"]
    assert all(rf["ms_per_step"] >= k["ms_per_step"] for k in rf["kernels"].values())        # the top entry is the dominant kernel
    assert rf["launches"] % 3 == 0                                                            # counted over the K timed steps only
    cb = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample", "extrapolated"):
        assert key in cb, key
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["extrapolated"] is False
    assert d.get("check") == "bit-exact"


def test_api_errors_on_device(zkey, wtns):
    import ultragroth_amd as ug
    with ug.Groth16Prover(zkey) as p:
        with pytest.raises(ug.ProverError) as e:
            p.prove(wtns, proof_size=100)
        assert e.value.code == ug.PROVER_ERROR_SHORT_BUFFER
        assert e.value.message == "Proof buffer is too short. Minimum size: 810, actual size: 100"     # prover.cpp:127-133
        with pytest.raises(ug.ProverError) as e:
            p.prove(wtns, public_size=5)
        assert e.value.message == "Public buffer is too short. Minimum size: 86, actual size: 5"
        short = bytearray(wtns)
        nv_off = O.section(wtns, "wtns", 1)[0] + 36
        short[nv_off:nv_off + 4] = (1002).to_bytes(4, "little")
        with pytest.raises(ug.ProverError) as e:
            p.prove(bytes(short))
        assert e.value.code == ug.PROVER_INVALID_WITNESS_LENGTH
        assert e.value.message == "Invalid witness length. Circuit: 1003, witness: 1002"               # prover.cpp:190-195
        with pytest.raises(ug.ProverError) as e:
            p.prove(zkey)
        assert e.value.message == "Invalid file type. It should be wtns and it is zkey"
        proof, pub = p.prove(wtns)                          # still usable after errors
        assert json.loads(proof)["protocol"] == "groth16"


def test_ultragroth_matches_oracle(device):
    """UltraGroth (parity unpinned upstream: no fixture exists): product == oracle on a synthetic protocol-1337 circuit"""
    import ultragroth_amd as ug
    from ultragroth_amd import synth
    zkey, uwtns, info = synth.build_ultra_circuit(device, 12)
    rk, r, s = bytes(range(1, 32)), bytes(range(40, 71)), bytes(range(80, 111))
    ug.set_test_blinding(rk + r + s)
    try:
        proof, pub = ug.ultra_groth_prover(zkey, uwtns)
    finally:
        ug.set_test_blinding(b"")
    exp = O.ultra_groth_prove(zkey, uwtns, int.from_bytes(rk, "little"), int.from_bytes(r, "little"), int.from_bytes(s, "little"))
    assert (proof, pub) == exp
    j = json.loads(proof)
    assert j["protocol"] == "ultragroth" and set(j) == {"pi_a", "pi_b", "pi_f", "pi_r", "protocol"}
    assert len(json.loads(pub)) == info["nPublic"] - 1            # rand_indx is skipped (prover.cpp:89-105)
    with pytest.raises(ug.ProverError) as e:
        ug.groth16_prover(zkey, uwtns)
    assert e.value.message == "zkey file is not groth16"


def test_lookup_table_rows_follow_the_reference_overloads(device):
    """ug_fr_lookup_table against the oracle's row (itself pinned to the reference's RawFr calls, tests/test_oracle.py):
    frequencies >= 2^31 enter as freq - 2^32 (mul(int, Element), build/fr.hpp:251), a zero sum inverts to zero; and a whole
    UltraGroth proof whose frequencies use the full uint32 range"""
    import struct
    import ultragroth_amd as ug
    from ultragroth_amd import synth
    rng = random.Random(31)
    L = 300
    freq = [rng.choice([0, 1, (1 << 31) - 1, 1 << 31, (1 << 32) - 1, rng.randrange(1 << 32)]) for _ in range(L)]
    for rand in (rng.randrange(O.R_MOD), O.R_MOD - 7, 0):
        table = C.create_string_buffer((1 + 2 * L) * 32)
        arr = (C.c_uint32 * L)(*freq)
        assert device._L.ug_fr_lookup_table(device._h, O.to_le(rand), arr, L, table) == 0
        assert O.from_le(table.raw[:32]) == rand
        for i in range(L):
            exp = O.lookup_row(i, freq[i], rand * (1 << 256) % O.R_MOD)
            got = (O.from_le(table.raw[32 * (1 + i):32 * (2 + i)]), O.from_le(table.raw[32 * (1 + L + i):32 * (2 + L + i)]))
            assert got == exp, (rand, i)
    zkey, uwtns, info = synth.build_ultra_circuit(device, 11)
    off, sz = O.section(uwtns, "wtns", 4)
    big = bytearray(uwtns)
    for k in range(0, sz // 4, 3):
        big[off + 4 * k:off + 4 * k + 4] = struct.pack("<I", rng.choice([1 << 31, (1 << 32) - 1, rng.randrange(1 << 31, 1 << 32)]))
    big = bytes(big)
    rk, r, s = bytes(range(1, 32)), bytes(range(40, 71)), bytes(range(80, 111))
    ug.set_test_blinding(rk + r + s)
    try:
        got = ug.ultra_groth_prover(zkey, big)
    finally:
        ug.set_test_blinding(b"")
    assert got == O.ultra_groth_prove(zkey, big, *(int.from_bytes(b, "little") for b in (rk, r, s)))


# ---- size-independent properties at full size ------------------------------------------------------------------

def test_full_size_g1_msm_in_the_exponent(device):
    """sum s_i P_i with P_i = (seed + i) G equals (sum s_i (seed + i) mod r) G, at the full configs[2] size"""
    from ultragroth_amd import synth
    n = (1 << FULL_LOG) - 1
    seed = synth.SEEDS["A"]
    pts = synth.synth_points(device, n, seed)
    sc = synth.scalars(n, "U", 99)
    b = device.bases(pts, n)
    del pts
    v = device.dvec(n, sc.tobytes())
    got = device.msm(b, device.schedule(v, 0, n))
    k = O.fr_dot_walk(sc.tobytes(), n, seed)
    assert got == O.g1_mul(synth.g1_generator_record(), k)
    # linearity / sharding property: MSM over two halves adds up to the whole
    half = n // 2
    lo = device.msm(b, device.schedule(v, 0, half))
    hi = device.msm(b, device.schedule(v, half, n - half))
    assert O.g1_add(lo, hi) == got
    # the same sum through the fixed-base window tables of the cost-model width (12 tables of 2^24 points at c = 22)
    del b
    c = device.table_window(n)
    bt = device.bases(synth.synth_points(device, n, seed), n, table_c=c)
    assert device.msm(bt, device.schedule(v, 0, n, table_c=c)) == got


def test_full_size_g2_msm_in_the_exponent(device):
    from ultragroth_amd import synth
    n = (1 << max(FULL_LOG - 2, 10)) - 1
    seed = synth.SEEDS["B2"]
    pts = synth.synth_points(device, n, seed, g2=True)
    sc = synth.scalars(n, "C", 98)
    b = device.bases(pts, n, g2=True)
    del pts
    v = device.dvec(n, sc.tobytes())
    got = device.msm(b, device.schedule(v, 0, n), g2=True)
    assert got == O.g2_mul(synth.g2_generator_record(), O.fr_dot_walk(sc.tobytes(), n, seed))
    del b
    c = device.table_window(n)
    bt = device.bases(synth.synth_points(device, n, seed, g2=True), n, g2=True, table_c=c)
    assert device.msm(bt, device.schedule(v, 0, n, table_c=c), g2=True) == got


def test_full_size_ntt_round_trip_and_delta(device):
    logn = FULL_LOG
    n = 1 << logn
    rng = np.random.Generator(np.random.PCG64(5))
    x = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    x[:, 3] &= np.uint64((1 << 60) - 1)
    data = x.tobytes()
    fwd = device.ntt(data, logn)
    assert device.ntt(fwd, logn, inverse=True) == data
    # transform of the delta at index 1 is (omega^k)_k: spot-check against the oracle's root
    R = 1 << 256
    delta = bytes(32) + O.to_le(R % O.R_MOD) + bytes(32 * (n - 2))
    out = device.ntt(delta, logn)
    w = O.root_of_unity(logn) * pow(R, -1, O.R_MOD) % O.R_MOD
    for k in (0, 1, 2, 12345, n // 2, n - 1):
        assert O.mont_decode(out[32 * k:32 * k + 32], O.R_MOD) == pow(w, k, O.R_MOD)


def test_cli_prover_end_to_end(tmp_path, zkey, wtns, vkey):
    """`prover <zkey> <wtns> <proof.json> <public.json>` (the reference's CLI contract, src/main_prover.cpp) with OS
    entropy for the blinding: the files it writes pass the reference's acceptance test (pairing check)"""
    import subprocess
    from oracle import pairing
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "ultragroth_amd", "csrc", "prover")
    golden = os.path.join(root, "tests", "golden")
    proof_path, public_path = str(tmp_path / "proof.json"), str(tmp_path / "public.json")
    r = subprocess.run([exe, os.path.join(golden, "circuit_final.zkey"), os.path.join(golden, "witness.wtns"), proof_path, public_path],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    proof, pub = open(proof_path).read(), open(public_path).read()
    assert "\0" not in proof and pub == '["7713112592372404476342535432037683616424591277138491596200192981572885523208"]'
    assert pairing.groth16_verify(vkey, pub, proof)
    # the reference's CI (.github/workflows/build.yml:69-81): `verifier` accepts the files, and rejects them after public[0] -= 1
    vk_path = os.path.join(golden, "verification_key.json")
    ver = os.path.join(root, "ultragroth_amd", "csrc", "verifier")
    r = subprocess.run([ver, vk_path, public_path, proof_path], capture_output=True, text=True)
    assert r.returncode == 0 and r.stderr == "Result: Valid proof\n"
    bad = json.loads(pub)
    bad[0] = str(int(bad[0]) - 1)
    bad_path = str(tmp_path / "public_bad.json")
    open(bad_path, "w").write(json.dumps(bad))
    r = subprocess.run([ver, vk_path, bad_path, proof_path], capture_output=True, text=True)
    assert r.returncode == 1 and r.stderr == "Result: Invalid proof\n"
    # wrong argument count and unreadable file: exit code 1 and the reference's messages
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 1 and "Usage: prover <circuit.zkey> <witness.wtns> <proof.json> <public.json>" in r.stderr
    r = subprocess.run([exe, "/nonexistent.zkey", "x", "y", "z"], capture_output=True, text=True)
    assert r.returncode == 1 and r.stderr.startswith("Error: ")
    # the UltraGroth CLI rejects a Groth16 key with the reference's message
    r = subprocess.run([exe + "_ultra_groth", os.path.join(golden, "circuit_final.zkey"), os.path.join(golden, "witness.wtns"),
                        proof_path, public_path], capture_output=True, text=True)
    assert r.returncode == 1 and "zkey file is not ultragroth" in r.stderr


@pytest.mark.parametrize("devices", ["0,0", "0,0,0,0", "0,0,0,0,0"])
def test_cli_and_api_on_several_devices(device, tmp_path, devices):
    """ULTRAGROTH_DEVICES: the reference's own entry points -- `prover <zkey> <wtns> <proof.json> <public.json>`
    (src/main_prover.cpp:17-85) and groth16_prover_create / _prove (src/prover.cpp:681-723) -- shard one proof over the
    listed devices inside the library (one host thread and context pair per rank, peer copies of the evaluation-vector
    slices, partial sums added on the host): no Python, no torch. Rehearsed with one device listed several times; the files
    and the API's strings equal the single-device proof byte for byte (fixed blinding through the test hook). Five ranks:
    the domain does not split evenly, the slices differ in length; two: a rank runs two chains."""
    import subprocess
    import ultragroth_amd as ug
    from ultragroth_amd import synth
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "ultragroth_amd", "csrc", "prover")
    zkey, wtns, info = synth.build_circuit(device, 14, mix="C", seed=0x5EED0A00)
    r, s = fixed_rs()
    exp = O.groth16_prove(zkey, wtns, int.from_bytes(r, "little"), int.from_bytes(s, "little"))[:2]
    zpath, wpath = tmp_path / "c.zkey", tmp_path / "w.wtns"
    zpath.write_bytes(bytes(zkey)); wpath.write_bytes(wtns)
    env = dict(os.environ, ULTRAGROTH_TEST_HOOKS="1", ULTRAGROTH_TEST_BLINDING=(r + s).hex())
    for dev in (None, devices):
        e = dict(env)
        if dev:
            e["ULTRAGROTH_DEVICES"] = dev
        out = subprocess.run([exe, str(zpath), str(wpath), str(tmp_path / "proof.json"), str(tmp_path / "public.json")],
                             capture_output=True, text=True, timeout=300, env=e)
        assert out.returncode == 0, out.stderr
        assert ((tmp_path / "proof.json").read_text(), (tmp_path / "public.json").read_text()) == exp, dev
    # the created-prover API in this process: the variable is read at create
    os.environ["ULTRAGROTH_DEVICES"] = devices
    try:
        with ug.Groth16Prover(zkey) as p:
            for _ in range(2):
                ug.set_test_blinding(r + s)
                try:
                    assert p.prove(wtns) == exp
                finally:
                    ug.set_test_blinding(b"")
            with pytest.raises(ug.ProverError) as err:
                p.prove(wtns[:-32])
            assert "Invalid" in err.value.message or "section" in err.value.message.lower()
            ug.set_test_blinding(r + s)
            try:
                assert p.prove(wtns) == exp
            finally:
                ug.set_test_blinding(b"")
            # (round 5) the witness on a node: by default a chain rank collects the other ranks' slices from their HBM (peer copies);
            # ULTRAGROTH_WITNESS_GATHER=0 keeps the older form, every chain rank uploading the rest itself -- read per proof
            os.environ["ULTRAGROTH_WITNESS_GATHER"] = "0"
            ug.set_test_blinding(r + s)
            try:
                assert p.prove(wtns) == exp
            finally:
                ug.set_test_blinding(b"")
                del os.environ["ULTRAGROTH_WITNESS_GATHER"]
            ug.set_test_blinding(r + s)
            try:
                assert p.prove(wtns) == exp
            finally:
                ug.set_test_blinding(b"")
        os.environ["ULTRAGROTH_DEVICES"] = "0,x"
        with pytest.raises(ug.ProverError, match="ULTRAGROTH_DEVICES"):
            ug.Groth16Prover(zkey)
    finally:
        del os.environ["ULTRAGROTH_DEVICES"]


@pytest.mark.parametrize("devices", ["0,0", "0,0,0,0,0"])
def test_ultragroth_api_on_several_devices(device, devices):
    """ULTRAGROTH_DEVICES for the UltraGroth entry points (src/prover.h:89-96,140-151): the round commitment's parts added on the
    host, rank 0 closes the round, every rank derives the challenge and completes its witness, final round as the Groth16 phases;
    == the oracle's proof, twice on one object, also after a refused witness"""
    import ultragroth_amd as ug
    from ultragroth_amd import synth
    zkey, uwtns, info = synth.build_ultra_circuit(device, 13)
    rk, r, s = bytes(range(1, 32)), bytes(range(40, 71)), bytes(range(80, 111))
    exp = O.ultra_groth_prove(zkey, uwtns, *(int.from_bytes(b, "little") for b in (rk, r, s)))
    os.environ["ULTRAGROTH_DEVICES"] = devices
    try:
        with ug.UltraGrothProver(zkey) as p:
            for k in range(4):
                if k == 1:
                    with pytest.raises(ug.ProverError):
                        p.prove(uwtns[:-40])
                if k == 3:      # (round 5) every rank uploads the whole witness itself instead of collecting its peers' slices
                    os.environ["ULTRAGROTH_WITNESS_GATHER"] = "0"
                ug.set_test_blinding(rk + r + s)
                try:
                    assert p.prove(uwtns) == exp
                finally:
                    ug.set_test_blinding(b"")
    finally:
        del os.environ["ULTRAGROTH_DEVICES"]
        os.environ.pop("ULTRAGROTH_WITNESS_GATHER", None)


@pytest.mark.parametrize("log_domain,n_public", [(2, 1), (3, 0), (5, 3), (7, 1), (10, 0)])
def test_tiny_circuits_and_public_counts(device, log_domain, n_public):
    """smallest domains (single NTT pass, one segment, windows wider than the scalar count) and 0 / several public
    signals: proof and public.json ("null" when empty, as nlohmann dumps an empty json) equal the oracle's"""
    import ultragroth_amd as ug
    from ultragroth_amd import synth
    zkey, wtns, info = synth.build_circuit(device, log_domain, mix="C", n_public=n_public, seed=0x5EED0100 + log_domain)
    r, s = fixed_rs()
    ug.set_test_blinding(r + s)
    try:
        proof, pub = ug.groth16_prover(zkey, wtns)
    finally:
        ug.set_test_blinding(b"")
    exp = O.groth16_prove(zkey, wtns, int.from_bytes(r, "little"), int.from_bytes(s, "little"))
    assert (proof, pub) == (exp[0], exp[1])
    if n_public == 0:
        assert pub == "null"
    else:
        assert len(json.loads(pub)) == n_public


@pytest.mark.parametrize("tables", ["1", "0"])
@pytest.mark.parametrize("log_domain,mix", [(15, "C"), (17, "U")])
def test_created_prover_with_and_without_window_tables(device, monkeypatch, tables, log_domain, mix):
    """a created prover builds fixed-base window tables (ULTRAGROTH_TABLES unset or 1) or runs the classic windows (0):
    both give the oracle's proof, and proving twice on one prover gives the same proof"""
    import ultragroth_amd as ug
    from ultragroth_amd import synth
    monkeypatch.setenv("ULTRAGROTH_TABLES", tables)
    zkey, wtns, info = synth.build_circuit(device, log_domain, mix=mix, seed=0x5EED0200 + log_domain)
    r, s = fixed_rs()
    exp = O.groth16_prove(zkey, wtns, int.from_bytes(r, "little"), int.from_bytes(s, "little"))
    free_before, _ = device.mem_info()
    with ug.Groth16Prover(zkey) as p:
        for k in range(3):
            if k == 2:
                p.tables_ready(wait=True)        # (round 5: create returns before the tables exist, their memory included)
            ug.set_test_blinding(r + s)
            try:
                proof, pub = p.prove(wtns)
            finally:
                ug.set_test_blinding(b"")
            assert (proof, pub) == (exp[0], exp[1])
        free_created, _ = device.mem_info()
    # the tables are really there (or really absent): 5 G1-sized sets + one G2 set, c = 16 -> 15 extra tables each
    n = info["nVars"]
    extra = 15 * n * (64 * 4 + 128)
    used = free_before - free_created
    if tables == "1":
        assert used > extra
    else:
        assert used < extra


def test_ultragroth_created_prover_with_window_tables(device):
    """UltraGroth on a created prover at 2^17: the witness, round (C1), final (C2) and H groups each get tables of
    their own width (C1 has ~2^15 points, the others ~2^17); proof == oracle, twice"""
    import ultragroth_amd as ug
    from ultragroth_amd import synth
    zkey, uwtns, info = synth.build_ultra_circuit(device, 17)
    assert info["nC1"] >= 1 << 14
    rk, r, s = bytes(range(1, 32)), bytes(range(40, 71)), bytes(range(80, 111))
    exp = O.ultra_groth_prove(zkey, uwtns, int.from_bytes(rk, "little"), int.from_bytes(r, "little"), int.from_bytes(s, "little"))
    with ug.UltraGrothProver(zkey) as p:
        for _ in range(2):
            ug.set_test_blinding(rk + r + s)
            try:
                got = p.prove(uwtns)
            finally:
                ug.set_test_blinding(b"")
            assert got == exp


def test_ultragroth_prover_object_from_three_host_threads(device):
    """ultra_groth_prover_prove from three threads on one prover object: the .uwtns of a waiting call (signals and the
    four lookup sections) is staged into the second witness buffer while another proof runs and patches ITS copy of the
    witness on the device; the three blinding draws of a proof happen inside its turn. Every proof == oracle."""
    import threading
    import ultragroth_amd as ug
    from ultragroth_amd import synth
    zkey, uwtns, info = synth.build_ultra_circuit(device, 15)
    rk, r, s = bytes(range(1, 32)), bytes(range(40, 71)), bytes(range(80, 111))
    exp = O.ultra_groth_prove(zkey, uwtns, int.from_bytes(rk, "little"), int.from_bytes(r, "little"), int.from_bytes(s, "little"))
    failures = []
    with ug.UltraGrothProver(zkey) as p:
        def caller(k):
            try:
                for it in range(6):
                    if p.prove(uwtns) != exp:
                        failures.append("thread %d proof %d differs" % (k, it))
            except Exception as e:                              # noqa: BLE001 (reported below)
                failures.append("thread %d: %r" % (k, e))
        ug.set_test_blinding(rk + r + s)
        try:
            threads = [threading.Thread(target=caller, args=(k,)) for k in range(3)]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
        finally:
            ug.set_test_blinding(b"")
    assert not failures, failures


def test_ranges_above_the_schedule_limit_are_proved_in_pieces(device, monkeypatch):
    """a schedule holds at most 2^26 scalars, so the reference's largest legal domain (2^27) is proved in pieces whose
    partial sums are added; ULTRAGROTH_MAX_RANGE lowers the limit so that a 2^13 circuit takes that path (7 pieces)"""
    import ultragroth_amd as ug
    from ultragroth_amd import synth
    zkey, wtns, info = synth.build_circuit(device, 13, mix="C", seed=0x5EED0300)
    r, s = fixed_rs()
    exp = O.groth16_prove(zkey, wtns, int.from_bytes(r, "little"), int.from_bytes(s, "little"))
    monkeypatch.setenv("ULTRAGROTH_MAX_RANGE", "1234")
    with ug.Groth16Prover(zkey) as p:
        ug.set_test_blinding(r + s)
        try:
            got = p.prove(wtns)
        finally:
            ug.set_test_blinding(b"")
    assert got == (exp[0], exp[1])


@pytest.mark.parametrize("seed", range(8))
def test_random_irregular_circuits(device, seed):
    """randomised shapes the fixed generator never produces: rows with no entry, rows only in A or only in B, rows with
    up to 40 entries, repeated (matrix, row, signal) records, coefficients above r, 0..3 public signals, domains 2^3..2^10;
    witness values include 0, 1 and r - 1. Proof and public.json equal the oracle's."""
    import ultragroth_amd as ug
    from ultragroth_amd import synth
    rng = np.random.Generator(np.random.PCG64(0xC0FFEE + seed))
    log_domain = int(rng.integers(3, 11))
    domain, nvars = 1 << log_domain, (1 << log_domain) - 1
    n_public = int(rng.integers(0, 4))
    recs = []
    for row in range(domain):
        kind = rng.integers(0, 6)
        if kind == 0:
            continue                                                   # empty row
        n_a = 0 if kind == 1 else int(rng.integers(1, 4))
        n_b = 0 if kind == 2 else int(rng.integers(1, 4))
        if kind == 5 and row % 17 == 0:
            n_a = 40                                                   # a long row
        for m, cnt in ((0, n_a), (1, n_b)):
            sigs = rng.integers(0, nvars, size=cnt)
            if cnt > 1 and rng.integers(0, 3) == 0:
                sigs[1] = sigs[0]                                      # the same record position twice
            for sg in sigs:
                recs.append((m, row, int(sg)))
    dt = np.dtype([("m", "<u4"), ("c", "<u4"), ("s", "<u4"), ("v", "<u8", (4,))], align=False)
    coefs = np.zeros(len(recs), dtype=dt)
    if recs:
        arr = np.array(recs, dtype=np.uint32)
        coefs["m"], coefs["c"], coefs["s"] = arr[:, 0], arr[:, 1], arr[:, 2]
        coefs["v"] = rng.integers(0, 1 << 63, size=(len(recs), 4), dtype=np.uint64) * np.uint64(2)   # up to 2^256: above r too
        coefs = coefs[rng.permutation(len(recs))]
    zkey, wtns, info = synth.build_circuit(device, log_domain, mix="C", seed=0x5EED0400 + seed, n_public=n_public, coefs=coefs)
    # sprinkle special witness values
    w = bytearray(wtns)
    off = O.section(wtns, "wtns", 2)[0]
    for i, val in ((3, 0), (4, 1), (5, O.R_MOD - 1)):
        if i < nvars:
            w[off + 32 * i:off + 32 * i + 32] = O.to_le(val)
    wtns = bytes(w)
    r, s = fixed_rs()
    exp = O.groth16_prove(zkey, wtns, int.from_bytes(r, "little"), int.from_bytes(s, "little"))
    ug.set_test_blinding(r + s)
    try:
        got = ug.groth16_prover(zkey, wtns)
    finally:
        ug.set_test_blinding(b"")
    assert got == (exp[0], exp[1])


@pytest.mark.parametrize("world,split_h", [(1, False), (3, False), (2, True)])
def test_sharded_ultragroth_equals_oracle(device, world, split_h):
    """BASELINE configs[4] on one GPU: `world` ranks hold slices of the witness-indexed, round, final and H sections;
    the round commitment parts are added, one rank closes the round, every rank applies the commitment (challenge +
    lookup signals) and runs its slices of the final MSMs; with split_h the NTT chains are split over the ranks too"""
    import torch
    import ultragroth_amd as ug
    from ultragroth_amd import synth
    zkey, uwtns, info = synth.build_ultra_circuit(device, 13)
    rk, r, s = bytes(range(1, 32)), bytes(range(40, 71)), bytes(range(80, 111))
    exp = O.ultra_groth_prove(zkey, uwtns, int.from_bytes(rk, "little"), int.from_bytes(r, "little"), int.from_bytes(s, "little"))
    ranks = [ug.ShardedUltraGrothProver(zkey, 0, k, world) for k in range(world)]
    ug.set_test_blinding(rk + r + s)
    try:
        for p in ranks:
            p.load_witness(uwtns)
        total = bytes(64)
        for p in ranks:
            total = ug.ShardedUltraGrothProver.add_records(total, p.round_commit())
        commitment = ranks[0].round_finish(total)
        for p in ranks:
            p.apply_commitment(commitment)
        parts = [p.run_witness_msm() for p in ranks]
        n_dom = info["domainSize"]
        if split_h:
            sl = n_dom // world
            full = torch.empty((3, n_dom, 32), dtype=torch.uint8, device="cuda")
            for k in range(3):
                ranks[k % world].hpoly_chain(k, full[k].data_ptr())
            torch.cuda.synchronize()
            for q, p in enumerate(ranks):
                bufs = [full[k, q * sl:(q + 1) * sl].contiguous() for k in range(3)]
                torch.cuda.synchronize()
                p.hpoly_combine(*(b.data_ptr() for b in bufs))
        else:                                   # every rank forms the whole h (three chains into scratch, combine its slice)
            full = torch.empty((3, n_dom, 32), dtype=torch.uint8, device="cuda")
            for q, p in enumerate(ranks):
                for k in range(3):
                    p.hpoly_chain(k, full[k].data_ptr())
                first, cnt, _ = p.h_range()
                bufs = [full[k, first:first + cnt].contiguous() for k in range(3)]
                torch.cuda.synchronize()
                p.hpoly_combine(*(b.data_ptr() for b in bufs))
        acc = None
        for q, p in enumerate(ranks):
            part = parts[q][:320] + p.run_h_msm()[320:384]
            acc = part if acc is None else ug.ShardedGroth16Prover.add_partials(acc, part)
        got = ranks[0].finish(acc)
    finally:
        ug.set_test_blinding(b"")
    assert got == exp
    with pytest.raises(ug.ProverError, match="did not close the round"):
        if world > 1:
            ranks[1].finish(acc)
        else:
            raise ug.ProverError(1, "finish on a rank that did not close the round")
    for p in ranks:
        p.close()


@pytest.mark.parametrize("world,chain_first", [(2, False), (4, True)])
def test_sharded_ultragroth_queued_final_round(device, world, chain_first):
    """the final round as bench.py --ultra and the many-device UltraGroth prover drive it since round 3: every rank's witness
    products (A | B1, B2, the gathered final set) queued with ug_groth16_prover_witness_msm_begin, its H branch -- chains, combine,
    H product, on the rank's second stream -- driven meanwhile, _end afterwards; chain first or beside the products. == the oracle,
    twice on the same objects; the blocking call is refused while products are queued"""
    import torch
    import ultragroth_amd as ug
    from ultragroth_amd import synth
    zkey, uwtns, info = synth.build_ultra_circuit(device, 13)
    rk, r, s = bytes(range(1, 32)), bytes(range(40, 71)), bytes(range(80, 111))
    exp = O.ultra_groth_prove(zkey, uwtns, int.from_bytes(rk, "little"), int.from_bytes(r, "little"), int.from_bytes(s, "little"))
    ranks = [ug.ShardedUltraGrothProver(zkey, 0, k, world) for k in range(world)]
    n_dom = info["domainSize"]
    sl = n_dom // world
    full = torch.empty((3, n_dom, 32), dtype=torch.uint8, device="cuda")
    try:
        for turn in range(2):
            ug.set_test_blinding(rk + r + s)
            try:
                for p in ranks:
                    p.load_witness(uwtns)
                with pytest.raises(ug.ProverError, match="round commitment has not been applied"):
                    ranks[0].witness_msm_begin()
                total = bytes(64)
                for p in ranks:
                    total = ug.ShardedUltraGrothProver.add_records(total, p.round_commit())
                commitment = ranks[0].round_finish(total)
                for p in ranks:
                    p.apply_commitment(commitment)
                if not chain_first:
                    for p in ranks:
                        p.witness_msm_begin()
                for k in range(3):
                    ranks[k % world].hpoly_chain(k, full[k].data_ptr())
                if chain_first:
                    for p in ranks:
                        p.witness_msm_begin()
                with pytest.raises(ug.ProverError, match="queued"):
                    ranks[world - 1].run_witness_msm()
                acc = None
                for q, p in enumerate(ranks):
                    bufs = [full[k, q * sl:(q + 1) * sl].contiguous() for k in range(3)]
                    torch.cuda.current_stream().synchronize()
                    p.hpoly_combine(*(b.data_ptr() for b in bufs))
                    hpart = p.run_h_msm()
                    part = p.witness_msm_end()[:320] + hpart[320:384]
                    acc = part if acc is None else ug.ShardedGroth16Prover.add_partials(acc, part)
                got = ranks[0].finish(acc)
            finally:
                ug.set_test_blinding(b"")
            assert got == exp, turn
    finally:
        for p in ranks:
            p.close()


@pytest.mark.parametrize("world", [3, 4])
def test_sharded_ultragroth_from_slices(device, world):
    """ug_ultra_groth_prover_create_sharded_slices: every rank from the header section and ITS slices of the six point sections and
    the two index lists only (byte counts checked), ranks beyond the three chain ranks without a coefficient matrix -- the
    generator's slices are the zkey's sections cut at ug_ultra_groth_shard_ranges -- and the proof of the sliced ranks == the proof of
    ranks made from the whole zkey == the oracle"""
    import torch
    import ultragroth_amd as ug
    from ultragroth_amd import synth
    log_domain = 13
    zkey, uwtns, info = synth.build_ultra_circuit(device, log_domain)
    assert synth.build_ultra_witness(log_domain) == uwtns
    si = synth.ultra_info(log_domain)
    assert all(si[k] == info[k] for k in ("nVars", "domainSize", "nC1", "nC2", "nPublic"))
    rk, r, s = bytes(range(1, 32)), bytes(range(40, 71)), bytes(range(80, 111))
    exp = O.ultra_groth_prove(zkey, uwtns, int.from_bytes(rk, "little"), int.from_bytes(r, "little"), int.from_bytes(s, "little"))
    n_dom, nv = info["domainSize"], info["nVars"]
    sec = lambda k: _sec_of(zkey, k)
    ranks = []
    for k in range(world):
        rg = ug.ShardedUltraGrothProver.shard_ranges(nv, n_dom, info["nC1"], info["nC2"], k, world)
        header, coefs, slices = synth.build_ultra_circuit_slices(device, log_domain, rg, with_coefs=(k < 3))
        (w0, w1), (r0, r1), (f0, f1), (h0, h1) = rg
        assert header == sec(2)
        for got_sl, (sid, rec, lo, hi) in zip(slices, ((5, 64, w0, w1), (6, 64, w0, w1), (7, 128, w0, w1), (8, 64, r0, r1), (9, 64, f0, f1),
                                                       (12, 64, h0, h1), (10, 4, r0, r1), (11, 4, f0, f1))):
            assert bytes(got_sl) == sec(sid)[lo * rec:hi * rec], sid
        ranks.append(ug.ShardedUltraGrothProver.from_slices(header, coefs, info.get("nCoefs", 4 * n_dom), slices, 0, k, world, public_size=86))
        if k == 1:
            short = list(slices)
            short[6] = bytes(short[6])[:-4]
            with pytest.raises(ug.ProverError, match="round_indexes slice is shorter"):
                ug.ShardedUltraGrothProver.from_slices(header, coefs, 4 * n_dom, short, 0, k, world, public_size=86)
    full = torch.empty((3, n_dom, 32), dtype=torch.uint8, device="cuda")
    try:
        ug.set_test_blinding(rk + r + s)
        try:
            total = bytes(64)
            for p in ranks:
                p.load_witness(uwtns)
                total = ug.ShardedUltraGrothProver.add_records(total, p.round_commit())
            commitment = ranks[0].round_finish(total)
            for p in ranks:
                p.apply_commitment(commitment)
            for k in range(3):
                ranks[k % world].hpoly_chain(k, full[k].data_ptr())
            if world > 3:
                with pytest.raises(ug.ProverError, match="created without the coefficient matrix"):
                    ranks[3].hpoly_chain(0, full[0].data_ptr())
            acc = None
            for q, p in enumerate(ranks):
                p.witness_msm_begin()
                first, cnt, _ = p.h_range()
                bufs = [full[k, first:first + cnt].contiguous() for k in range(3)]
                torch.cuda.current_stream().synchronize()
                p.hpoly_combine(*(b.data_ptr() for b in bufs))
                hpart = p.run_h_msm()
                part = p.witness_msm_end()[:320] + hpart[320:384]
                acc = part if acc is None else ug.ShardedGroth16Prover.add_partials(acc, part)
            got = ranks[0].finish(acc)
        finally:
            ug.set_test_blinding(b"")
        assert got == exp
    finally:
        for p in ranks:
            p.close()


def _sec_of(buf, sid):
    off, sz = O.section(buf, "zkey", sid)
    return bytes(buf[off:off + sz])


def test_created_prover_at_2_20_bit_exact(device):
    """the largest size the oracle still proves in seconds: 2^20 constraints through a created prover (window tables of
    the cost-model width, batched witness MSMs, three-pass NTTs), uniform scalars, twice with different witnesses"""
    import ultragroth_amd as ug
    from ultragroth_amd import synth
    zkey, wtns, info = synth.build_circuit(device, 20, mix="U", seed=0x5EED0500)
    r, s = fixed_rs()
    with ug.Groth16Prover(zkey) as p:
        for k, w in enumerate((wtns, None)):
            if w is None:                                           # second witness: the same values rotated by one limb
                body = np.frombuffer(wtns, dtype=np.uint8).copy()
                off = O.section(wtns, "wtns", 2)[0]
                vals = body[off:].reshape(-1, 4, 8)
                vals[1:, :3] = np.roll(vals[1:, :3], 1, axis=1)     # top limb untouched: values stay below r
                w = body.tobytes()
            ug.set_test_blinding(r + s)
            try:
                got = p.prove(w)
            finally:
                ug.set_test_blinding(b"")
            exp = O.groth16_prove(zkey, w, int.from_bytes(r, "little"), int.from_bytes(s, "little"))
            assert got == (exp[0], exp[1]), "witness %d" % k


def test_prove_resident_equals_prove(device, zkey, wtns):
    """ug_groth16_prover_prove_resident (bench.py's timed step): the witness is loaded once, every call is a whole proof with
    fresh blinding draws -- byte for byte what groth16_prover_prove returns for the same draws; refused before a witness is there"""
    import ultragroth_amd as ug
    r, s = fixed_rs()
    exp = O.groth16_prove(zkey, wtns, int.from_bytes(r, "little"), int.from_bytes(s, "little"))[:2]
    with ug.Groth16Prover(zkey) as p:
        with pytest.raises(ug.ProverError, match="no witness loaded"):
            p.prove_resident()
        p.load_witness(wtns)
        for _ in range(3):
            ug.set_test_blinding(r + s)
            try:
                assert p.prove_resident() == exp
            finally:
                ug.set_test_blinding(b"")
        a, b = p.prove_resident(), p.prove_resident()              # OS entropy: two different, valid-looking proofs of one witness
        assert a != b and a[1] == exp[1]
        import ctypes as C                                         # a short proof buffer: the reference's error and code
        psz, qsz = C.c_ulonglong(10), C.c_ulonglong(4096)
        err = C.create_string_buffer(256)
        rc = ug.load().ug_groth16_prover_prove_resident(p._h, C.create_string_buffer(10), C.byref(psz), C.create_string_buffer(4096),
                                                        C.byref(qsz), err, 255)
        assert rc == ug.PROVER_ERROR_SHORT_BUFFER and err.value == b"Proof buffer is too short. Minimum size: 810, actual size: 10"


def test_overlap_mode_is_bit_exact(device, monkeypatch):
    """ULTRAGROTH_OVERLAP=1: the H branch (mat-vec, NTT chains, h schedule, H MSM) runs on a second stream from a second
    host thread beside the witness MSMs; same proof, also with tables forced for a one-shot call (ULTRAGROTH_TABLES=2)"""
    import ultragroth_amd as ug
    from ultragroth_amd import synth
    zkey, wtns, info = synth.build_circuit(device, 15, mix="U", seed=0x5EED0600)
    r, s = fixed_rs()
    exp = O.groth16_prove(zkey, wtns, int.from_bytes(r, "little"), int.from_bytes(s, "little"))
    with ug.Groth16Prover(zkey) as p:
        for mode in ("1", "2", "1", "0", "2"):      # 2: the G2 product first, the H branch's preparation beside it, its MSM last
            monkeypatch.setenv("ULTRAGROTH_OVERLAP", mode)
            ug.set_test_blinding(r + s)
            try:
                assert p.prove(wtns) == (exp[0], exp[1]), mode
            finally:
                ug.set_test_blinding(b"")
    monkeypatch.setenv("ULTRAGROTH_OVERLAP", "1")
    monkeypatch.setenv("ULTRAGROTH_TABLES", "2")
    ug.set_test_blinding(r + s)
    try:
        assert ug.groth16_prover(zkey, wtns) == (exp[0], exp[1])
    finally:
        ug.set_test_blinding(b"")


def test_one_prover_object_from_three_host_threads(device):
    """The reference's prover keeps no per-proof state, so callers may prove from several threads on one object
    (src/prover.cpp:341-391 allocates everything per call). Here the object owns the device buffers: calls take turns on
    the device, and the witness of a waiting call is staged into the second witness buffer meanwhile (Groth16Prover::
    proveTurn). Three threads, three different witnesses, eight proofs each, every one bit-exact -- a witness staged
    into a buffer a running proof still reads, or public signals taken from the wrong call, would show here. Sized so
    that the copy takes one staging lane (2^13 constraints) and, second round, all of them (2^20: 32 MiB). The phase
    calls of a sharded rank (load_witness / run / finish) stage the same way: test_sharded_* above."""
    import threading
    import ultragroth_amd as ug
    from ultragroth_amd import synth
    r, s = fixed_rs()
    ri, si = int.from_bytes(r, "little"), int.from_bytes(s, "little")
    for log_domain, rounds in ((13, 8), (20, 3)):
        zkey, wtns, info = synth.build_circuit(device, log_domain, mix="U", seed=0x5EED0700 + log_domain)
        witnesses = [wtns]
        off = O.section(wtns, "wtns", 2)[0]
        for k in (1, 2):                                        # other witnesses: the low limbs rotated (values stay below r)
            body = np.frombuffer(wtns, dtype=np.uint8).copy()
            vals = body[off:].reshape(-1, 4, 8)
            vals[1:, :3] = np.roll(vals[1:, :3], k, axis=1)
            witnesses.append(body.tobytes())
        expected = [O.groth16_prove(zkey, w, ri, si)[:2] for w in witnesses]
        assert len({e[0] for e in expected}) == 3
        failures = []
        with ug.Groth16Prover(zkey) as p:
            def caller(k):
                try:
                    for it in range(rounds):
                        got = p.prove(witnesses[k])
                        if got != expected[k]:
                            failures.append("2^%d: thread %d proof %d differs" % (log_domain, k, it))
                except Exception as e:                          # noqa: BLE001 (reported below)
                    failures.append("2^%d: thread %d: %r" % (log_domain, k, e))
            ug.set_test_blinding(r + s)
            try:
                threads = [threading.Thread(target=caller, args=(k,)) for k in range(3)]
                for t in threads:
                    t.start()
                for t in threads:
                    t.join()
            finally:
                ug.set_test_blinding(b"")
        assert not failures, failures
