"""N > 1 exchange path on CPU: two gloo ranks each compute the partial MSM sums of their contiguous base-point range
(with the oracle standing in for the device), all-gather the 384-byte partial records and add them with the
product's ug_groth16_partials_add -- the same code bench.py runs over RCCL. The sum must equal the unsharded sums."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle as O
    import ultragroth_amd as ug
    zkey = open(os.path.join(ROOT, "tests", "golden", "circuit_final.zkey"), "rb").read()
    wtns = open(os.path.join(ROOT, "tests", "golden", "witness.wtns"), "rb").read()
    info = O.zkey_info(zkey)
    M, N, P = info["nVars"], info["domainSize"], info["nPublic"]
    sec = lambda sid: zkey[O.section(zkey, "zkey", sid)[0]: sum(O.section(zkey, "zkey", sid))]
    w = wtns[O.section(wtns, "wtns", 2)[0]: sum(O.section(wtns, "wtns", 2))]
    h = O.hpoly(sec(4)[4:], info["nCoefs"], w, M, N)
    lo, hi = M * rank // world, M * (rank + 1) // world            # same split as Groth16Prover (prover_api.cpp)
    hlo, hhi = N * rank // world, N * (rank + 1) // world
    shift = P + 1
    nC = M - P - 1
    clo, chi = min(max(lo - shift, 0), nC), min(max(hi - shift, 0), nC)
    part = O.g1_msm(sec(5)[64 * lo:64 * hi], w[32 * lo:32 * hi], hi - lo)
    part += O.g1_msm(sec(6)[64 * lo:64 * hi], w[32 * lo:32 * hi], hi - lo)
    part += O.g2_msm(sec(7)[128 * lo:128 * hi], w[32 * lo:32 * hi], hi - lo)
    part += O.g1_msm(sec(8)[64 * clo:64 * chi], w[32 * (clo + shift):32 * (chi + shift)], chi - clo)
    part += O.g1_msm(sec(9)[64 * hlo:64 * hhi], h[32 * hlo:32 * hhi], hhi - hlo)
    mine = torch.frombuffer(bytearray(part), dtype=torch.uint8)
    allp = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(allp, mine)
    total = bytes(allp[0].numpy())
    for other in allp[1:]:
        total = ug.ShardedGroth16Prover.add_partials(total, bytes(other.numpy()))
    if rank == 0:
        q.put(total)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_partials_sum_to_the_unsharded_result(world, zkey, wtns):
    import oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + world + (os.getpid() % 200)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    total = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    r, s = 5, 7
    _, _, raw = O.groth16_prove(zkey, wtns, r, s, want_raw=True)
    assert total == raw


def _commit_worker(rank, world, port, q):
    """the UltraGroth round-commitment exchange of bench.py --ultra: 64-byte parts all-gathered and added on every rank,
    the closing rank's record broadcast to the others"""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle as O
    import ultragroth_amd as ug
    zkey = open(os.path.join(ROOT, "tests", "golden", "circuit_final.zkey"), "rb").read()
    sec = lambda sid: zkey[O.section(zkey, "zkey", sid)[0]: sum(O.section(zkey, "zkey", sid))]
    pts = sec(9)
    n = 64
    lo, hi = n * rank // world, n * (rank + 1) // world
    sc = b"".join(O.to_le(1000003 * (i + 1)) for i in range(n))
    part = O.g1_msm(pts[64 * lo:64 * hi], sc[32 * lo:32 * hi], hi - lo) if hi > lo else bytes(64)
    mine = torch.frombuffer(bytearray(part), dtype=torch.uint8)
    allp = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(allp, mine)
    total = bytes(64)
    for other in allp:
        total = ug.ShardedUltraGrothProver.add_records(total, bytes(other.numpy()))
    closing = torch.frombuffer(bytearray(total if rank == 0 else bytes(64)), dtype=torch.uint8)
    dist.broadcast(closing, src=0)
    assert bytes(closing.numpy()) == total                      # every rank derived the same sum on its own
    if rank == world - 1:
        q.put(total)
    dist.barrier()
    dist.destroy_process_group()


def test_round_commitment_exchange(zkey):
    import oracle as O
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29800 + (os.getpid() % 150)
    procs = [ctx.Process(target=_commit_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    total = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    pts = zkey[O.section(zkey, "zkey", 9)[0]: sum(O.section(zkey, "zkey", 9))]
    sc = b"".join(O.to_le(1000003 * (i + 1)) for i in range(64))
    assert total == O.g1_msm(pts[:64 * 64], sc, 64)


def _scatter_worker(rank, world, port, q):
    """bench.py's exchange of the evaluation slices when the ranks' h ranges are UNEVEN (the chain ranks of a bucket-class layout
    hold none): equal padded pieces from every rank's first element on (bench.padded_h_slices), three roots"""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    import ultragroth_amd as ug
    n_dom = 1 << 10
    lays = [ug.ShardedGroth16Prover.shard_layout(n_dom - 1, 1, n_dom, r, world, 1) for r in range(world)]
    h_first, sl = bench.padded_h_slices([L.h for L in lays])
    ok = True
    for k in range(3):
        src = k % world
        full = None
        if rank == src:                                  # the chain's vector, padded behind its end
            full = (torch.arange((n_dom + sl) * 32, dtype=torch.int64) * (k + 3) % 251).to(torch.uint8).reshape(n_dom + sl, 32)
        out = torch.empty((sl, 32), dtype=torch.uint8)
        dist.scatter(out, [full[h_first[r]:h_first[r] + sl].contiguous() for r in range(world)] if rank == src else None, src=src)
        h0, h1 = lays[rank].h
        exp = (torch.arange((n_dom + sl) * 32, dtype=torch.int64) * (k + 3) % 251).to(torch.uint8).reshape(n_dom + sl, 32)[h0:h1]
        ok = ok and bool((out[:h1 - h0] == exp).all())
    q.put((rank, ok, lays[rank].h))
    dist.barrier()
    dist.destroy_process_group()


def test_uneven_h_slices_travel_padded():
    world = 5                                            # five ranks: the three chain ranks of a class layout take no part of h
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29300 + (os.getpid() % 150)
    procs = [ctx.Process(target=_scatter_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in got)
    hs = [h for _, _, h in got]
    assert hs[0] == (0, 0) and hs[2] == (0, 0) and hs[3][0] == 0 and hs[4][1] == 1 << 10 and hs[3][1] == hs[4][0]


def test_bench_witness_slices_tile_the_witness():
    """bench.py gives the ranks that also run an NTT chain smaller witness slices: for every world size the slices are
    contiguous, ordered, cover [0, nVars) exactly, and the chain-carrying ranks get the smaller ones"""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for log_domain in (3, 10, 24):
        info = {"nVars": (1 << log_domain) - 1, "domainSize": 1 << log_domain}
        assert bench.witness_slice(info, 0, 1) is None
        for world in (2, 3, 4, 8):
            if info["domainSize"] % world:
                assert bench.witness_slice(info, 0, world) is None         # falls back to the even split
                continue
            cuts = [bench.witness_slice(info, k, world) for k in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == info["nVars"]
            for a, b in zip(cuts, cuts[1:]):
                assert a[1] == b[0] and a[0] <= a[1]
            if log_domain == 24 and world == 8:
                sizes = [hi - lo for lo, hi in cuts]
                assert max(sizes[:3]) < min(sizes[3:])
