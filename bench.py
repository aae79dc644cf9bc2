#!/usr/bin/env python3
"""bench.py -- proofs/s of the Groth16 hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--log-domain 24] [--mix U|C]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step is one pass of the prover's hot path over one witness of a seeded synthetic circuit
(ultragroth_amd/synth.py, shapes of SURVEY.md section 8d): the five MSMs and the H-polynomial block on
the device, then blinding and JSON on the host -- i.e. groth16_prover_prove on a created prover, with the
witness already resident in HBM when the timed region starts. Default workload: configs[2] of
BASELINE.json, the 2^24-constraint circuit with full G1+G2 MSMs that the 10x target is quoted on.

With N > 1 ranks the base points of every section are sharded by contiguous range (one process per GPU),
each rank computes partial sums of the five MSMs over its slice, the 384-byte partial records are
all-gathered over RCCL and added on every rank (an EC addition is not an RCCL reduction operator), and rank 0
finishes the proof. The three NTT chains of the H polynomial are taken by ranks 0..2 and their evaluation vectors
scattered slice-wise over RCCL, so each rank forms only its own slice of h. The same proof is produced at every N ("strong" scaling of one proof).

Rank 0 prints ONE JSON line. `roofline` is for the G1 bucket-accumulation kernel, measured with HIP events
on the launch stream inside the library; `cpu_baseline` times the CPU oracle (oracle/, OpenMP) on a bounded
sample. The oracle is the checker/baseline only; the timed path never touches it.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--log-domain", type=int, default=24)
    ap.add_argument("--mix", default="U", choices=["U", "C"])
    ap.add_argument("--cpu-sample-log", type=int, default=None, help="log2 domain of the CPU-baseline sample circuit")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--check", action="store_true", help="also compare the proof with the oracle (small sizes)")
    ap.add_argument("--g1-only", action="store_true", help="BASELINE.json configs[1]: G1 MSM + NTT only (B1/B2/C sets at infinity)")
    ap.add_argument("--overlap", action="store_true", help="ULTRAGROTH_OVERLAP=1: H branch on a second stream beside the witness MSMs "
                                                           "(faster, but per-kernel times and the MSM | FFT split stretch)")
    ap.add_argument("--ultra", action="store_true", help="BASELINE.json configs[4]: UltraGroth two-round prove (single GPU)")
    return ap.parse_args()


def cpu_share():
    """CPU threads this process may really use: affinity mask capped by the cgroup CPU quota (v2 or v1), if any"""
    cores = len(os.sched_getaffinity(0))
    for quota_file, period_file in (("/sys/fs/cgroup/cpu.max", None),
                                    ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us")):
        try:
            if period_file is None:
                quota, period = open(quota_file).read().split()
            else:
                quota, period = open(quota_file).read().strip(), open(period_file).read().strip()
            if quota not in ("max", "-1"):
                cores = max(1, min(cores, int(quota) // int(period)))
        except (OSError, ValueError):
            pass
    return cores


def cpu_baseline(dev, args, log_domain):
    """Oracle (plain C + OpenMP) on a bounded sample: the same generator at 2^sample, scaled linearly in N.
    The box may show more cores than its share (16 per GPU): the sample is timed with the detected share and, when
    that is larger, with 16 and 32 threads too, and the FASTEST is reported (the baseline gets every benefit)."""
    import oracle as O
    from ultragroth_amd import synth
    sample_log = args.cpu_sample_log if args.cpu_sample_log is not None else min(log_domain, 19)
    zk, wt, _ = synth.build_circuit(dev, sample_log, mix=args.mix)
    share = cpu_share()
    best = None
    for threads in sorted({share, min(share, 16), min(share, 32)}):
        O.lib.ugo_set_num_threads(threads)
        t0 = time.perf_counter()
        _, _, (msm_s, fft_s) = O.groth16_prove(zk, wt, 12345, 67890, want_timings=True)
        dt = time.perf_counter() - t0
        if best is None or dt < best[0]:
            best = (dt, msm_s, fft_s, O.lib.ugo_num_threads())
    dt, msm_s, fft_s, cores = best
    scale = float(1 << (log_domain - sample_log))
    return {
        "value": 1.0 / (dt * scale), "unit": "proofs/s", "cores": cores, "kind": "port",
        "sample": "oracle (restated rapidsnark-equivalent CPU path, OpenMP, best of 16/32/%d threads) proving the 2^%d "
                  "circuit of the same generator in %.2f s (MSM %.2f s | FFT %.2f s), scaled x%d linearly in N to 2^%d"
                  % (share, sample_log, dt, msm_s, fft_s, int(scale), log_domain),
    }


# One iFFT/twist/FFT chain costs about this fraction of ALL the witness MSMs of a proof (2^24, measured with
# tools/phase_times.py: chains a, b 7.1 ms and c 9.0 ms -- it also forms a.b -- against 113 ms; both sides grow ~linearly)
CHAIN_SHARE = (0.063, 0.063, 0.080)


def witness_slice(info, rank, world):
    """Rank k's slice of the witness-indexed sections. Chain k of the H polynomial runs on rank k mod N beside that
    rank's witness MSMs, so the ranks that carry chains get fewer points: shares s_k = base - chains_k, sum s_k = 1."""
    if world == 1 or info["domainSize"] % world:
        return None                              # even split (and no split H polynomial)
    extra = [sum(CHAIN_SHARE[c] for c in range(3) if c % world == k) for k in range(world)]
    base = (1.0 + sum(extra)) / world
    shares = [max(base - e, 0.0) for e in extra]
    tot = sum(shares)
    n = info["nVars"]
    cuts = [0]
    for k in range(world):
        cuts.append(n if k == world - 1 else min(n, int(round(n * sum(shares[:k + 1]) / tot))))
    return cuts[rank], cuts[rank + 1]


def bench_ultra(args, dev, ug, synth, torch, dist, backend, rank, world, local_rank):
    """UltraGroth (BASELINE.json configs[4]). N = 1: the whole ultra_groth_prover_prove call (witness upload, round
    commitment MSM, Keccak challenge, lookup completion, final-round MSMs + H polynomial, blinding, JSON).
    N > 1: the sharded prover -- every rank uploads the witness and commits to its slice of the round set; the 64-byte
    parts are all-gathered and added, rank 0 closes the round and broadcasts the commitment; every rank applies it and
    runs its slices of the final MSMs; the three NTT chains go to ranks 0..2 with their evaluation vectors scattered
    slice-wise, as for Groth16; the 384-byte partial blocks are all-gathered and rank 0 finishes."""
    zkey, uwtns, info = synth.build_ultra_circuit(dev, args.log_domain, mix="C")
    workload = ("ultragroth-bn254 2^%d constraints, two rounds, lookup 2^8, circom-like witness (BASELINE.json configs[4] "
                "shape; witness from host memory)" % args.log_domain)

    def line(elapsed, msm_ms, fft_ms, create_s, parallelism):
        print(json.dumps({
            "metric": "proofs/s", "value": args.steps / elapsed, "unit": "proofs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "u32x9 (29-bit limbs, 254-bit modular integers)", "data": "synthetic",
            "config": {"workload": workload, "log_domain": args.log_domain, "protocol": "ultragroth", "parallelism": parallelism},
            "msm_ms_per_proof": msm_ms / args.steps, "fft_ms_per_proof": fft_ms / args.steps, "create_s": create_s,
        }))

    if world == 1:
        t0 = time.perf_counter()
        prover = ug.UltraGrothProver(zkey)
        create_s = time.perf_counter() - t0
        for _ in range(args.warmup):
            prover.prove(uwtns)
        msm_ms = fft_ms = 0.0
        t0 = time.perf_counter()
        for _ in range(args.steps):
            prover.prove(uwtns)
            m, f, _ = prover.last_timings()
            msm_ms += m
            fft_ms += f
        return line(time.perf_counter() - t0, msm_ms, fft_ms, create_s, "one GPU")

    t0 = time.perf_counter()
    prover = ug.ShardedUltraGrothProver(zkey, local_rank, rank, world)
    create_s = time.perf_counter() - t0
    del zkey
    cuda = backend == "nccl"
    n_dom = info["domainSize"]
    split_h = n_dom % world == 0
    sl = n_dom // world if split_h else 0
    my_chains = [k for k in range(3) if k % world == rank] if split_h else list(range(3))
    fulls = {k: torch.empty((n_dom, 32), dtype=torch.uint8, device="cuda") for k in my_chains}
    bufs = torch.empty((3, max(sl, 1), 32), dtype=torch.uint8, device="cuda")

    def to_comm(b):
        t = torch.frombuffer(bytearray(b), dtype=torch.uint8)
        return t.cuda() if cuda else t

    def gather_all(b):
        mine = to_comm(b)
        allp = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allp, mine)
        return [bytes(t.cpu().numpy()) for t in allp]

    def step():
        prover.load_witness(uwtns)
        total = bytes(64)
        for part in gather_all(prover.round_commit()):
            total = ug.ShardedUltraGrothProver.add_records(total, part)
        commitment = to_comm(prover.round_finish(total) if rank == 0 else bytes(64))
        dist.broadcast(commitment, src=0)
        prover.apply_commitment(bytes(commitment.cpu().numpy()))
        part = prover.run_witness_msm()
        if split_h:
            for k in my_chains:                 # (the UltraGroth prover has one stream: its chains follow its MSMs)
                prover.hpoly_chain(k, fulls[k].data_ptr())
            for k in range(3):
                src = k % world
                if cuda:
                    dist.scatter(bufs[k], [fulls[k][q * sl:(q + 1) * sl] for q in range(world)] if rank == src else None, src=src)
                else:                           # gloo rehearsal: through host memory
                    o = torch.empty(bufs[k].shape, dtype=torch.uint8)
                    dist.scatter(o, [fulls[k][q * sl:(q + 1) * sl].cpu() for q in range(world)] if rank == src else None, src=src)
                    bufs[k].copy_(o)
            torch.cuda.synchronize()
            prover.hpoly_combine(bufs[0].data_ptr(), bufs[1].data_ptr(), bufs[2].data_ptr())
        else:                                   # the domain does not split evenly: every rank forms h itself
            for k in range(3):
                prover.hpoly_chain(k, fulls[k].data_ptr())
            first, cnt, _ = prover.h_range()
            sl_bufs = [fulls[k][first:first + cnt].contiguous() for k in range(3)]
            torch.cuda.synchronize()
            prover.hpoly_combine(*(b.data_ptr() for b in sl_bufs))
        part = part[:320] + prover.run_h_msm()[320:384]
        acc = None
        for other in gather_all(part):
            acc = other if acc is None else ug.ShardedGroth16Prover.add_partials(acc, other)
        return prover.finish(acc) if rank == 0 else None

    def barrier():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    msm_ms = fft_ms = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        m, f, _ = prover.last_timings()
        msm_ms += m
        fft_ms += f
    barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if cuda else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    chk = None
    if args.check:
        rk, r, s = bytes(range(1, 32)), bytes(range(40, 71)), bytes(range(80, 111))
        ug.set_test_blinding(rk + r + s)
        chk = step()
        ug.set_test_blinding(b"")
    if rank == 0:
        if args.check:
            import oracle as O
            zk, uw, _ = synth.build_ultra_circuit(dev, args.log_domain, mix="C")
            exp = O.ultra_groth_prove(zk, uw, int.from_bytes(bytes(range(1, 32)), "little"),
                                      int.from_bytes(bytes(range(40, 71)), "little"), int.from_bytes(bytes(range(80, 111)), "little"))
            workload += " [check: %s]" % ("bit-exact" if chk == exp else "MISMATCH")
        line(float(t.item()), msm_ms, fft_ms, create_s,
             "section-range shard x%d%s" % (world, ", H-poly chains split over ranks" if split_h else ""))
    dist.barrier()
    dist.destroy_process_group()


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if world == 1 and args.gpus > 1:
        raise SystemExit("launch N > 1 with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                         "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")

    import torch
    import ultragroth_amd as ug
    from ultragroth_amd import synth
    if args.overlap:
        os.environ["ULTRAGROTH_OVERLAP"] = "1"

    # UG_BENCH_BACKEND=gloo + UG_BENCH_ONE_DEVICE=1 rehearse the N > 1 control flow on a one-GPU box
    # (RCCL refuses two ranks on one device); the driver's multi-GPU runs use the defaults: nccl, one GPU per rank.
    backend = os.environ.get("UG_BENCH_BACKEND", "nccl")
    if os.environ.get("UG_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    dev = ug.Device(local_rank)
    log_domain = args.log_domain
    if args.ultra:
        return bench_ultra(args, dev, ug, synth, torch, dist, backend, rank, world, local_rank)
    zkey, wtns, info = synth.build_circuit(dev, log_domain, mix=args.mix, g1_only=args.g1_only)
    t0 = time.perf_counter()
    prover = ug.ShardedGroth16Prover(zkey, local_rank, rank, world, witness_range=witness_slice(info, rank, world))
    create_s = time.perf_counter() - t0
    zkey_bytes = len(zkey)
    del zkey
    t0 = time.perf_counter()
    prover.load_witness(wtns)
    upload_s = time.perf_counter() - t0

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # N > 1: the three iFFT/twist/FFT chains of the H polynomial go to ranks 0..2 (k mod N); each source rank scatters
    # the slices of its evaluation vector over RCCL, so every rank holds its slice of all three and forms its slice of h.
    split_h = dist is not None and info["domainSize"] % world == 0
    if split_h:
        n_dom = info["domainSize"]
        sl = n_dom // world
        ev_dev = "cuda"
        fulls = {k: torch.empty((n_dom, 32), dtype=torch.uint8, device=ev_dev) for k in range(3) if k % world == rank}
        bufs = torch.empty((3, sl, 32), dtype=torch.uint8, device=ev_dev)

    def scatter_slices(out, src_full, src):
        if backend == "nccl":
            dist.scatter(out, [src_full[r * sl:(r + 1) * sl] for r in range(world)] if rank == src else None, src=src)
        else:                                   # gloo rehearsal: through host memory
            o = torch.empty(out.shape, dtype=torch.uint8)
            lst = [src_full[r * sl:(r + 1) * sl].cpu() for r in range(world)] if rank == src else None
            dist.scatter(o, lst, src=src)
            out.copy_(o)

    my_chains = [k for k in range(3) if k % world == rank] if split_h else []

    def run_chains():
        # the H-polynomial branch has its own stream inside the library: it runs beside this rank's witness MSMs
        for k in my_chains:
            prover.hpoly_chain(k, fulls[k].data_ptr())

    def step():
        if split_h:
            th = None
            if my_chains:
                th = threading.Thread(target=run_chains)
                th.start()
            part = prover.run_witness_msm()
            if th is not None:
                th.join()
            for k in range(3):
                scatter_slices(bufs[k], fulls.get(k), k % world)
            torch.cuda.synchronize()
            prover.hpoly_combine(bufs[0].data_ptr(), bufs[1].data_ptr(), bufs[2].data_ptr())
            part = part[:320] + prover.run_h_msm()[320:384]
        else:
            part = prover.run()
        if dist is not None:
            mine = torch.frombuffer(bytearray(part), dtype=torch.uint8)
            if backend == "nccl":
                mine = mine.cuda()
            allp = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(allp, mine)
            total = bytes(allp[0].cpu().numpy())
            for other in allp[1:]:
                total = prover.add_partials(total, bytes(other.cpu().numpy()))
        else:
            total = part
        return prover.finish(total) if rank == 0 else None

    out = None
    for _ in range(args.warmup):
        out = step()
    prover.kernel_stats(g2=False, reset=True)
    prover.kernel_stats(g2=True, reset=True)
    msm_ms = fft_ms = 0.0
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
        m, f, _ = prover.last_timings()
        msm_ms += m
        fft_ms += f
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    chk = None
    if args.check:                              # one more step on EVERY rank (it contains collectives), fixed blinding
        ug.set_test_blinding(bytes(range(1, 32)) + bytes(range(31, 62)))
        chk = step()
        ug.set_test_blinding(b"")

    if rank == 0:
        acc_ms, launches, entries = prover.kernel_stats(g2=False)
        g2_ms, g2_launches, _ = prover.kernel_stats(g2=True)
        ws = witness_slice(info, 0, world)
        n_local = (ws[1] - ws[0]) if ws else info["nVars"] // world
        # G1 bucket accumulation: algorithmic bytes of one G1 MSM launch = 96 B per point of the slice
        # (64 B affine base + 32 B scalar, each read once)
        g1_bytes = 96.0 * n_local
        achieved = g1_bytes / (acc_ms * 1e-3) / 1e9 if acc_ms > 0 else 0.0
        # HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE + WRITE_SIZE, corrected as the microarch guide
        # says), measured separately on this workload and committed under profiles/; null when not measured
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_summary.json")))
            k = pmc.get(str(log_domain), {}).get("segment_accumulate_kernel<G1Cfg>")
            if k and world == 1 and args.mix == "U":
                traffic = k["fetch"] + k["write"]
        except Exception:
            traffic = None
        res = {
            "metric": "proofs/s", "value": args.steps / elapsed, "unit": "proofs/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u32x9 (29-bit limbs, 254-bit modular integers)",
            "data": "synthetic",
            "config": {"workload": "groth16-bn254 2^%d constraints, nVars 2^%d-1, nCoefs 4N, full G1+G2 MSM + H-poly FFT, "
                                   "scalar mix %s (BASELINE.json configs[%d] shape)" % (log_domain, log_domain, args.mix, 1 if args.g1_only else 2),
                       "log_domain": log_domain, "mix": args.mix, "overlap": bool(os.environ.get("ULTRAGROTH_OVERLAP", "0") not in ("", "0")), "parallelism": "base-range shard x%d%s" % (world, ", H-poly chains split over ranks" if split_h else "")},
            "msm_ms_per_proof": msm_ms / args.steps, "fft_ms_per_proof": fft_ms / args.steps,
            "create_s": create_s, "witness_upload_s": upload_s, "zkey_bytes": zkey_bytes,
            "host_peak_rss_gb": round(__import__("resource").getrusage(__import__("resource").RUSAGE_SELF).ru_maxrss / 1048576.0, 2),
            "roofline": {"bound": "hbm", "kernel": "segment_accumulate_kernel<G1Cfg>", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": g1_bytes,
                         "avg_launch_ms": acc_ms, "launches": launches,
                         # the bound that actually binds: v_mad_u64_u32 issue (29 T mad/s measured, tools/ubench_int.hip);
                         # one G1 mixed addition = 1467 mads, entries = (scalar, window) digits
                         "issue_bound": {"unit": "T mad/s", "peak": 29.0,
                                         "achieved": (1467.0 * entries / max(launches, 1) / (acc_ms * 1e-3) / 1e12) if acc_ms > 0 else 0.0,
                                         "frac": (1467.0 * entries / max(launches, 1) / (acc_ms * 1e-3) / 29.0e12) if acc_ms > 0 else 0.0},
                         "g2_kernel": {"avg_launch_ms": g2_ms, "launches": g2_launches,
                                       "achieved": (160.0 * n_local / (g2_ms * 1e-3) / 1e9) if g2_ms > 0 else 0.0},
                         "note": "integer-issue-bound kernel: see DESIGN.md for modmul/s against the v_mad_u64_u32 peak"},
        }
        if not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(dev, args, log_domain)
        if args.check:
            import oracle as O
            zk, wt, _ = synth.build_circuit(dev, log_domain, mix=args.mix)
            exp = O.groth16_prove(zk, wt, int.from_bytes(bytes(range(1, 32)), "little"), int.from_bytes(bytes(range(31, 62)), "little"))
            res["check"] = "bit-exact" if (chk[0], chk[1]) == (exp[0], exp[1]) else "MISMATCH"
        print(json.dumps(res))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
