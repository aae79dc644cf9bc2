#!/usr/bin/env python3
"""Per-phase times of ONE rank of a sharded Groth16 prover (rank R of W, created from its slices as bench.py does) on one
GPU: the critical path of a W-GPU run without the collectives. Phases are timed one after the other (in a real run the
chain thread runs beside the witness MSMs). Usage: python tools/phase_times.py [log_domain] [world] [rank]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                  # noqa: E402
import bench                                  # noqa: E402
import ultragroth_amd as ug                   # noqa: E402
from ultragroth_amd import synth              # noqa: E402

log_domain = int(sys.argv[1]) if len(sys.argv) > 1 else 24
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rank = int(sys.argv[3]) if len(sys.argv) > 3 else 0
torch.cuda.set_device(0)
dev = ug.Device(0)
domain = 1 << log_domain
info = dict(domainSize=domain, nVars=domain - 1, nPublic=1, nCoefs=4 * domain)
wr = bench.witness_slice(info, rank, world)
ranges = ug.ShardedGroth16Prover.shard_ranges(info["nVars"], 1, domain, rank, world, wr)
chains = [k for k in range(3) if k % world == rank]
header, coefs, slices = synth.build_circuit_slices(dev, log_domain, ranges, with_coefs=bool(chains))
wtns = synth.build_witness(log_domain, "U")
t0 = time.perf_counter()
p = ug.ShardedGroth16Prover.from_slices(header, coefs, info["nCoefs"], slices, 0, rank, world, witness_range=wr, public_size=86)
print("rank %d of %d at 2^%d: witness slice %s (%d points), chains %s, create %.2f s" % (rank, world, log_domain, wr, wr[1] - wr[0], chains, time.perf_counter() - t0))
del coefs, slices
sl = domain // world
full = torch.empty((domain, 32), dtype=torch.uint8, device="cuda")
bufs = torch.zeros((3, sl, 32), dtype=torch.uint8, device="cuda")


def timed(name, fn, acc):
    torch.cuda.synchronize()
    t = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    acc[name] = acc.get(name, 0.0) + (time.perf_counter() - t) * 1e3
    return r


for it in range(3):
    acc = {}
    timed("upload_part0", lambda: p.load_witness_part(wtns, 0), acc)
    part = timed("witness_msm", p.run_witness_msm, acc)
    if chains:
        timed("upload_part1", lambda: p.load_witness_part(wtns, 1), acc)
        for k in chains:
            timed("chain_%d" % k, lambda: p.hpoly_chain(k, full.data_ptr()), acc)
    timed("combine", lambda: p.hpoly_combine(bufs[0].data_ptr(), bufs[1].data_ptr(), bufs[2].data_ptr()), acc)
    hp = timed("h_msm", p.run_h_msm, acc)
    total = part[:320] + hp[320:384]
    timed("finish", lambda: p.finish(total), acc)
print("  ".join("%s %.2f" % kv for kv in acc.items()), " | sum %.2f ms" % sum(acc.values()))
