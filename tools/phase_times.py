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
# argument 6: base-point ranges of the layout (world = base-point form of rounds 1-3; 1 = bucket classes over the whole witness;
# 0 = the library's choice); argument 7: scalar mix U | C
point_ranges = int(sys.argv[6]) if len(sys.argv) > 6 else world
mix = sys.argv[7] if len(sys.argv) > 7 else "U"
layout = ug.ShardedGroth16Prover.shard_layout(info["nVars"], 1, domain, rank, world, point_ranges)
wr = layout.witness
chains = layout.chains
header, coefs, slices = synth.build_circuit_slices(dev, log_domain, layout.ranges, with_coefs=bool(chains))
wtns = synth.build_witness(log_domain, mix)
t0 = time.perf_counter()
p = ug.ShardedGroth16Prover.from_slices(header, coefs, info["nCoefs"], slices, 0, rank, world, public_size=86, layout=layout)
print("rank %d of %d at 2^%d, mix %s: %s (%d points), create %.2f s" % (rank, world, log_domain, mix, layout, wr[1] - wr[0], time.perf_counter() - t0))
del coefs, slices
sl = max(layout.h[1] - layout.h[0], 1)
full = torch.empty((domain, 32), dtype=torch.uint8, device="cuda")
# (random slices, top byte small: h = a.b - c is then a vector of full-size scalars, as in a real proof -- with zeros the H
# product has no entries and takes no time)
bufs = torch.randint(0, 256, (3, sl, 32), dtype=torch.uint8, device="cuda")
bufs[:, :, 31] &= 0x0f


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

# The same rank driven as bench.py drives it since round 3: the witness products queued (witness_msm_begin), the H branch on the
# second stream meanwhile, one host thread. A rank without a chain waits `wait_ms` for the chain ranks' evaluation vectors
# (their chain + the scatter: argument 4, default 9 ms) -- on its witness products, which is the point.
wait_ms = float(sys.argv[4]) if len(sys.argv) > 4 else 9.0
# argument 5: where a chain rank starts its chain -- "products_first" (queued behind nothing, beside the products), "chain_first"
# (the chain alone, then the products), "chain_thread" (queued half a millisecond BEFORE the products, from a second thread)
order = sys.argv[5] if len(sys.argv) > 5 else "products_first"
import threading


def run_my_chains():
    for k in chains:
        p.hpoly_chain(k, full.data_ptr())


p.load_witness_part(wtns, 0)
if chains:
    p.load_witness_part(wtns, 1)
walls = []
for it in range(4):
    torch.cuda.synchronize()
    t = time.perf_counter()
    if order == "chain_first":
        run_my_chains()
        t_chain = time.perf_counter()
        p.witness_msm_begin()
        t_begin = time.perf_counter()
    elif order == "chain_thread" and chains:
        th = threading.Thread(target=run_my_chains)
        th.start()
        time.sleep(0.0005)
        p.witness_msm_begin()
        t_begin = time.perf_counter()
        th.join()
        t_chain = time.perf_counter()
    else:
        p.witness_msm_begin()
        t_begin = time.perf_counter()
        run_my_chains()
        t_chain = time.perf_counter()
    if not chains:
        time.sleep(max(0.0, wait_ms * 1e-3 - (time.perf_counter() - t)))
    p.hpoly_combine(bufs[0].data_ptr(), bufs[1].data_ptr(), bufs[2].data_ptr())
    hp = p.run_h_msm()
    t_h = time.perf_counter()
    part = p.witness_msm_end()
    t_end = time.perf_counter()
    p.finish(part[:320] + hp[320:384])
    t_fin = time.perf_counter()
    walls.append([1e3 * (x - t) for x in (t_begin, t_chain, t_h, t_end, t_fin)])
w = walls[-1]
print("order %s, ULTRAGROTH_H_PRIORITY %s" % (order, os.environ.get("ULTRAGROTH_H_PRIORITY", "(default: normal)")))
print("queued form (ms after the start of the step): products queued %.2f  chains done %.2f  H product done %.2f  witness products done %.2f  "
      "finish %.2f   | wall of the last three steps %s" % (w[0], w[1], w[2], w[3], w[4], " ".join("%.2f" % x[4] for x in walls[1:])))
