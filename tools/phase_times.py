#!/usr/bin/env python3
"""Per-phase times of ONE rank of a sharded Groth16 prover (rank r of W) on one GPU: the critical path of a W-GPU run
without the collectives. Usage: python tools/phase_times.py [log_domain] [world] [rank]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ultragroth_amd as ug
from ultragroth_amd import synth

log_domain = int(sys.argv[1]) if len(sys.argv) > 1 else 24
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rank = int(sys.argv[3]) if len(sys.argv) > 3 else 0
torch.cuda.set_device(0)
dev = ug.Device(0)
zkey, wtns, info = synth.build_circuit(dev, log_domain, mix="U")
t0 = time.perf_counter()
p = ug.ShardedGroth16Prover(zkey, 0, rank, world)
print("create %.2f s" % (time.perf_counter() - t0))
del zkey
p.load_witness(wtns)
n = info["domainSize"]
sl = n // world
full = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
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
    part = timed("witness_msm", p.run_witness_msm, acc)
    for k in range(3):
        timed("chain_%d" % k, lambda: p.hpoly_chain(k, full.data_ptr()), acc)
    timed("combine", lambda: p.hpoly_combine(bufs[0].data_ptr(), bufs[1].data_ptr(), bufs[2].data_ptr()), acc)
    hp = timed("h_msm", p.run_h_msm, acc)
    total = part[:320] + hp[320:384]
    timed("finish", lambda: p.finish(total), acc)
print("rank %d of %d at 2^%d:" % (rank, world, log_domain), "  ".join("%s %.2f" % kv for kv in acc.items()))
