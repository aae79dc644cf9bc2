#!/usr/bin/env python3
"""Per-phase times of ONE rank of a sharded UltraGroth prover (rank R of W, created from its slices as bench.py --ultra does) on one
GPU: the critical path of a W-GPU run of BASELINE.json configs[4] without the collectives.
    python tools/phase_times_ultra.py [log_domain=22] [world=8] [rank=5] [wait_ms=3]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["ULTRAGROTH_TEST_HOOKS"] = "1"
import torch                                  # noqa: E402
import ultragroth_amd as ug                   # noqa: E402
from ultragroth_amd import synth              # noqa: E402

log_domain = int(sys.argv[1]) if len(sys.argv) > 1 else 22
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rank = int(sys.argv[3]) if len(sys.argv) > 3 else 5
wait_ms = float(sys.argv[4]) if len(sys.argv) > 4 else 3.0
torch.cuda.set_device(0)
dev = ug.Device(0)
info = synth.ultra_info(log_domain)
n_dom = info["domainSize"]
rg = ug.ShardedUltraGrothProver.shard_ranges(info["nVars"], n_dom, info["nC1"], info["nC2"], rank, world)
chains = [k for k in range(3) if k % world == rank]
header, coefs, slices = synth.build_ultra_circuit_slices(dev, log_domain, rg, with_coefs=bool(chains))
uwtns = synth.build_ultra_witness(log_domain, "C", lookup_log=16)
t0 = time.perf_counter()
p = ug.ShardedUltraGrothProver.from_slices(header, coefs, info["nCoefs"], slices, 0, rank, world, public_size=86)
print("ultragroth rank %d of %d at 2^%d: witness %s, round set %s, final set %s, h %s, chains %s, create %.2f s"
      % (rank, world, log_domain, rg[0], rg[1], rg[2], rg[3], chains, time.perf_counter() - t0))
del coefs, slices
first, cnt, _ = p.h_range()
full = torch.empty((n_dom, 32), dtype=torch.uint8, device="cuda")
bufs = torch.randint(0, 256, (3, max(cnt, 1), 32), dtype=torch.uint8, device="cuda")
bufs[:, :, 31] &= 0x0f
ug.set_test_blinding(bytes(range(1, 32)) * 200)        # (every draw of the run comes from here: the commitment must be a curve point)
walls = []
for it in range(4):
    torch.cuda.synchronize()
    t = time.perf_counter()
    p.load_witness(uwtns)
    t_load = time.perf_counter()
    part = p.round_commit()
    t_commit = time.perf_counter()
    commitment = p.round_finish(part)                   # (this rank's part alone stands for the sum: a valid point)
    p.apply_commitment(commitment)
    t_apply = time.perf_counter()
    if chains:                                          # eight ranks: chain first, alone
        for k in chains:
            p.hpoly_chain(k, full.data_ptr())
    t_chain = time.perf_counter()
    p.witness_msm_begin()
    if not chains:
        time.sleep(wait_ms * 1e-3)                      # the chain ranks' vectors arrive
    p.hpoly_combine(bufs[0].data_ptr(), bufs[1].data_ptr(), bufs[2].data_ptr())
    hp = p.run_h_msm()
    t_h = time.perf_counter()
    blk = p.witness_msm_end()
    t_end = time.perf_counter()
    walls.append([1e3 * (x - t) for x in (t_load, t_commit, t_apply, t_chain, t_h, t_end)])
w = walls[-1]
print("ms after the start of the step: witness loaded %.2f  round commitment %.2f  challenge + lookup applied %.2f  chains done %.2f  "
      "H product done %.2f  final-round products done %.2f   | step wall of the last three: %s"
      % (w[0], w[1], w[2], w[3], w[4], w[5], " ".join("%.2f" % x[5] for x in walls[1:])))
