#!/usr/bin/env python3
"""Round-4 experiment (second half): do the latency-bound ends of a many-GPU rank's witness products overlap
when the G1 group and the G2 product run on TWO streams? One rank of eight at 2^24 (2.49 M points, window tables): the group product
and the B2 product over one schedule, queued (a) on one context one after the other, as the prover does, (b) on two contexts of
the device at once (the second ordered behind the schedule only). Sums compared.   python tools/exp_two_streams.py [points]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                  # noqa: E402,F401
import ultragroth_amd as ug                   # noqa: E402
from ultragroth_amd import synth              # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2487222
d1, d2 = ug.Device(0), ug.Device(0)
L = d1._L
c = d1.table_window(n)
A = synth.synth_points(d1, n, synth.SEEDS["A"]); B1 = synth.synth_points(d1, n, synth.SEEDS["B1"]); Cc = synth.synth_points(d1, n, synth.SEEDS["C"])
B2 = synth.synth_points(d1, n, synth.SEEDS["B2"], g2=True)
grp = d1.bases_group([(A, n, 0), (B1, n, 0), (Cc, n, 0)], 0, n, table_c=c)
g2 = d1.bases(B2, n, g2=True, table_c=c)
w = synth.scalars(n, "U", 12345)
v = d1.dvec(n, w.tobytes())
print("points %d, table window %d" % (n, c))


def run(two):
    sch = d1.schedule(v, 0, n, table_c=c)              # (queued on d1's stream, no host wait)
    outs = [C.create_string_buffer(64) for _ in range(3)]
    arr = (C.c_void_p * 3)(*[C.cast(o, C.c_void_p) for o in outs])
    o2 = C.create_string_buffer(128)
    a2 = (C.c_void_p * 1)(C.cast(o2, C.c_void_p))
    b2 = (C.c_void_p * 1)(g2.h)
    ctx2 = d2 if two else d1
    if two:
        assert L.ug_ctx_wait(d2._h, d1._h) == 0          # the second stream starts behind the schedule
    assert L.ug_msm_group_enqueue(d1._h, grp.h, sch.h, arr) == 0
    assert L.ug_msm_batch_enqueue(ctx2._h, 1, b2, sch.h, None, a2) == 0
    assert L.ug_ctx_collect(d1._h) == 0
    if two:
        assert L.ug_ctx_collect(d2._h) == 0
    return [o.raw for o in outs] + [o2.raw]


ref = run(False)
for two in (False, True, False, True):
    ts = []
    for _ in range(5):
        torch.cuda.synchronize()
        t = time.perf_counter()
        got = run(two)
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t))
        assert got == ref
    print("%s: schedule + group + G2 products, ms: %s" % ("two streams" if two else "one stream ", " ".join("%.2f" % x for x in ts)))
