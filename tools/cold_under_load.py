#!/usr/bin/env python3
"""Cold start under load: groth16_prover_create, then proofs back to back (T host threads, .wtns in host memory) from the first moment;
how many proofs and how long until the window tables are in use, and what a proof costs meanwhile.
    python tools/cold_under_load.py [L=24] [T=1]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                  # noqa: E402,F401
import ultragroth_amd as ug                   # noqa: E402
from ultragroth_amd import synth              # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 24
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = ug.Device(0)
zkey, wtns, info = synth.build_circuit(dev, L, mix="U")
t0 = time.perf_counter()
p = ug.Groth16Prover(zkey)
create = time.perf_counter() - t0
lock = threading.Lock()
log = []


def work():
    while True:
        ready = p.tables_ready()
        t = time.perf_counter()
        p.prove(wtns)
        with lock:
            log.append((time.perf_counter() - t0, 1e3 * (time.perf_counter() - t), ready))
            done = sum(1 for x in log if x[2]) >= 4 * T
        if done:
            return


th = [threading.Thread(target=work) for _ in range(T)]
for x in th:
    x.start()
for x in th:
    x.join()
log.sort()
before = [x for x in log if not x[2]]
after = [x for x in log if x[2]]
print("2^%d, %d host thread(s): create %.2f s; %d proofs before the tables were in use (%.1f ms per call, first %.1f), tables in use %.2f s after "
      "the start of create; then %.1f ms per call" % (L, T, create, len(before), sum(x[1] for x in before) / max(len(before), 1), before[0][1] if before else 0,
                                                      after[0][0] if after else -1, sum(x[1] for x in after[1:]) / max(len(after) - 1, 1)))
p.close()
