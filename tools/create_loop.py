import sys, time, os, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import ultragroth_amd as ug
from ultragroth_amd import synth
dev = ug.Device(0)
zkey, wtns, info = synth.build_circuit(dev, 24, mix="U")
for i in range(4):
    t0 = time.perf_counter(); p = ug.Groth16Prover(zkey); t1 = time.perf_counter()
    p.prove(wtns); t2 = time.perf_counter()
    p.close(); t3 = time.perf_counter()
    print("iter %d: create %.2f s  first prove %.3f s  destroy %.2f s" % (i, t1 - t0, t2 - t1, t3 - t2), flush=True)
