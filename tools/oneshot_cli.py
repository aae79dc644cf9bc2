#!/usr/bin/env python3
"""Wall time of the reference's one-shot entry points at a given size (default 2^24): `prover <zkey> <wtns> <proof.json>
<public.json>` (src/main_prover.cpp: read the files, create, prove once, write) and groth16_prover() on buffers. A one-shot
call never builds window tables (they cannot pay for one proof), so this is the classic-window path end to end.
    python tools/oneshot_cli.py [log_domain] > gpurun_out/oneshot.json"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ultragroth_amd as ug
from ultragroth_amd import synth

log_domain = int(sys.argv[1]) if len(sys.argv) > 1 else 24
dev = ug.Device(0)
zkey, wtns, info = synth.build_circuit(dev, log_domain, mix="U")
tmp = os.environ.get("TMPDIR", "/tmp")
zp, wp = os.path.join(tmp, "oneshot.zkey"), os.path.join(tmp, "oneshot.wtns")
with open(zp, "wb") as f:
    f.write(memoryview(zkey))
with open(wp, "wb") as f:
    f.write(wtns)
res = {"log_domain": log_domain, "zkey_bytes": len(zkey)}
t0 = time.perf_counter()
ug.groth16_prover(zkey, wtns)
res["groth16_prover_buffers_s"] = time.perf_counter() - t0
t0 = time.perf_counter()
ug.groth16_prover(zkey, wtns)
res["groth16_prover_buffers_second_call_s"] = time.perf_counter() - t0
del zkey
dev.close()
exe = os.path.join(ROOT, "ultragroth_amd", "csrc", "prover")
for label, env in (("cli_s", {}), ("cli_second_run_s", {}), ("cli_two_ranks_one_device_s", {"ULTRAGROTH_DEVICES": "0,0"})):
    t0 = time.perf_counter()
    r = subprocess.run([exe, zp, wp, os.path.join(tmp, "p.json"), os.path.join(tmp, "pub.json")], capture_output=True, text=True,
                       env=dict(os.environ, **env))
    res[label] = time.perf_counter() - t0
    if r.returncode != 0:
        res[label + "_error"] = r.stderr[-300:]
os.remove(zp); os.remove(wp)
print(json.dumps(res))
