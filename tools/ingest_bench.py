#!/usr/bin/env python3
"""zkey ingest: time groth16_prover_create from a buffer and groth16_prover_create_zkey_file
from an mmap'ed file, with and without the fixed-base window tables, and report the host-to-HBM rate of the zkey.

    python tools/ingest_bench.py [log_domain=24] [dir=/dev/shm]
"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ultragroth_amd as ug          # noqa: E402
from ultragroth_amd import synth      # noqa: E402


def main():
    log = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    where = sys.argv[2] if len(sys.argv) > 2 else "/dev/shm"
    dev = ug.Device(0)
    zkey, wtns, info = synth.build_circuit(dev, log, mix="U")
    path = os.path.join(where, "ug_ingest_%d.zkey" % log)
    with open(path, "wb") as f:
        f.write(zkey)
    L = ug.load()
    res = {"log_domain": log, "zkey_bytes": len(zkey), "file": path}
    try:
        for tables in ("0", "1"):
            os.environ["ULTRAGROTH_TABLES"] = tables
            for kind in ("buffer", "file"):
                h = C.c_void_p()
                err = C.create_string_buffer(256)
                t0 = time.perf_counter()
                if kind == "buffer":
                    rc = L.groth16_prover_create(C.byref(h), zkey, len(zkey), err, 255)
                else:
                    rc = L.groth16_prover_create_zkey_file(C.byref(h), path.encode(), err, 255)
                dt = time.perf_counter() - t0
                assert rc == 0, err.value
                L.groth16_prover_destroy(h)
                res["create_s_%s_tables%s" % (kind, tables)] = round(dt, 3)
                if tables == "0":
                    res["ingest_gbs_%s" % kind] = round(len(zkey) / dt / 1e9, 2)
    finally:
        os.remove(path)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
