#!/usr/bin/env python3
"""Soak of the staged witness buffers at a large size: T host threads prove the same 2^L circuit on ONE prover object, N proofs
each, alternating two witnesses, fixed blinding; every proof is compared byte for byte with the expected one
(oracle/closed_form.py).   python tools/soak_pipe.py [L=22] [T=2] [N=12] [b_zero=0]
b_zero: fraction of the signals without a B-side point (the prover's sparse-B form from a quarter on)."""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["ULTRAGROTH_TEST_HOOKS"] = "1"


def main():
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 22
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    N = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    b_zero = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
    import torch  # noqa: F401  (initialises its HIP runtime first, as bench.py does)
    import ultragroth_amd as ug
    from ultragroth_amd import synth
    from oracle import closed_form
    dev = ug.Device(0)
    zkey, _, info = synth.build_circuit(dev, L, mix="U", b_zero=b_zero)
    mask = synth.b_zero_mask(info["nVars"], b_zero) if b_zero else None
    wit = [synth.build_witness(L, "U"), synth.build_witness(L, "C")]
    r, s = bytes(range(1, 32)), bytes(range(31, 62))
    ri, si = int.from_bytes(r, "little"), int.from_bytes(s, "little")
    exp = [closed_form.groth16_expected(zkey, w, synth.SEEDS, synth.g1_generator_record(), synth.g2_generator_record(), ri, si, b_zero_mask=mask) for w in wit]
    assert exp[0] != exp[1]
    bad = []
    with ug.Groth16Prover(zkey) as p:
        ug.set_test_blinding(r + s)

        def caller(k):
            for it in range(N):
                which = (k + it) & 1
                if p.prove(wit[which]) != exp[which]:
                    bad.append((k, it))
        t0 = time.perf_counter()
        th = [threading.Thread(target=caller, args=(k,)) for k in range(T)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        dt = time.perf_counter() - t0
        ug.set_test_blinding(b"")
    print("2^%d, b_zero %.2f, %d threads x %d proofs: %.1f ms per proof, %d mismatches %s" % (L, b_zero, T, N, 1e3 * dt / (T * N), len(bad), bad[:5]))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
