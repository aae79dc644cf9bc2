"""Whole-proof bit-exactness at BASELINE.json's full sizes (configs[2] = 2^24, configs[3] = 2^26 constraints).

A CPU Pippenger at these sizes takes minutes to hours, so the expected proof is assembled without one
(oracle/closed_form.py): the H polynomial by the oracle at full size, the five MSMs in the exponent (the synthetic
base points are the generator walk P_i = (seed + i) G, so sum s_i P_i = (sum s_i (seed + i) mod r) G), blinding and
JSON by the oracle's restatement of src/groth16.cpp:158-250. The GPU side is the product path end to end: a CREATED
prover (window tables at 2^24; classic windows at 2^26, where the tables do not fit one GPU), the witness handed over in
host memory, groth16_prover_prove. Everything the bench line times is therefore compared byte for byte: CSR sort and
mat-vec with 2^26 / 2^28 coefficient records, the NTT chains with the fused twist, h = a.b - c, the batched
A/B1/B2/C products (G2 at the full size) and the H product.

Sizes: UG_FULL_LOG (default 24) and UG_HUGE_LOG (default 26; 0 skips that test). Progress goes to
gpurun_out/fullsize_progress.log (pytest captures stdout; a long silent test would look hung).
"""
import os
import time

import pytest

import oracle as O
from oracle import closed_form
from conftest import fixed_rs, ROOT

pytestmark = pytest.mark.gpu

FULL_LOG = int(os.environ.get("UG_FULL_LOG", "24"))
HUGE_LOG = int(os.environ.get("UG_HUGE_LOG", "26"))


def _progress(msg):
    d = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "fullsize_progress.log"), "a") as f:
            f.write("%s %s\n" % (time.strftime("%H:%M:%S"), msg))
    except OSError:
        pass


def _expected(zkey, wtns, log_domain):
    from ultragroth_amd import synth
    r, s = fixed_rs()
    return closed_form.groth16_expected(zkey, wtns, synth.SEEDS, synth.g1_generator_record(), synth.g2_generator_record(),
                                        int.from_bytes(r, "little"), int.from_bytes(s, "little"),
                                        progress=lambda m: _progress("2^%d: %s" % (log_domain, m)))


def _prove(prover, wtns):
    import ultragroth_amd as ug
    r, s = fixed_rs()
    ug.set_test_blinding(r + s)
    try:
        return prover.prove(wtns)
    finally:
        ug.set_test_blinding(b"")


@pytest.fixture(scope="module")
def full_zkey(device):
    """the 2^FULL_LOG circuit of the bench (the zkey does not depend on the scalar mix)"""
    from ultragroth_amd import synth
    _progress("2^%d: building the circuit" % FULL_LOG)
    zkey, _, info = synth.build_circuit(device, FULL_LOG, mix="C")
    return zkey, info


@pytest.fixture(scope="module")
def full_prover(full_zkey):
    import ultragroth_amd as ug
    _progress("2^%d: creating the prover (window tables)" % FULL_LOG)
    p = ug.Groth16Prover(full_zkey[0])
    yield p
    p.close()


def test_closed_form_equals_the_oracle_prover(device):
    """the closed-form assembly itself, against the oracle's real Pippenger where that is cheap (2^13)"""
    from ultragroth_amd import synth
    r, s = fixed_rs()
    ri, si = int.from_bytes(r, "little"), int.from_bytes(s, "little")
    for mix, g1_only in (("U", False), ("C", False), ("U", True)):
        zkey, wtns, info = synth.build_circuit(device, 13, mix=mix, g1_only=g1_only)
        exp = O.groth16_prove(zkey, wtns, ri, si)
        got = closed_form.groth16_expected(zkey, wtns, synth.SEEDS, synth.g1_generator_record(), synth.g2_generator_record(),
                                           ri, si, g1_only=g1_only)
        assert got == (exp[0], exp[1])


def test_whole_proof_at_configs2_size_uniform(full_zkey, full_prover):
    """BASELINE.json configs[2]: 2^24 constraints, uniform scalars, created prover with window tables -- the bench's
    default workload (a second witness on the same prover object follows in the circom-like test below; two different
    witnesses through one prover are also compared at 2^20, test_created_prover_at_2_20_bit_exact)"""
    from ultragroth_amd import synth
    zkey, info = full_zkey
    wtns = synth.build_witness(FULL_LOG, "U")
    _progress("2^%d U: proving" % FULL_LOG)
    assert _prove(full_prover, wtns) == _expected(zkey, wtns, FULL_LOG)


def test_whole_proof_at_configs2_size_circom_like(full_zkey, full_prover):
    """the SAME prover object with another witness, the circom-like one (40 % zeros and ones: million-entry buckets, the
    heavy-bucket path with window tables): nothing may leak from the previous proof"""
    from ultragroth_amd import synth
    zkey, info = full_zkey
    wtns = synth.build_witness(FULL_LOG, "C")
    _progress("2^%d C: proving" % FULL_LOG)
    assert _prove(full_prover, wtns) == _expected(zkey, wtns, FULL_LOG)


def test_piecewise_ranges_at_full_size(full_zkey, full_prover, monkeypatch):
    """ranges above ULTRAGROTH_MAX_RANGE scalars are proved in pieces whose partial sums are added (the path the
    reference's largest legal domain, 2^27, takes): forced here at the full size with 2^22-scalar pieces, which also
    run the classic windows (no tables) at this size"""
    import ultragroth_amd as ug
    from ultragroth_amd import synth
    zkey, info = full_zkey
    full_prover.close()                                         # one 2^24 prover with tables at a time is enough
    monkeypatch.setenv("ULTRAGROTH_MAX_RANGE", str(1 << (FULL_LOG - 2)))
    wtns = synth.build_witness(FULL_LOG, "C")
    _progress("2^%d C piecewise: creating + proving" % FULL_LOG)
    with ug.Groth16Prover(zkey) as p:
        assert _prove(p, wtns) == _expected(zkey, wtns, FULL_LOG)


@pytest.mark.skipif(HUGE_LOG == 0, reason="UG_HUGE_LOG=0")
def test_whole_proof_at_configs3_size(device):
    """BASELINE.json configs[3]'s circuit, 2^26 constraints, on ONE GPU (classic windows; 2^28 coefficient records)"""
    import ultragroth_amd as ug
    from ultragroth_amd import synth
    free, total = device.mem_info()
    if total < (200 << 30):
        pytest.skip("needs a 288 GB device")
    _progress("2^%d: building the circuit" % HUGE_LOG)
    zkey, wtns, info = synth.build_circuit(device, HUGE_LOG, mix="U")
    _progress("2^%d: creating the prover" % HUGE_LOG)
    with ug.Groth16Prover(zkey) as p:
        _progress("2^%d: proving" % HUGE_LOG)
        got = _prove(p, wtns)
    assert got == _expected(zkey, wtns, HUGE_LOG)
    _progress("2^%d: done" % HUGE_LOG)
