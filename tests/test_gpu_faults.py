"""Error paths of a running proof (round-2 advice): a call that fails after it has queued work on the device must leave
nothing behind -- no queued MSM whose result pointer went away with the caller's frame, no leaked timing events, the
witness lease and the registry's accounting released -- so the SAME prover object proves bit-exact afterwards.
Failures are injected with the test hook ug_test_inject_fault (honoured only under ULTRAGROTH_TEST_HOOKS=1)."""
import pytest

import oracle as O
from conftest import fixed_rs

pytestmark = pytest.mark.gpu


def _fixed(ug, blob, call):
    ug.set_test_blinding(blob)
    try:
        return call()
    finally:
        ug.set_test_blinding(b"")


def test_groth16_prover_survives_mid_run_failures(device):
    import ultragroth_amd as ug
    from ultragroth_amd import synth
    zkey, wtns, info = synth.build_circuit(device, 14, mix="U", seed=0x5EED0900)
    r, s = fixed_rs()
    exp = O.groth16_prove(zkey, wtns, int.from_bytes(r, "little"), int.from_bytes(s, "little"))[:2]
    with ug.Groth16Prover(zkey) as p:
        assert _fixed(ug, r + s, lambda: p.prove(wtns)) == exp
        # the H polynomial fails with the four witness products already queued (their results point into run()'s frame)
        ug.inject_fault(ug.FAULT_HPOLY_RUN)
        with pytest.raises(ug.ProverError, match="injected fault"):
            p.prove(wtns)
        assert _fixed(ug, r + s, lambda: p.prove(wtns)) == exp
        # the h schedule (second schedule of a proof) fails: witness products and the whole NTT block are in flight
        ug.inject_fault(ug.FAULT_SCHEDULE_BUILD, after=2)
        with pytest.raises(ug.ProverError, match="injected fault"):
            p.prove(wtns)
        # ... and the very first call of the device part
        ug.inject_fault(ug.FAULT_SCHEDULE_BUILD)
        with pytest.raises(ug.ProverError, match="injected fault"):
            p.prove(wtns)
        for _ in range(3):
            assert _fixed(ug, r + s, lambda: p.prove(wtns)) == exp


def test_ultragroth_prover_survives_mid_run_failures(device):
    import ultragroth_amd as ug
    from ultragroth_amd import synth
    zkey, uwtns, info = synth.build_ultra_circuit(device, 12)
    rk, r, s = bytes(range(1, 32)), bytes(range(40, 71)), bytes(range(80, 111))
    exp = O.ultra_groth_prove(zkey, uwtns, *(int.from_bytes(b, "little") for b in (rk, r, s)))
    with ug.UltraGrothProver(zkey) as p:
        assert _fixed(ug, rk + r + s, lambda: p.prove(uwtns)) == exp
        for site, after in ((ug.FAULT_HPOLY_RUN, 1), (ug.FAULT_SCHEDULE_BUILD, 3), (ug.FAULT_SCHEDULE_BUILD, 4), (ug.FAULT_SCHEDULE_BUILD, 1)):
            ug.inject_fault(site, after)
            with pytest.raises(ug.ProverError, match="injected fault"):
                p.prove(uwtns)
            assert _fixed(ug, rk + r + s, lambda: p.prove(uwtns)) == exp


def test_registry_accounting_after_a_failed_proof(device):
    import ultragroth_amd as ug
    from ultragroth_amd import synth
    zkey, wtns, info = synth.build_circuit(device, 13, mix="C", seed=0x5EED0901)
    r, s = fixed_rs()
    exp = O.groth16_prove(zkey, wtns, int.from_bytes(r, "little"), int.from_bytes(s, "little"))[:2]
    with ug.Registry(0) as reg:
        reg.load("c13", zkey)
        assert _fixed(ug, r + s, lambda: reg.prove("c13", wtns)) == exp
        ug.inject_fault(ug.FAULT_HPOLY_RUN)
        with pytest.raises(ug.ProverError, match="injected fault"):
            reg.prove("c13", wtns)
        assert reg.info("c13")[2] == 2                               # the failed call's bracket was closed as well
        assert _fixed(ug, r + s, lambda: reg.prove("c13", wtns)) == exp
        # a bad replacement leaves the resident circuit as it was; a stale file path does not survive load + evict
        with pytest.raises(ug.ProverError):
            reg.load("c13", bytes(zkey)[:4096])
        assert reg.info("c13")[1] in (ug.Registry.RESIDENT, ug.Registry.RESIDENT_WITH_TABLES)
        assert _fixed(ug, r + s, lambda: reg.prove("c13", wtns)) == exp
        reg.evict("c13")
        assert reg.info("c13")[1] == ug.Registry.NOT_LOADED
        with pytest.raises(ug.ProverError, match="circuit not loaded"):
            reg.prove("c13", wtns)


def test_short_slice_buffers_are_refused(device):
    import ultragroth_amd as ug
    from ultragroth_amd import synth
    log_domain, world, k = 12, 2, 1
    nv, n_dom = (1 << log_domain) - 1, 1 << log_domain
    rg = ug.ShardedGroth16Prover.shard_ranges(nv, 1, n_dom, k, world)
    header, coefs, slices = synth.build_circuit_slices(device, log_domain, rg, with_coefs=False)
    for bad in range(5):
        cut = list(slices)
        cut[bad] = bytes(cut[bad])[:-64]
        with pytest.raises(ug.ProverError, match="slice is shorter than this rank's range"):
            ug.ShardedGroth16Prover.from_slices(header, None, 0, tuple(bytes(x) for x in cut), 0, k, world, public_size=86)
    p = ug.ShardedGroth16Prover.from_slices(header, None, 0, slices, 0, k, world, public_size=86)
    p.close()


def test_queued_witness_products_survive_failures(device):
    """the two-call form (ug_groth16_prover_witness_msm_begin / _end): a schedule that fails inside _begin leaves nothing queued;
    an H branch that fails BETWEEN the two calls leaves the queued products collectable; the same objects prove bit-exact
    afterwards. And the many-device prover (ULTRAGROTH_DEVICES, one rank per listed device, the products queued on every rank):
    a failing chain on one rank reaches every waiting rank, the queued products are dropped, the next proof is bit-exact."""
    import os
    import torch
    import ultragroth_amd as ug
    from ultragroth_amd import synth
    zkey, wtns, info = synth.build_circuit(device, 13, mix="U", seed=0x5EED0902)
    r, s = fixed_rs()
    exp = O.groth16_prove(zkey, wtns, int.from_bytes(r, "little"), int.from_bytes(s, "little"))[:2]
    n_dom = info["domainSize"]
    p = ug.ShardedGroth16Prover(zkey, 0, 0, 1)
    try:
        p.load_witness(wtns)
        full = torch.empty((3, n_dom, 32), dtype=torch.uint8, device="cuda")

        def whole():
            ug.set_test_blinding(r + s)
            try:
                p.witness_msm_begin()
            finally:
                ug.set_test_blinding(b"")
            for k in range(3):
                p.hpoly_chain(k, full[k].data_ptr())
            p.hpoly_combine(full[0].data_ptr(), full[1].data_ptr(), full[2].data_ptr())
            hpart = p.run_h_msm()
            return p.finish(p.witness_msm_end()[:320] + hpart[320:384])
        assert whole() == exp
        ug.inject_fault(ug.FAULT_SCHEDULE_BUILD)                 # inside _begin: the witness schedule
        with pytest.raises(ug.ProverError, match="injected fault"):
            p.witness_msm_begin()
        with pytest.raises(ug.ProverError, match="no witness products queued"):
            p.witness_msm_end()
        assert whole() == exp
        p.witness_msm_begin()
        ug.inject_fault(ug.FAULT_SCHEDULE_BUILD)                 # the H schedule fails while the witness products are queued
        p.hpoly_chain(0, full[0].data_ptr())
        with pytest.raises(ug.ProverError, match="injected fault"):
            p.run_h_msm()
        assert len(p.witness_msm_end()) == 384                   # ... they are still there to be collected
        assert whole() == exp
    finally:
        p.close()
    os.environ["ULTRAGROTH_DEVICES"] = "0,0,0,0"
    try:
        with ug.Groth16Prover(zkey) as mp:
            assert _fixed(ug, r + s, lambda: mp.prove(wtns)) == exp
            ug.inject_fault(ug.FAULT_SCHEDULE_BUILD, after=2)    # some rank's schedule: that rank gives up, the others finish
            with pytest.raises(ug.ProverError, match="injected fault"):
                mp.prove(wtns)
            for _ in range(2):
                assert _fixed(ug, r + s, lambda: mp.prove(wtns)) == exp
    finally:
        del os.environ["ULTRAGROTH_DEVICES"]
