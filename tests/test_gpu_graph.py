"""ULTRAGROTH_GRAPH=1: the device part of a created prover's proof recorded once per witness buffer as a hipGraph and replayed
(csrc/ug_api.hip ug_graph_*, Groth16Prover::run). The replayed proofs are the eager proofs byte for byte; the MSM | FFT split
and the per-kernel statistics (external event records inside the graph) keep counting; a graph whose buffers were re-allocated
is dropped and recorded again; a failure while recording leaves a prover that proves."""
import threading

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


@pytest.mark.parametrize("overlap", ["0", "1", "2"])
def test_replayed_proofs_are_the_eager_proofs(device, monkeypatch, overlap):
    import ultragroth_amd as ug
    from ultragroth_amd import synth
    monkeypatch.setenv("ULTRAGROTH_GRAPH", "1")
    monkeypatch.setenv("ULTRAGROTH_OVERLAP", overlap)
    zkey, wtns, info = synth.build_circuit(device, 15, mix="U", seed=0x5EED0A00)
    wtns2 = synth.build_witness(15, "C", seed=0x5EED0A01)
    r, s = fixed_rs()
    exp = [O.groth16_prove(zkey, w, int.from_bytes(r, "little"), int.from_bytes(s, "little"))[:2] for w in (wtns, wtns2)]
    with ug.Groth16Prover(zkey) as p:
        # eager, recorded + launched, replayed ..., through the reference's call (witness in host memory: re-staged every time
        # into the same buffer) and with the witness changing under the graph
        for k in range(6):
            assert _fixed(ug, r + s, lambda: p.prove((wtns, wtns2)[k % 2])) == exp[k % 2], k
        # ... and on a resident witness
        p.load_witness(wtns2)
        for k in range(4):
            assert _fixed(ug, r + s, p.prove_resident) == exp[1], k
        msm, fft, _ = p.last_timings()
        assert msm > 0 and fft > 0                                   # the spans inside the graph are accounted after every launch
        # per-kernel statistics switched on: the sequences are recorded again WITH their event pairs
        for which in range(4):
            p.kernel_stats(which=which, reset=True)
        for k in range(3):
            assert _fixed(ug, r + s, p.prove_resident) == exp[1]
        ms, launches, units = p.kernel_stats(which=2)                # NTT pass launches: 3 proofs x the same count
        assert launches and launches % 3 == 0 and ms > 0 and units > 0
        grp = p.kernel_stats(which=3)
        assert grp[1] == 3                                           # one group launch (A | B1 | C) per proof


def test_stale_graphs_are_recorded_again_and_failures_leave_a_prover(device, monkeypatch):
    import ultragroth_amd as ug
    from ultragroth_amd import synth
    monkeypatch.setenv("ULTRAGROTH_GRAPH", "1")
    zkey, wtns, info = synth.build_circuit(device, 14, mix="C", seed=0x5EED0A02)
    r, s = fixed_rs()
    exp = O.groth16_prove(zkey, wtns, int.from_bytes(r, "little"), int.from_bytes(s, "little"))[:2]
    with ug.Groth16Prover(zkey) as p, ug.Registry(0) as reg:
        for _ in range(3):
            assert _fixed(ug, r + s, lambda: p.prove(wtns)) == exp
        # another circuit on the device allocates and frees: every recorded sequence of the process is stale afterwards
        zk2, wt2, _ = synth.build_circuit(device, 12, mix="U", seed=0x5EED0A03)
        exp2 = O.groth16_prove(zk2, wt2, int.from_bytes(r, "little"), int.from_bytes(s, "little"))[:2]
        reg.load("other", zk2)
        assert _fixed(ug, r + s, lambda: reg.prove("other", wt2)) == exp2
        for _ in range(3):
            assert _fixed(ug, r + s, lambda: p.prove(wtns)) == exp
        reg.evict("other")
        for _ in range(2):
            assert _fixed(ug, r + s, lambda: p.prove(wtns)) == exp
        # a failure inside the recording (the second proof after the drop records): nothing is queued, the prover proves on
        monkeypatch.setenv("ULTRAGROTH_GRAPH", "0")
        assert _fixed(ug, r + s, lambda: p.prove(wtns)) == exp
        monkeypatch.setenv("ULTRAGROTH_GRAPH", "1")
    with ug.Groth16Prover(zkey) as p:
        assert _fixed(ug, r + s, lambda: p.prove(wtns)) == exp       # eager: sizes the buffers
        ug.inject_fault(ug.FAULT_HPOLY_RUN)                          # the recording proof fails half way
        with pytest.raises(ug.ProverError, match="injected fault"):
            p.prove(wtns)
        ug.inject_fault(ug.FAULT_SCHEDULE_BUILD, after=2)
        with pytest.raises(ug.ProverError, match="injected fault"):
            p.prove(wtns)
        for _ in range(3):
            assert _fixed(ug, r + s, lambda: p.prove(wtns)) == exp


def test_two_host_threads_on_one_prover_with_graphs(device, monkeypatch):
    """two callers alternate between the prover's two witness buffers: one graph per buffer"""
    import ultragroth_amd as ug
    from ultragroth_amd import synth
    monkeypatch.setenv("ULTRAGROTH_GRAPH", "1")
    zkey, wtns, info = synth.build_circuit(device, 16, mix="U", seed=0x5EED0A04)
    r, s = fixed_rs()
    exp = O.groth16_prove(zkey, wtns, int.from_bytes(r, "little"), int.from_bytes(s, "little"))[:2]
    ug.set_test_blinding((r + s) * 1)
    try:
        with ug.Groth16Prover(zkey) as p:
            bad = []

            def work():
                for _ in range(6):
                    try:
                        # (the fixed blinding is process-wide and consumed per draw: both callers draw the same r, s)
                        if p.prove(wtns) != exp:
                            bad.append("mismatch")
                    except BaseException as e:       # noqa: BLE001
                        bad.append(repr(e))
            th = [threading.Thread(target=work) for _ in range(2)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            assert not bad, bad
    finally:
        ug.set_test_blinding(b"")


def test_cold_start_proofs_during_and_after_the_table_build(device, monkeypatch):
    """groth16_prover_create no longer waits for the window tables (ug_prover_tables_ready): proofs that arrive while they are
    built use the classic windows beside the table kernels, the first one that finds them built switches over; every proof is
    the oracle's, the two scalar mixes alternate, and ULTRAGROTH_TABLES_BG=0 keeps the old blocking create"""
    import ultragroth_amd as ug
    from ultragroth_amd import synth
    from oracle import closed_form
    zkey, wtns, info = synth.build_circuit(device, 20, mix="U")
    wtns2 = synth.build_witness(20, "C")
    r, s = fixed_rs()
    ri, si = int.from_bytes(r, "little"), int.from_bytes(s, "little")
    exp = [closed_form.groth16_expected(zkey, w, synth.SEEDS, synth.g1_generator_record(), synth.g2_generator_record(), ri, si) for w in (wtns, wtns2)]
    with ug.Groth16Prover(zkey) as p:
        during = 0
        k = 0
        while True:
            ready = p.tables_ready()
            assert _fixed(ug, r + s, lambda: p.prove((wtns, wtns2)[k % 2])) == exp[k % 2], (k, ready)
            k += 1
            during += 0 if ready else 1
            if ready and k >= 4:
                break
            assert k < 400
        assert p.tables_ready()
        for k in range(4):                                           # ... and on the tables
            assert _fixed(ug, r + s, lambda: p.prove((wtns, wtns2)[k % 2])) == exp[k % 2]
        assert during >= 1                                           # (the build takes ~0.15 s at this size, a classic proof ~15 ms)
    monkeypatch.setenv("ULTRAGROTH_TABLES_BG", "0")
    with ug.Groth16Prover(zkey) as p:
        assert p.tables_ready()                                      # create waited
        assert _fixed(ug, r + s, lambda: p.prove(wtns)) == exp[0]
    # a prover that is destroyed while its tables are still being built
    monkeypatch.delenv("ULTRAGROTH_TABLES_BG")
    ug.Groth16Prover(zkey).close()
    with ug.Groth16Prover(zkey) as p:
        p.tables_ready(wait=True)
        assert _fixed(ug, r + s, lambda: p.prove(wtns2)) == exp[1]


@pytest.mark.parametrize("b_zero", [0.5, 0.97, 0.3])
def test_sparse_b_circuits(device, monkeypatch, b_zero):
    """Real circuits leave many signals off the B side of every constraint: B1 and B2 hold points at infinity for them. From a quarter
    of such points on the prover keeps B1 / B2 compacted over the signals that have a real point, with a schedule of its own over the
    gathered scalars, and [A | C] as the G1 group (DeviceProver: sparse B); below that, and with ULTRAGROTH_SPARSE_B=0, the dense
    form. Every form proves the oracle's proof: eager and recorded, on the classic windows and on the tables, two witnesses."""
    import ultragroth_amd as ug
    from ultragroth_amd import synth
    zkey, wtns, info = synth.build_circuit(device, 15, mix="U", seed=0x5EED0B00, b_zero=b_zero)
    wtns2 = synth.build_witness(15, "C", seed=0x5EED0B01)
    r, s = fixed_rs()
    exp = [O.groth16_prove(zkey, w, int.from_bytes(r, "little"), int.from_bytes(s, "little"))[:2] for w in (wtns, wtns2)]
    from oracle import closed_form
    mask = synth.b_zero_mask(info["nVars"], b_zero, 0x5EED0B00)
    assert closed_form.groth16_expected(zkey, wtns, synth.SEEDS, synth.g1_generator_record(), synth.g2_generator_record(),
                                        int.from_bytes(r, "little"), int.from_bytes(s, "little"), b_zero_mask=mask) == exp[0]
    for sparse, graph in (("1", "0"), ("1", "1"), ("0", "0")):
        monkeypatch.setenv("ULTRAGROTH_SPARSE_B", sparse)
        monkeypatch.setenv("ULTRAGROTH_GRAPH", graph)
        with ug.Groth16Prover(zkey) as p:
            for k in range(4):                                       # (classic windows first, the tables arrive meanwhile)
                assert _fixed(ug, r + s, lambda: p.prove((wtns, wtns2)[k % 2])) == exp[k % 2], (sparse, graph, k)
            p.tables_ready(wait=True)
            for k in range(4):
                assert _fixed(ug, r + s, lambda: p.prove((wtns, wtns2)[k % 2])) == exp[k % 2], (sparse, graph, k)
    # the one-shot call (no tables) and the phase calls of an unsharded prover take the same path
    monkeypatch.setenv("ULTRAGROTH_SPARSE_B", "1")
    monkeypatch.setenv("ULTRAGROTH_GRAPH", "0")
    ug.set_test_blinding(r + s)
    try:
        assert ug.groth16_prover(zkey, wtns) == exp[0]
    finally:
        ug.set_test_blinding(b"")
    # ... and the resident multi-circuit prover, which builds (and may drop) the tables of all three groups itself
    with ug.Registry(0) as reg:
        reg.load("sparse", zkey)
        for k in range(3):
            assert _fixed(ug, r + s, lambda: reg.prove("sparse", (wtns, wtns2)[k % 2])) == exp[k % 2]


@pytest.mark.parametrize("b_zero", [0.5, 0.9])
def test_sparse_b_ultragroth(device, monkeypatch, b_zero):
    """the same for the UltraGroth prover (src/ultra_groth.cpp:201,214,227: A, B1, B2 over the completed witness): A alone over the
    witness schedule, B1 / B2 compacted with a schedule of their own; the oracle's proof with and without the sparse form"""
    import ultragroth_amd as ug
    from ultragroth_amd import synth
    zkey, uwtns, info = synth.build_ultra_circuit(device, 15, b_zero=b_zero)
    rk, r, s = bytes(range(1, 32)), bytes(range(40, 71)), bytes(range(80, 111))
    exp = O.ultra_groth_prove(zkey, uwtns, *(int.from_bytes(b, "little") for b in (rk, r, s)))
    for sparse in ("1", "0"):
        monkeypatch.setenv("ULTRAGROTH_SPARSE_B", sparse)
        with ug.UltraGrothProver(zkey) as p:
            for k in range(3):
                assert _fixed(ug, rk + r + s, lambda: p.prove(uwtns)) == exp, (sparse, k)


def test_sparse_b_on_the_ranks_of_a_many_device_prover(device, monkeypatch):
    """every rank of a sharded prover keeps ITS range of B1 / B2 compacted when enough of its B points are at infinity (signal
    numbers of the whole witness, the rank's own slice of the scalars): the reference API over four ranks on one device, and the
    ranks made from their slices only (what bench.py --gpus N does), against the oracle"""
    import os
    import ultragroth_amd as ug
    from ultragroth_amd import synth
    log_domain = 17
    zkey, wtns, info = synth.build_circuit(device, log_domain, mix="C", seed=0x5EED0C00, b_zero=0.6)
    r, s = fixed_rs()
    exp = O.groth16_prove(zkey, wtns, int.from_bytes(r, "little"), int.from_bytes(s, "little"))[:2]
    for sparse in ("1", "0"):
        monkeypatch.setenv("ULTRAGROTH_SPARSE_B", sparse)
        monkeypatch.setenv("ULTRAGROTH_DEVICES", "0,0,0,0")
        with ug.Groth16Prover(zkey) as p:
            for k in range(2):
                assert _fixed(ug, r + s, lambda: p.prove(wtns)) == exp, (sparse, k)
        monkeypatch.delenv("ULTRAGROTH_DEVICES")
    monkeypatch.setenv("ULTRAGROTH_SPARSE_B", "1")
    world, n_dom, nv = 4, info["domainSize"], info["nVars"]
    ranks = []
    for k in range(world):
        lay = ug.ShardedGroth16Prover.shard_layout(nv, 1, n_dom, k, world, world)
        header, coefs, slices = synth.build_circuit_slices(device, log_domain, lay.ranges, with_coefs=bool(lay.chains), seed=0x5EED0C00, b_zero=0.6)
        ranks.append(ug.ShardedGroth16Prover.from_slices(header, coefs, info["nCoefs"], slices, 0, k, world, public_size=86, layout=lay))
    total = None
    for p in ranks:
        p.load_witness_part(wtns, 0)
        part = p.run_witness_msm()
        total = part if total is None else ug.ShardedGroth16Prover.add_partials(total, part)
    # (A, B1, B2 and C of the whole proof; H is left out here: compare with the oracle's sums through a one-device prover's phases)
    whole = ug.ShardedGroth16Prover(zkey, 0, 0, 1)
    try:
        whole.load_witness(wtns)
        assert whole.run_witness_msm()[:320] == total[:320]
    finally:
        whole.close()
        for p in ranks:
            p.close()
