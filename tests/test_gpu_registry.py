"""SURVEY.md section 8(f) rows 2 and 4 on the GPU:

* ug_registry_* -- the resident multi-circuit prover that replaces FullProver's map<circuit, Prover>
  (src/fullprover.cpp:21-63): several circuits on one device under an HBM budget, interleaved proofs bit-exact, window
  tables / workspaces / whole circuits given back least-recently-used first.
* zkey ingest from a file (`*_create_zkey_file`, src/prover.cpp:449-473 with src/fileloader.cpp:23-51): the mmap is
  streamed through pinned staging buffers with the point conversion behind each chunk, and -- unlike the reference,
  whose prover keeps pointers into a mapping that is gone when create returns -- the created
  prover owns everything it needs.
"""
import json
import os

import pytest

import oracle as O
from conftest import fixed_rs, GOLDEN

pytestmark = pytest.mark.gpu


def _prove_fixed(fn, *args):
    import ultragroth_amd as ug
    r, s = fixed_rs()
    ug.set_test_blinding(r + s)
    try:
        return fn(*args)
    finally:
        ug.set_test_blinding(b"")


def _expected(zkey, wtns):
    r, s = fixed_rs()
    e = O.groth16_prove(zkey, wtns, int.from_bytes(r, "little"), int.from_bytes(s, "little"))
    return e[0], e[1]


@pytest.fixture(scope="module")
def circuits(device):
    from ultragroth_amd import synth
    out = {}
    for name, log, mix, seed in (("small", 14, "U", 0x5EED0A00), ("mid", 15, "C", 0x5EED0B00), ("large", 16, "U", 0x5EED0C00)):
        zkey, wtns, _ = synth.build_circuit(device, log, mix=mix, seed=seed)
        out[name] = (bytes(zkey), wtns, _expected(bytes(zkey), wtns))
    return out


def test_registry_interleaved_proofs_and_eviction(device, circuits, zkey, wtns, tmp_path, monkeypatch):
    import ultragroth_amd as ug
    R = ug.Registry
    # -- no budget of its own: everything resident, tables everywhere they apply (>= 2^14 scalars)
    with ug.Registry(0) as reg:
        for name, (zk, _, _) in circuits.items():
            reg.load(name, zk)
        reg.load("fixture", zkey)                                        # the reference's 1 024-constraint circuit beside them
        assert reg.info()[1] == 4
        for name in circuits:
            assert reg.info(name)[1] == R.RESIDENT_WITH_TABLES
        assert reg.info("fixture")[1] == R.RESIDENT                      # too small for tables
        for rnd in range(2):
            for name in ("large", "fixture", "small", "mid", "small"):
                zk, wt, exp = circuits[name] if name != "fixture" else (zkey, wtns, _expected(zkey, wtns))
                assert _prove_fixed(reg.prove, name, wt) == exp, (rnd, name)
        assert reg.info("small")[2] == 4 and reg.info()[2] == 10
        full_bytes = {name: reg.info(name)[0] for name in circuits}
        total = reg.info()[0]
        with pytest.raises(ug.ProverError, match="circuit not loaded: nope"):
            reg.prove("nope", wtns)
        with pytest.raises(ug.ProverError) as e:                         # the prover's own errors pass through
            reg.prove("small", wtns)
        assert e.value.code == ug.PROVER_INVALID_WITNESS_LENGTH
        reg.evict("mid")
        assert reg.info("mid")[1] == R.NOT_LOADED and reg.info()[1] == 3
    # -- a budget that cannot hold every table: the least recently used circuits lose theirs first, proofs stay exact
    budget = int(total * 0.55)
    with ug.Registry(0, budget) as reg:
        for name in ("small", "mid", "large"):
            reg.load(name, circuits[name][0])
        assert reg.info()[0] <= budget
        for name in ("small", "mid", "large", "small", "large", "mid"):
            assert _prove_fixed(reg.prove, name, circuits[name][1]) == circuits[name][2], name
            assert reg.info()[0] <= budget
            assert reg.info()[1] == 3                                    # nobody was evicted as a whole
        states = {name: reg.info(name)[1] for name in circuits}
        assert R.RESIDENT in states.values(), states                     # ... but somebody proves without tables now
    # -- a budget below the three cores: whole circuits go; one that came from a file returns by itself
    paths = {}
    for name in circuits:
        paths[name] = str(tmp_path / (name + ".zkey"))
        open(paths[name], "wb").write(circuits[name][0])
    monkeypatch.setenv("ULTRAGROTH_TABLES", "0")                         # cores alone: bases, CSR matrix, twiddles, vectors
    with ug.Registry(0) as reg:
        for name in circuits:
            reg.load(name, circuits[name][0])
        core = {name: reg.info(name)[0] for name in circuits}
    monkeypatch.delenv("ULTRAGROTH_TABLES")
    assert all(core[n] < full_bytes[n] for n in circuits)
    with ug.Registry(0, int(sum(core.values()) * 0.8)) as reg:
        for name in ("small", "mid", "large"):
            reg.load_file(paths[name])
        resident = [n for n in circuits if reg.info(n)[1] in (R.RESIDENT, R.RESIDENT_WITH_TABLES)]
        evicted = [n for n in circuits if reg.info(n)[1] == R.EVICTED]
        assert "large" in resident and evicted, (resident, evicted)
        for name in ("small", "mid", "large", "small"):
            assert _prove_fixed(reg.prove, name, circuits[name][1]) == circuits[name][2], name
            assert reg.info(name)[1] in (R.RESIDENT, R.RESIDENT_WITH_TABLES)
    with pytest.raises(ug.ProverError, match="does not fit the HBM budget"):
        with ug.Registry(0, 1 << 20) as reg:
            reg.load("small", circuits["small"][0])


def test_registry_from_several_threads_under_pressure(device, circuits):
    """callers on several threads (ctypes drops the GIL inside the calls) prove different circuits while the budget keeps
    the registry giving memory back: a circuit that is proving is never trimmed or evicted under it, proofs stay exact"""
    import threading
    import ultragroth_amd as ug
    with ug.Registry(0) as reg:
        for name in circuits:
            reg.load(name, circuits[name][0])
        for name in circuits:
            _prove_fixed(reg.prove, name, circuits[name][1])
        total = reg.info()[0]
    r, s = fixed_rs()
    ug.set_test_blinding(r + s)                   # an even number of 31-byte draws per proof: every proof sees (r, s)
    try:
        with ug.Registry(0, int(total * 0.5)) as reg:
            for name in circuits:
                reg.load(name, circuits[name][0])
            errors = []

            def worker(name):
                try:
                    for _ in range(4):
                        if reg.prove(name, circuits[name][1]) != circuits[name][2]:
                            errors.append("wrong proof for " + name)
                except Exception as e:              # noqa: BLE001
                    errors.append("%s: %r" % (name, e))
            threads = [threading.Thread(target=worker, args=(n,)) for n in circuits for _ in range(2)]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
            assert not errors, errors
            assert reg.info()[2] == 4 * len(threads)
    finally:
        ug.set_test_blinding(b"")


def test_registry_holds_ultragroth_and_groth16_together(device):
    import ultragroth_amd as ug
    td = os.path.join(GOLDEN, "trapdoor")
    with ug.Registry(0) as reg:
        reg.load_file(os.path.join(td, "ultra.zkey"))
        reg.load_file(os.path.join(td, "groth16.zkey"))
        uw = open(os.path.join(td, "ultra.uwtns"), "rb").read()
        w = open(os.path.join(td, "groth16.wtns"), "rb").read()
        for _ in range(2):
            proof, pub = reg.prove("ultra", uw)
            assert ug.ultra_groth_verify(proof, pub, json.load(open(os.path.join(td, "ultra_vkey.json"))))
            proof, pub = reg.prove("groth16", w)
            assert ug.groth16_verify(proof, pub, json.load(open(os.path.join(td, "groth16_vkey.json"))))
        with pytest.raises(ug.ProverError) as e:
            reg.prove("ultra", w, proof_size=100)
        assert e.value.code == ug.PROVER_ERROR_SHORT_BUFFER and "Minimum size: 1400" in e.value.message


def test_create_from_zkey_file_owns_its_data(device, tmp_path):
    """groth16_prover_create_zkey_file: the file may vanish after create (the reference's prover would be left with
    dangling pointers); big sections go through the pinned staging path (>= 32 MiB)"""
    import ctypes as C
    import ultragroth_amd as ug
    from ultragroth_amd import synth
    zkey, wtns, info = synth.build_circuit(device, 19, mix="C", seed=0x5EED0D00)      # 290 MB zkey: A, B1, H 32 MiB, B2 64 MiB
    path = str(tmp_path / "c19.zkey")
    with open(path, "wb") as f:
        f.write(zkey)
    L = ug.load()
    h = C.c_void_p()
    err = C.create_string_buffer(256)
    assert L.groth16_prover_create_zkey_file(C.byref(h), path.encode(), err, 255) == 0, err.value
    os.remove(path)
    r, s = fixed_rs()
    psz, qsz = C.c_ulonglong(810), C.c_ulonglong(ug.groth16_public_size_for_zkey_buf(zkey))
    proof, pub = C.create_string_buffer(810), C.create_string_buffer(qsz.value)
    ug.set_test_blinding(r + s)
    try:
        rc = L.groth16_prover_prove(h, wtns, len(wtns), proof, C.byref(psz), pub, C.byref(qsz), err, 255)
    finally:
        ug.set_test_blinding(b"")
        L.groth16_prover_destroy(h)
    assert rc == 0, err.value
    from oracle import closed_form
    exp = closed_form.groth16_expected(zkey, wtns, synth.SEEDS, synth.g1_generator_record(), synth.g2_generator_record(),
                                       int.from_bytes(r, "little"), int.from_bytes(s, "little"))
    assert (proof.value.decode(), pub.value.decode()) == exp
    assert L.groth16_prover_create_zkey_file(C.byref(h), path.encode(), err, 255) == 1       # the file is gone: open fails
