"""Real Groth16 / UltraGroth instances with a known trapdoor (tests/golden/trapdoor/, made by
tests/golden/make_trapdoor_fixtures.py): the proofs are judged by the reference's acceptance criterion, the verifier
equations of src/groth16.cpp:314-364 and src/ultra_groth.cpp:582-648, not only by comparison with the oracle.

The reference has no protocol-1337 fixture at all; this one pins the UltraGroth-specific steps together -- a proof
verifies only if the round commitment (execute_round, :161-184), the Fiat-Shamir challenge (derive_challenge, :33-58),
every value compute_lookup writes (:62-106; the circuit's constraints are the lookup argument over exactly those
values) and the final round with its - r_k * round_delta1 term (:386-388) are all right.

CPU part: the oracle's proofs. GPU part (-m gpu): the product's proofs, with OS entropy and with fixed blinding."""
import json
import os
import subprocess
import sys

import pytest

import oracle as O
from oracle import pairing
from conftest import GOLDEN, ROOT

TD = os.path.join(GOLDEN, "trapdoor")


def _load(name, mode="rb"):
    with open(os.path.join(TD, name), mode) as f:
        return f.read()


@pytest.fixture(scope="module")
def ultra():
    return _load("ultra.zkey"), _load("ultra.uwtns"), json.loads(_load("ultra_vkey.json", "r"))


@pytest.fixture(scope="module")
def twin():
    return _load("groth16.zkey"), _load("groth16.wtns"), json.loads(_load("groth16_vkey.json", "r"))


def _both_verifiers_ultra(vk, pub, proof):
    """pure-Python pairing (oracle/pairing.py) and the product's host verifier (include/verifier.h) must agree"""
    import ultragroth_amd as ug
    a = pairing.ultra_groth_verify(vk, json.loads(pub) if isinstance(pub, str) else pub, json.loads(proof) if isinstance(proof, str) else proof)
    b = ug.ultra_groth_verify(proof, pub, vk)
    assert a == b
    return a


def _both_verifiers_groth16(vk, pub, proof):
    import ultragroth_amd as ug
    a = pairing.groth16_verify(vk, pub, proof)
    b = ug.groth16_verify(proof, pub, vk)
    assert a == b
    return a


def test_fixtures_are_what_the_generator_makes(tmp_path):
    """the committed files are reproducible from the committed script"""
    env = dict(os.environ, UG_TRAPDOOR_OUT=str(tmp_path))
    subprocess.run([sys.executable, os.path.join(GOLDEN, "make_trapdoor_fixtures.py")], check=True, env=env, capture_output=True)
    for name in sorted(os.listdir(TD)):
        assert (tmp_path / name).read_bytes() == _load(name), name


def test_oracle_ultragroth_proof_is_accepted_by_the_verifier_equation(ultra):
    zkey, uwtns, vk = ultra
    info = O.zkey_info(zkey)
    assert info["ultra"] and info["nPublic"] == 2
    for rk, r, s in ((111, 222, 333), (0, 0, 0), (2 ** 248 - 1, 2 ** 248 - 2, 2 ** 248 - 3)):
        proof, pub = O.ultra_groth_prove(zkey, uwtns, rk, r, s)
        assert len(json.loads(pub)) == 1                                  # rand_indx is not a public input of the verifier
        assert _both_verifiers_ultra(vk, pub, proof)
        bad = json.loads(pub); bad[0] = str(int(bad[0]) - 1)              # the reference's CI tamper (build.yml:69-81)
        assert not _both_verifiers_ultra(vk, json.dumps(bad), proof)
    # a proof is bound to ITS round commitment: pi_r of another proof changes the challenge
    p1 = json.loads(O.ultra_groth_prove(zkey, uwtns, 1, 2, 3)[0])
    p2 = json.loads(O.ultra_groth_prove(zkey, uwtns, 4, 5, 6)[0])
    mixed = dict(p1); mixed["pi_r"] = p2["pi_r"]
    assert not _both_verifiers_ultra(vk, pub, json.dumps(mixed))
    # the two deltas are not interchangeable (execute_round blinds with final_delta1, the final round subtracts round_delta1)
    swapped = dict(vk); swapped["vk_delta_c1_2"], swapped["vk_delta_c2_2"] = vk["vk_delta_c2_2"], vk["vk_delta_c1_2"]
    assert not _both_verifiers_ultra(swapped, pub, json.dumps(p2))


def test_a_wrong_lookup_value_breaks_the_proof(ultra):
    """the lookup argument is live: one frequency off by one (so that prod_i no longer matches the chunks) -> invalid"""
    zkey, uwtns, vk = ultra
    off, sz = O.section(uwtns, "wtns", 4)
    bad = bytearray(uwtns)
    bad[off] ^= 1                                                          # frequencies[0] +- 1 in the lookup section only
    proof, pub = O.ultra_groth_prove(zkey, bytes(bad), 7, 8, 9)
    assert not _both_verifiers_ultra(vk, pub, proof)


def test_oracle_groth16_twin_is_accepted(twin):
    zkey, wtns, vk = twin
    proof, pub = O.groth16_prove(zkey, wtns, 12345, 67890)
    assert _both_verifiers_groth16(vk, pub, proof)
    bad = json.loads(pub); bad[1] = str(int(bad[1]) + 1)
    assert not _both_verifiers_groth16(vk, json.dumps(bad), proof)


# ------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_gpu_ultragroth_proof_is_accepted_by_the_verifier_equation(device, ultra):
    import ultragroth_amd as ug
    zkey, uwtns, vk = ultra
    with ug.UltraGrothProver(zkey) as p:
        seen = set()
        for _ in range(3):                                                  # OS entropy: three different valid proofs
            proof, pub = p.prove(uwtns)
            assert _both_verifiers_ultra(vk, pub, proof)
            seen.add(proof)
            bad = json.loads(pub); bad[0] = str(int(bad[0]) - 1)
            assert not _both_verifiers_ultra(vk, json.dumps(bad), proof)
        assert len(seen) == 3
        rk, r, s = bytes(range(1, 32)), bytes(range(40, 71)), bytes(range(80, 111))
        ug.set_test_blinding(rk + r + s)
        try:
            got = p.prove(uwtns)
        finally:
            ug.set_test_blinding(b"")
    assert got == O.ultra_groth_prove(zkey, uwtns, *(int.from_bytes(b, "little") for b in (rk, r, s)))
    assert ug.ultra_groth_prover(zkey, uwtns)[1] == got[1]                 # one-shot entry point, same public.json


@pytest.mark.gpu
def test_gpu_ultragroth_cli_end_to_end(tmp_path, ultra):
    """prover_ultra_groth <zkey> <uwtns> <proof.json> <public.json>, then verifier_ultra_groth on the files it wrote"""
    csrc = os.path.join(ROOT, "ultragroth_amd", "csrc")
    proof_path, public_path = str(tmp_path / "proof.json"), str(tmp_path / "public.json")
    r = subprocess.run([os.path.join(csrc, "prover_ultra_groth"), os.path.join(TD, "ultra.zkey"), os.path.join(TD, "ultra.uwtns"),
                        proof_path, public_path], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([os.path.join(csrc, "verifier_ultra_groth"), os.path.join(TD, "ultra_vkey.json"), public_path, proof_path],
                       capture_output=True, text=True)
    assert r.returncode == 0 and r.stderr == "Result: Valid proof\n"
    bad = json.loads(open(public_path).read()); bad[0] = str(int(bad[0]) - 1)
    (tmp_path / "bad.json").write_text(json.dumps(bad))
    r = subprocess.run([os.path.join(csrc, "verifier_ultra_groth"), os.path.join(TD, "ultra_vkey.json"), str(tmp_path / "bad.json"), proof_path],
                       capture_output=True, text=True)
    assert r.returncode == 1 and r.stderr == "Result: Invalid proof\n"


@pytest.mark.gpu
def test_gpu_groth16_twin_is_accepted(device, twin):
    import ultragroth_amd as ug
    zkey, wtns, vk = twin
    proof, pub = ug.groth16_prover(zkey, wtns)
    assert _both_verifiers_groth16(vk, pub, proof)
    bad = json.loads(pub); bad[0] = str(int(bad[0]) - 1)
    assert not _both_verifiers_groth16(vk, json.dumps(bad), proof)
