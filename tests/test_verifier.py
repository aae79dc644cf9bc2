"""groth16_verify / ultra_groth_verify (include/verifier.h, host code of the C-ABI library): the reference's acceptance
test (.github/workflows/build.yml:69-81: the prover's output verifies, and stops verifying after public[0] -= 1), its error
codes and strings (src/verifier.cpp), and agreement with the independent pure-Python pairing (oracle/pairing.py).
No GPU needed: verification is host work in the reference and here."""
import ctypes as C
import json
import os
import random
import subprocess

import pytest

import oracle as O
from oracle import pairing
import ultragroth_amd as ug
from ultragroth_amd import _lib

VALID, INVALID, ERROR = 0, 1, 2
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    L = ug.load()
    for f in (L.groth16_verify, L.ultra_groth_verify):
        f.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_ulong]
    return L


def _call(fn, proof, inputs, key):
    enc = lambda v: v if isinstance(v, bytes) else (v if isinstance(v, str) else json.dumps(v)).encode()
    err = C.create_string_buffer(256)
    rc = fn(enc(proof), enc(inputs), enc(key), err, 255)
    return rc, err.value.decode()


@pytest.fixture(scope="module")
def fixture_proof(zkey, wtns):
    proof, pub = O.groth16_prove(zkey, wtns, 0x1234567, 0x7654321)[:2]
    return proof, pub


def test_reference_acceptance_test(lib, fixture_proof, vkey):
    proof, pub = fixture_proof
    assert _call(lib.groth16_verify, proof, pub, vkey) == (VALID, "")
    bad = json.loads(pub)
    bad[0] = str(int(bad[0]) - 1)
    assert _call(lib.groth16_verify, proof, bad, vkey)[0] == INVALID
    # inputs are reduced mod r, as E.fr.fromString does
    big = json.loads(pub)
    big[0] = str(int(big[0]) + O.R_MOD)
    assert _call(lib.groth16_verify, proof, big, vkey)[0] == VALID


def test_agrees_with_the_python_pairing_on_tampered_proofs(lib, fixture_proof, vkey):
    proof, pub = fixture_proof
    rng = random.Random(3)
    pj = json.loads(proof)
    other = json.loads(O.groth16_prove(open(os.path.join(ROOT, "tests", "golden", "circuit_final.zkey"), "rb").read(),
                                       open(os.path.join(ROOT, "tests", "golden", "witness.wtns"), "rb").read(), 5, 6)[0])
    cases = []
    for key in ("pi_a", "pi_b", "pi_c"):                       # a valid point of another proof in one slot
        t = dict(pj); t[key] = other[key]; cases.append(t)
    t = dict(pj); t["pi_a"] = ["0", "0", "1"]; cases.append(t)  # infinity: its pair is skipped
    t = dict(pj); t["pi_c"] = [pj["pi_c"][0], str(int(pj["pi_c"][1]) ^ 1), "1"]; cases.append(t)   # off the curve
    cases.append(other)                                         # a different valid proof
    for t in cases:
        exp = pairing.groth16_verify(vkey, pub, t)
        assert _call(lib.groth16_verify, t, pub, vkey)[0] == (VALID if exp else INVALID)
    assert rng is not None


def test_error_codes_and_strings(lib, fixture_proof, vkey):
    proof, pub = fixture_proof
    vk = dict(vkey)
    assert _call(lib.groth16_verify, "{", pub, vkey) == (ERROR, "invalid proof data")                 # verifier.cpp:31-33
    assert _call(lib.groth16_verify, proof.replace("groth16", "plonk"), pub, vkey) == (ERROR, "invalid proof data")
    assert _call(lib.groth16_verify, {"protocol": "groth16"}, pub, vkey) == (ERROR, "invalid proof data")
    assert _call(lib.groth16_verify, proof, "[]", vkey) == (ERROR, "invalid inputs data")             # :71-73
    assert _call(lib.groth16_verify, proof, "[1]", vkey) == (ERROR, "invalid inputs data")            # not strings
    assert _call(lib.groth16_verify, proof, '["12x"]', vkey) == (ERROR, "invalid inputs data")
    assert _call(lib.groth16_verify, proof, json.loads(pub) + ["1"], vkey) == (ERROR, "len(inputs)+1 != len(vk.IC)")
    for mut in ({"protocol": "ultragroth"}, {"curve": "bls12381"}, {"IC": []}):
        k = dict(vk); k.update(mut)
        assert _call(lib.groth16_verify, proof, pub, k) == (ERROR, "invalid verification key data")  # :104-114
    k = dict(vk); del k["nPublic"]
    assert _call(lib.groth16_verify, proof, pub, k) == (ERROR, "invalid verification key data")
    assert _call(lib.ultra_groth_verify, proof, pub, vkey) == (ERROR, "invalid proof data")           # protocol mismatch
    assert lib.groth16_verify(proof.encode(), pub.encode(), json.dumps(vkey).encode(), None, 0) == VALID


# ---- an UltraGroth instance built in the exponent (the reference ships no protocol-1337 fixture) -------------------------
G1 = (1, 2)
G2 = ((10857046999023057135944570762232829481370756359578518086990519993285655852781,
       11559732032986387107991004021392285783925812861821192530917403151452391805634),
      (8495653923123431417604973247489272438418190587263600148770280649306958101930,
       4082367875863433681332203403145435568316851327593401208105741076214120093531))


def _g2_mul(k):
    acc, base = None, G2
    while k:
        if k & 1:
            acc = pairing.g2_add(acc, base)
        base = pairing.g2_dbl(base)
        k >>= 1
    return acc


def _j1(p):
    return ["0", "1", "0"] if p is None else [str(p[0]), str(p[1]), "1"]


def _j2(q):
    return [[str(q[0][0]), str(q[0][1])], [str(q[1][0]), str(q[1][1])], ["1", "0"]]


def _ultra_instance(seed):
    rng = random.Random(seed)
    R = pairing.R
    s = {k: rng.randrange(1, R) for k in ("alpha", "beta", "gamma", "d1", "d2", "ic0", "ic1", "ic2", "icr", "a", "b", "rr")}
    inputs = [rng.randrange(R), rng.randrange(1 << 64)]
    rnd = pairing.g1_mul(G1, s["rr"])
    c = pairing.derive_challenge(rnd)
    vkx = (s["ic0"] + inputs[0] * s["ic1"] + inputs[1] * s["ic2"] + c * s["icr"]) % R
    f = (s["a"] * s["b"] - s["alpha"] * s["beta"] - vkx * s["gamma"] - s["rr"] * s["d1"]) * pow(s["d2"], -1, R) % R
    proof = {"pi_a": _j1(pairing.g1_mul(G1, s["a"])), "pi_b": _j2(_g2_mul(s["b"])), "pi_f": _j1(pairing.g1_mul(G1, f)),
             "pi_r": _j1(rnd), "protocol": "ultragroth"}
    vk = {"protocol": "ultragroth", "curve": "bn128", "nPublic": 2,
          "vk_alpha_1": _j1(pairing.g1_mul(G1, s["alpha"])), "vk_beta_2": _j2(_g2_mul(s["beta"])),
          "vk_gamma_2": _j2(_g2_mul(s["gamma"])), "vk_delta_c1_2": _j2(_g2_mul(s["d1"])), "vk_delta_c2_2": _j2(_g2_mul(s["d2"])),
          "IC": [_j1(pairing.g1_mul(G1, s[k])) for k in ("ic0", "ic1", "ic2")], "IC_rand": _j1(pairing.g1_mul(G1, s["icr"]))}
    return proof, [str(v) for v in inputs], vk


def test_ultragroth_instance_in_the_exponent(lib):
    proof, inputs, vk = _ultra_instance(11)
    assert pairing.ultra_groth_verify(vk, inputs, proof)                      # the construction is right
    assert _call(lib.ultra_groth_verify, proof, inputs, vk) == (VALID, "")
    bad = [str(int(inputs[0]) + 1), inputs[1]]
    assert _call(lib.ultra_groth_verify, proof, bad, vk)[0] == INVALID
    # a different round commitment changes the challenge: invalid although every point is on its curve
    proof2, _, _ = _ultra_instance(12)
    t = dict(proof); t["pi_r"] = proof2["pi_r"]
    assert not pairing.ultra_groth_verify(vk, inputs, t)
    assert _call(lib.ultra_groth_verify, t, inputs, vk)[0] == INVALID
    k = dict(vk); k["vk_delta_c1_2"], k["vk_delta_c2_2"] = vk["vk_delta_c2_2"], vk["vk_delta_c1_2"]      # swapped deltas
    assert _call(lib.ultra_groth_verify, proof, inputs, k)[0] == INVALID
    assert _call(lib.ultra_groth_verify, proof, inputs[:1], vk) == (ERROR, "len(inputs) != len(vk.IC)")   # ultra_groth.cpp:585-587
    k = dict(vk); del k["IC_rand"]
    assert _call(lib.ultra_groth_verify, proof, inputs, k) == (ERROR, "invalid verification key data")
    assert _call(lib.groth16_verify, proof, inputs, vk) == (ERROR, "invalid proof data")


def test_verifier_cli(tmp_path, lib, fixture_proof, vkey):
    """`verifier <verification_key.json> <inputs.json> <proof.json>` (src/main_verifier.cpp): messages and exit codes"""
    exe = os.path.join(ROOT, "ultragroth_amd", "csrc", "verifier")
    proof, pub = fixture_proof
    (tmp_path / "proof.json").write_text(proof)
    (tmp_path / "public.json").write_text(pub)
    (tmp_path / "vk.json").write_text(json.dumps(vkey))
    r = subprocess.run([exe, str(tmp_path / "vk.json"), str(tmp_path / "public.json"), str(tmp_path / "proof.json")], capture_output=True, text=True)
    assert r.returncode == 0 and r.stderr == "Result: Valid proof\n"
    bad = json.loads(pub); bad[0] = str(int(bad[0]) - 1)
    (tmp_path / "bad.json").write_text(json.dumps(bad))
    r = subprocess.run([exe, str(tmp_path / "vk.json"), str(tmp_path / "bad.json"), str(tmp_path / "proof.json")], capture_output=True, text=True)
    assert r.returncode == 1 and r.stderr == "Result: Invalid proof\n"
    r = subprocess.run([exe, str(tmp_path / "vk.json")], capture_output=True, text=True)
    assert r.returncode == 1 and r.stderr.startswith("Invalid number of parameters:\nUsage: verifier <verification_key.json> <inputs.json> <proof.json>")
    r = subprocess.run([exe + "_ultra_groth", str(tmp_path / "vk.json"), str(tmp_path / "public.json"), str(tmp_path / "proof.json")], capture_output=True, text=True)
    assert r.returncode == 1 and r.stderr == "Error: invalid proof data\n"
