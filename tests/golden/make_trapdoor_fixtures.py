#!/usr/bin/env python3
"""Generator of the trapdoor fixtures under tests/golden/trapdoor/ -- TEST TOOLING (own code, no reference source).

The reference ships ONE fixture (testdata/, Groth16) and none for UltraGroth (protocol 1337): no zkey, no .uwtns,
no test. This script makes a small REAL instance of both protocols from a known trapdoor, so that proofs of the
product are judged by the reference's own acceptance criterion -- the verifier equations of
src/groth16.cpp:314-364 and src/ultra_groth.cpp:582-648 -- instead of by comparison with our own oracle only:

    circuit.json                 sizes and signal map of the circuit (a log-derivative lookup argument, below)
    ultra.zkey, ultra.uwtns      protocol-1337 proving key and first-round witness (lookup signals still zero)
    ultra_vkey.json              its verification key (vk_delta_c1_2 = round delta, vk_delta_c2_2 = final delta, IC_rand)
    groth16.zkey, groth16.wtns   the same circuit as a plain Groth16 instance (challenge fixed, witness complete)
    groth16_vkey.json
    trapdoor.json                tau, alpha, beta, gamma, deltas (this is a TEST setup: the toxic waste is the point)

Circuit (N = 2^10 domain). Signals: 0 one | 1 out (public) | 2 rho (public; rand_indx of UltraGroth) |
x, y | c_j (J chunk values, each in [0, L)) | f_i (L table frequencies) | inv1_j (J) | inv2_i (L) | prod_i (L).
Constraints (rows):
    x * y = out
    (c_j + rho) * inv1_j = 1                  j < J     -- inv1_j = 1 / (c_j + rho)
    (i + rho)   * inv2_i = 1                  i < L     -- inv2_i = 1 / (i + rho)
    f_i         * inv2_i = prod_i             i < L
    (sum_j inv1_j - sum_i prod_i) * 1 = 0               -- the lookup identity: every c_j is a table row
plus snarkjs' public-input rows (signal s at row nConstraints + s, A side only). inv1, inv2, prod and rho are exactly
what UltraGroth's compute_lookup (src/ultra_groth.cpp:62-106) writes from `chunks`, `frequencies` and the challenge:
push_vector = [rho | inv1 | inv2 | prod]; the .uwtns index sections map them to their signals. Round-1 signals
(committed before the challenge): x, y, c_j, f_i. Final-round signals: inv1, inv2, prod.

Setup (Groth16 with two deltas, as the verifier equation requires):
    A_s = u_s(tau) G1, B1_s = v_s(tau) G1, B2_s = v_s(tau) G2, K_s = beta u_s + alpha v_s + w_s at tau,
    IC_s = K_s / gamma (public s; IC_rand for s = rand_indx), C1 = K_s / delta_round (round signals),
    C2 = K_s / delta_final (final signals), H_k = L_k(tau / g) (tau^N - 1) / (-2 delta_final) with g = omega_2N:
the prover's h scalars are (a b - c) on the odd coset g omega^k, where x^N - 1 = -2, so
sum_k h_k H_k = h(tau) t(tau) / delta_final.

Run from the repository root:  python tests/golden/make_trapdoor_fixtures.py
Uses the CPU oracle (oracle/) only for scalar multiplications of the generators.
"""
import hashlib
import json
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle as O  # noqa: E402

R = O.R_MOD
Q = O.Q_MOD
MONT = 1 << 256
OUT = os.environ.get("UG_TRAPDOOR_OUT") or os.path.join(ROOT, "tests", "golden", "trapdoor")

G1 = (1, 2)
G2 = ((10857046999023057135944570762232829481370756359578518086990519993285655852781,
       11559732032986387107991004021392285783925812861821192530917403151452391805634),
      (8495653923123431417604973247489272438418190587263600148770280649306958101930,
       4082367875863433681332203403145435568316851327593401208105741076214120093531))


def _m(x):
    return (x * MONT % Q).to_bytes(32, "little")


G1_REC = _m(G1[0]) + _m(G1[1])
G2_REC = _m(G2[0][0]) + _m(G2[0][1]) + _m(G2[1][0]) + _m(G2[1][1])


def g1(k):
    k %= R
    return O.g1_mul(G1_REC, k) if k else bytes(64)


def g2(k):
    k %= R
    return O.g2_mul(G2_REC, k) if k else bytes(128)


def det(label):
    """deterministic field element from a label (fixtures must be reproducible)"""
    return int.from_bytes(hashlib.sha256(b"ultragroth-amd trapdoor fixture: " + label.encode()).digest(), "big") % R


def rec_to_json_g1(rec):
    if rec == bytes(64):
        return ["0", "1", "0"]
    return [str(O.mont_decode(rec[:32])), str(O.mont_decode(rec[32:])), "1"]


def rec_to_json_g2(rec):
    c = [str(O.mont_decode(rec[32 * k:32 * k + 32])) for k in range(4)]
    return [[c[0], c[1]], [c[2], c[3]], ["1", "0"]]


def section(sid, payload):
    return struct.pack("<IQ", sid, len(payload)) + payload


def binfile(magic, version, sections):
    return magic + struct.pack("<II", version, len(sections)) + b"".join(section(i, p) for i, p in sections)


# ------------------------------------------------------------------------------------------- circuit
LOG_N = 10
N = 1 << LOG_N
J, L = 300, 64                                 # chunks, table rows
ONE, OUT_S, RHO = 0, 1, 2
N_PUBLIC = 2
X, Y = 3, 4
C0 = 5
F0 = C0 + J
INV1 = F0 + L
INV2 = INV1 + J
PROD = INV2 + L
N_VARS = PROD + L


def build_r1cs():
    """rows of (A, B, C), each a dict signal -> coefficient"""
    rows = [({X: 1}, {Y: 1}, {OUT_S: 1})]
    for j in range(J):
        rows.append(({C0 + j: 1, RHO: 1}, {INV1 + j: 1}, {ONE: 1}))
    for i in range(L):
        rows.append(({ONE: i, RHO: 1} if i else {RHO: 1}, {INV2 + i: 1}, {ONE: 1}))
    for i in range(L):
        rows.append(({F0 + i: 1}, {INV2 + i: 1}, {PROD + i: 1}))
    lin = {INV1 + j: 1 for j in range(J)}
    lin.update({PROD + i: R - 1 for i in range(L)})
    rows.append((lin, {ONE: 1}, {}))
    n_constraints = len(rows)
    for s in range(N_PUBLIC + 1):              # snarkjs: one extra row per public signal (and signal 0), A side only
        rows.append(({s: 1}, {}, {}))
    assert len(rows) <= N
    return rows, n_constraints


def first_round_witness():
    """signal values known before the challenge; lookup signals are zero (src/README.md: 'zeros for signals belonging to
    the second round')"""
    w = [0] * N_VARS
    w[ONE] = 1
    w[X], w[Y] = det("x"), det("y")
    w[OUT_S] = w[X] * w[Y] % R
    chunks = [det("chunk %d" % j) % L for j in range(J)]
    freq = [0] * L
    for c in chunks:
        freq[c] += 1
    for j in range(J):
        w[C0 + j] = chunks[j]
    for i in range(L):
        w[F0 + i] = freq[i]
    return w, chunks, freq


def complete_witness(w, chunks, freq, rho):
    """what compute_lookup writes for challenge rho (src/ultra_groth.cpp:62-106)"""
    w = list(w)
    w[RHO] = rho
    inv2 = [pow((i + rho) % R, -1, R) for i in range(L)]
    for j in range(J):
        w[INV1 + j] = inv2[chunks[j]]
    for i in range(L):
        w[INV2 + i] = inv2[i]
        w[PROD + i] = freq[i] * inv2[i] % R
    return w


def check_r1cs(rows, w):
    dot = lambda row: sum(c * w[s] for s, c in row.items()) % R
    for k, (a, b, c) in enumerate(rows):
        assert dot(a) * dot(b) % R == dot(c), "constraint %d" % k


# ------------------------------------------------------------------------------------------- setup
def lagrange_at(tau, shift=1):
    """L_k(tau / shift) for the size-N domain, k < N"""
    x = tau * pow(shift, -1, R) % R
    omega = pow(5, (R - 1) // N, R)
    zn = (pow(x, N, R) - 1) % R
    ninv = pow(N, -1, R)
    out, wk = [], 1
    for k in range(N):
        out.append(zn * wk % R * ninv % R * pow((x - wk) % R, -1, R) % R)
        wk = wk * omega % R
    return out


def setup(rows, ultra, round_signals, final_signals):
    t = {k: det("trapdoor " + k) for k in ("tau", "alpha", "beta", "gamma", "delta_round", "delta_final")}
    tau, alpha, beta, gamma = t["tau"], t["alpha"], t["beta"], t["gamma"]
    d_round, d_final = t["delta_round"], t["delta_final"]
    lag = lagrange_at(tau)
    u, v, wv = [0] * N_VARS, [0] * N_VARS, [0] * N_VARS
    for k, (a, b, c) in enumerate(rows):
        for s, co in a.items():
            u[s] = (u[s] + co * lag[k]) % R
        for s, co in b.items():
            v[s] = (v[s] + co * lag[k]) % R
        for s, co in c.items():
            wv[s] = (wv[s] + co * lag[k]) % R
    K = [(beta * u[s] + alpha * v[s] + wv[s]) % R for s in range(N_VARS)]
    g = pow(5, (R - 1) // (2 * N), R)                                   # omega_2N: the prover's coset shift
    lag_coset = lagrange_at(tau, g)
    h_scale = (pow(tau, N, R) - 1) * pow((-2 * d_final) % R, -1, R) % R
    pts = {
        "A": b"".join(g1(u[s]) for s in range(N_VARS)),
        "B1": b"".join(g1(v[s]) for s in range(N_VARS)),
        "B2": b"".join(g2(v[s]) for s in range(N_VARS)),
        "H": b"".join(g1(lag_coset[k] * h_scale) for k in range(N)),
        "IC": b"".join(g1(K[s] * pow(gamma, -1, R)) for s in range(N_PUBLIC + 1)),
    }
    if ultra:
        pts["C1"] = b"".join(g1(K[s] * pow(d_round, -1, R)) for s in round_signals)
        pts["C2"] = b"".join(g1(K[s] * pow(d_final, -1, R)) for s in final_signals)
    else:
        pts["C"] = b"".join(g1(K[s] * pow(d_final, -1, R)) for s in range(N_PUBLIC + 1, N_VARS))
    vk = {"alpha1": g1(alpha), "beta1": g1(beta), "beta2": g2(beta), "gamma2": g2(gamma),
          "round_delta1": g1(d_round), "round_delta2": g2(d_round), "final_delta1": g1(d_final), "final_delta2": g2(d_final)}
    return t, pts, vk


def coef_section(rows):
    recs = []
    for k, (a, b, _) in enumerate(rows):
        for m, row in ((0, a), (1, b)):
            for s, co in sorted(row.items()):
                recs.append(struct.pack("<III", m, k, s) + (co * MONT * MONT % R).to_bytes(32, "little"))
    return struct.pack("<I", len(recs)) + b"".join(recs)


def zkey_header(ultra, vk, n_c1=0, n_c2=0):
    h = struct.pack("<I", 32) + Q.to_bytes(32, "little") + struct.pack("<I", 32) + R.to_bytes(32, "little")
    h += struct.pack("<III", N_VARS, N_PUBLIC, N)
    if ultra:
        h += struct.pack("<III", n_c1, n_c2, RHO)
    h += vk["alpha1"] + vk["beta1"] + vk["beta2"] + vk["gamma2"]
    if ultra:
        h += vk["round_delta1"] + vk["round_delta2"]
    return h + vk["final_delta1"] + vk["final_delta2"]


def wtns_file(w, extra=()):
    secs = [(1, struct.pack("<I", 32) + R.to_bytes(32, "little") + struct.pack("<I", len(w))),
            (2, b"".join(x.to_bytes(32, "little") for x in w))] + list(extra)
    return binfile(b"wtns", 2, secs)


def main():
    os.makedirs(OUT, exist_ok=True)
    rows, n_constraints = build_r1cs()
    w1, chunks, freq = first_round_witness()
    round_signals = [X, Y] + [C0 + j for j in range(J)] + [F0 + i for i in range(L)]
    final_signals = [INV1 + j for j in range(J)] + [INV2 + i for i in range(L)] + [PROD + i for i in range(L)]
    assert sorted(round_signals + final_signals) == list(range(N_PUBLIC + 1, N_VARS))
    check_r1cs(rows, complete_witness(w1, chunks, freq, det("any challenge")))      # satisfiable for EVERY challenge

    # ---- UltraGroth
    t, pts, vk = setup(rows, True, round_signals, final_signals)
    u32s = lambda xs: b"".join(struct.pack("<I", x) for x in xs)
    zkey = binfile(b"zkey", 1, [
        (1, struct.pack("<I", 1337)), (2, zkey_header(True, vk, len(round_signals), len(final_signals))), (3, pts["IC"]),
        (4, coef_section(rows)), (5, pts["A"]), (6, pts["B1"]), (7, pts["B2"]), (8, pts["C1"]), (9, pts["C2"]),
        (10, u32s(round_signals)), (11, u32s(final_signals)), (12, pts["H"]), (13, b"")])
    # push_vector = [rho | inv1 (J) | inv2 (L) | prod (L)]  (src/ultra_groth.cpp:88-98)
    wtns_idx = [RHO] + [INV1 + j for j in range(J)] + [INV2 + i for i in range(L)] + [PROD + i for i in range(L)]
    push_idx = list(range(1 + J + 2 * L))
    uwtns = wtns_file(w1, [(3, u32s(chunks)), (4, u32s(freq)), (5, u32s(wtns_idx)), (6, u32s(push_idx))])
    ic = [pts["IC"][64 * s:64 * s + 64] for s in range(N_PUBLIC + 1)]
    vkey = {"protocol": "ultragroth", "curve": "bn128", "nPublic": N_PUBLIC - 1,
            "vk_alpha_1": rec_to_json_g1(vk["alpha1"]), "vk_beta_2": rec_to_json_g2(vk["beta2"]),
            "vk_gamma_2": rec_to_json_g2(vk["gamma2"]), "vk_delta_c1_2": rec_to_json_g2(vk["round_delta2"]),
            "vk_delta_c2_2": rec_to_json_g2(vk["final_delta2"]),
            # inputs = public signals without rand_indx (src/prover.cpp:89-105): IC holds signals 0 and 1, IC_rand signal 2
            "IC": [rec_to_json_g1(ic[0]), rec_to_json_g1(ic[1])], "IC_rand": rec_to_json_g1(ic[RHO])}
    open(os.path.join(OUT, "ultra.zkey"), "wb").write(zkey)
    open(os.path.join(OUT, "ultra.uwtns"), "wb").write(uwtns)
    json.dump(vkey, open(os.path.join(OUT, "ultra_vkey.json"), "w"), indent=1)

    # ---- the same circuit as plain Groth16 (one delta; challenge fixed so that the witness is complete)
    rho = det("groth16 twin challenge")
    w = complete_witness(w1, chunks, freq, rho)
    check_r1cs(rows, w)
    _, pts16, vk16 = setup(rows, False, None, None)
    zkey16 = binfile(b"zkey", 1, [
        (1, struct.pack("<I", 1)), (2, zkey_header(False, vk16)), (3, pts16["IC"]), (4, coef_section(rows)),
        (5, pts16["A"]), (6, pts16["B1"]), (7, pts16["B2"]), (8, pts16["C"]), (9, pts16["H"]), (10, b"")])
    ic16 = [pts16["IC"][64 * s:64 * s + 64] for s in range(N_PUBLIC + 1)]
    vkey16 = {"protocol": "groth16", "curve": "bn128", "nPublic": N_PUBLIC,
              "vk_alpha_1": rec_to_json_g1(vk16["alpha1"]), "vk_beta_2": rec_to_json_g2(vk16["beta2"]),
              "vk_gamma_2": rec_to_json_g2(vk16["gamma2"]), "vk_delta_2": rec_to_json_g2(vk16["final_delta2"]),
              "IC": [rec_to_json_g1(p) for p in ic16]}
    open(os.path.join(OUT, "groth16.zkey"), "wb").write(zkey16)
    open(os.path.join(OUT, "groth16.wtns"), "wb").write(wtns_file(w))
    json.dump(vkey16, open(os.path.join(OUT, "groth16_vkey.json"), "w"), indent=1)

    json.dump({k: str(v) for k, v in t.items()}, open(os.path.join(OUT, "trapdoor.json"), "w"), indent=1)
    json.dump({"nVars": N_VARS, "nPublic": N_PUBLIC, "domainSize": N, "nConstraints": n_constraints, "rand_indx": RHO,
               "chunks": J, "lookup_size": L,
               "signals": {"one": ONE, "out": OUT_S, "rho": RHO, "x": X, "y": Y, "c": [C0, C0 + J], "f": [F0, F0 + L],
                           "inv1": [INV1, INV1 + J], "inv2": [INV2, INV2 + L], "prod": [PROD, PROD + L]},
               "constraints": "see the module docstring of make_trapdoor_fixtures.py"},
              open(os.path.join(OUT, "circuit.json"), "w"), indent=1)
    print("wrote", OUT, {f: os.path.getsize(os.path.join(OUT, f)) for f in sorted(os.listdir(OUT))})


if __name__ == "__main__":
    main()
