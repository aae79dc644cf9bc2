"""Pure-Python BN254 (alt_bn128) Groth16 verifier -- TEST INFRASTRUCTURE ONLY.

Restates the reference's acceptance test: ``verifier verification_key.json public.json
proof.json`` must accept the prover's output and reject it after ``public[0] -= 1``
(.github/workflows/build.yml:69-81). The check itself is the one in
``Groth16::Verifier::verify`` (src/groth16.cpp:314-364):

    vkX = IC[0] + sum_i public[i] * IC[i+1]
    e(A, B) * e(-alpha1, beta2) * e(-vkX, gamma2) * e(-C, delta2) == 1

Textbook arithmetic on Python ints (no reference code): Fp2 = Fp[u]/(u^2+1),
Fp12 = Fp[w]/(w^12 - 18 w^6 + 82) with u = w^6 - 9, D-type twist (x, y) -> (x w^2, y w^3),
optimal-ate Miller loop over 6t+2 = 29793968203157093288 plus the two Frobenius lines,
final exponent (p^12 - 1)/r by square-and-multiply.
"""
import json

P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
ATE = 29793968203157093288
XI = (9, 1)  # 9 + u


# ---- Fp2 --------------------------------------------------------------------------------
def f2_add(a, b): return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)
def f2_sub(a, b): return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)
def f2_neg(a): return ((-a[0]) % P, (-a[1]) % P)
def f2_mul(a, b): return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)
def f2_muls(a, k): return (a[0] * k % P, a[1] * k % P)
def f2_conj(a): return (a[0], (-a[1]) % P)


def f2_inv(a):
    n = pow((a[0] * a[0] + a[1] * a[1]) % P, -1, P)
    return (a[0] * n % P, (-a[1]) * n % P)


def f2_pow(a, e):
    r = (1, 0)
    while e:
        if e & 1:
            r = f2_mul(r, a)
        a = f2_mul(a, a)
        e >>= 1
    return r


# ---- Fp12 as 12 coefficients in w -------------------------------------------------------
def f12_mul(a, b):
    t = [0] * 23
    for i, ai in enumerate(a):
        if ai:
            for j, bj in enumerate(b):
                t[i + j] += ai * bj
    for k in range(22, 11, -1):       # w^12 = 18 w^6 - 82
        c = t[k]
        if c:
            t[k - 6] += 18 * c
            t[k - 12] -= 82 * c
    return [x % P for x in t[:12]]


F12_ONE = [1] + [0] * 11


def f12_pow(a, e):
    r = F12_ONE
    for bit in bin(e)[2:]:
        r = f12_mul(r, r)
        if bit == "1":
            r = f12_mul(r, a)
    return r


def _embed(c, k):
    """(a + b u) * w^k with u = w^6 - 9  ->  sparse {power: coeff}"""
    return {k: (c[0] - 9 * c[1]) % P, k + 6: c[1] % P}


def _sparse_to_f12(d):
    out = [0] * 12
    for k, v in d.items():
        out[k] = (out[k] + v) % P
    return out


# ---- G2 on the twist y^2 = x^3 + 3/(9+u), affine, None = infinity -----------------------
def g2_add(p, q):
    if p is None: return q
    if q is None: return p
    if p[0] == q[0]:
        if p[1] == q[1]:
            return g2_dbl(p)
        return None
    m = f2_mul(f2_sub(q[1], p[1]), f2_inv(f2_sub(q[0], p[0])))
    x = f2_sub(f2_sub(f2_mul(m, m), p[0]), q[0])
    return (x, f2_sub(f2_mul(m, f2_sub(p[0], x)), p[1]))


def g2_dbl(p):
    if p is None or p[1] == (0, 0): return None
    m = f2_mul(f2_muls(f2_mul(p[0], p[0]), 3), f2_inv(f2_muls(p[1], 2)))
    x = f2_sub(f2_mul(m, m), f2_muls(p[0], 2))
    return (x, f2_sub(f2_mul(m, f2_sub(p[0], x)), p[1]))


def _line(t1, t2, pt):
    """line through twist points t1, t2 (tangent when equal), evaluated at pt in G1 -> Fp12"""
    xp, yp = pt
    if t1[0] != t2[0]:
        m = f2_mul(f2_sub(t2[1], t1[1]), f2_inv(f2_sub(t2[0], t1[0])))
    elif t1[1] == t2[1]:
        m = f2_mul(f2_muls(f2_mul(t1[0], t1[0]), 3), f2_inv(f2_muls(t1[1], 2)))
    else:  # vertical: xP - x1 w^2
        d = {0: xp}
        for k, v in _embed(f2_neg(t1[0]), 2).items():
            d[k] = (d.get(k, 0) + v) % P
        return _sparse_to_f12(d)
    # -yP + (m xP) w + (y1 - m x1) w^3
    d = {0: (-yp) % P}
    for k, v in _embed(f2_muls(m, xp), 1).items():
        d[k] = (d.get(k, 0) + v) % P
    for k, v in _embed(f2_sub(t1[1], f2_mul(m, t1[0])), 3).items():
        d[k] = (d.get(k, 0) + v) % P
    return _sparse_to_f12(d)


_G12 = f2_pow(XI, (P - 1) // 3)
_G13 = f2_pow(XI, (P - 1) // 2)
_G22 = f2_pow(XI, (P * P - 1) // 3)
_G23 = f2_pow(XI, (P * P - 1) // 2)


def miller(q, pt):
    """Miller loop of the optimal ate pairing; q in G2 (twist, affine), pt in G1 (affine ints)."""
    if q is None or pt is None:
        return F12_ONE
    f = F12_ONE
    r = q
    for i in range(ATE.bit_length() - 2, -1, -1):
        f = f12_mul(f12_mul(f, f), _line(r, r, pt))
        r = g2_dbl(r)
        if (ATE >> i) & 1:
            f = f12_mul(f, _line(r, q, pt))
            r = g2_add(r, q)
    q1 = (f2_mul(f2_conj(q[0]), _G12), f2_mul(f2_conj(q[1]), _G13))
    nq2 = (f2_mul(q[0], _G22), f2_neg(f2_mul(q[1], _G23)))
    f = f12_mul(f, _line(r, q1, pt))
    r = g2_add(r, q1)
    f = f12_mul(f, _line(r, nq2, pt))
    return f


def final_exp(f):
    return f12_pow(f, (P ** 12 - 1) // R)


# ---- G1 affine on y^2 = x^3 + 3 ---------------------------------------------------------
def g1_add(p, q):
    if p is None: return q
    if q is None: return p
    if p[0] == q[0]:
        if (p[1] + q[1]) % P == 0:
            return None
        m = 3 * p[0] * p[0] * pow(2 * p[1], -1, P) % P
    else:
        m = (q[1] - p[1]) * pow(q[0] - p[0], -1, P) % P
    x = (m * m - p[0] - q[0]) % P
    return (x, (m * (p[0] - x) - p[1]) % P)


def g1_mul(p, k):
    r = None
    while k:
        if k & 1:
            r = g1_add(r, p)
        p = g1_add(p, p)
        k >>= 1
    return r


def g1_neg(p):
    return None if p is None else (p[0], (-p[1]) % P)


def g1_on_curve(p):
    return p is None or (p[1] * p[1] - p[0] ** 3 - 3) % P == 0


def g2_on_curve(q):
    if q is None:
        return True
    b = f2_muls(f2_inv(XI), 3)
    return f2_sub(f2_mul(q[1], q[1]), f2_add(f2_mul(f2_mul(q[0], q[0]), q[0]), b)) == (0, 0)


# ---- Groth16 verification ---------------------------------------------------------------
def _g1(j):
    p = (int(j[0]), int(j[1]))
    return None if p == (0, 0) or (len(j) > 2 and int(j[2]) == 0) else p


def _g2(j):
    q = ((int(j[0][0]), int(j[0][1])), (int(j[1][0]), int(j[1][1])))
    return None if q == ((0, 0), (0, 0)) else q


def groth16_verify(vk, public, proof):
    """vk, public, proof: parsed JSON (dict / list / dict) in snarkjs layout. Returns bool."""
    if isinstance(vk, (str, bytes)): vk = json.loads(vk)
    if isinstance(public, (str, bytes)): public = json.loads(public)
    if isinstance(proof, (str, bytes)): proof = json.loads(proof)
    a, b, c = _g1(proof["pi_a"]), _g2(proof["pi_b"]), _g1(proof["pi_c"])
    if not (g1_on_curve(a) and g1_on_curve(c) and g2_on_curve(b)):
        return False
    ic = [_g1(x) for x in vk["IC"]]
    if len(public) + 1 != len(ic):
        return False
    vkx = ic[0]
    for v, pt in zip(public, ic[1:]):
        vkx = g1_add(vkx, g1_mul(pt, int(v) % R))
    f = miller(b, a)
    f = f12_mul(f, miller(_g2(vk["vk_beta_2"]), g1_neg(_g1(vk["vk_alpha_1"]))))
    f = f12_mul(f, miller(_g2(vk["vk_gamma_2"]), g1_neg(vkx)))
    f = f12_mul(f, miller(_g2(vk["vk_delta_2"]), g1_neg(c)))
    return final_exp(f) == F12_ONE


def derive_challenge(commit):
    """derive_challenge (src/ultra_groth.cpp:33-58): keccak256(x_BE32 || y_BE32) of the affine round commitment, read as a
    big-endian integer, mod r. commit = (x, y) ints."""
    from . import keccak256
    d = keccak256(commit[0].to_bytes(32, "big") + commit[1].to_bytes(32, "big"))
    return int.from_bytes(d, "big") % R


def ultra_groth_verify(vk, public, proof):
    """UltraGroth::Verifier::verify (src/ultra_groth.cpp:581-648):
        vkX = IC[0] + sum_i public[i] * IC[i+1] + derive_challenge(pi_r) * IC_rand
        e(A, B) e(-alpha1, beta2) e(-vkX, gamma2) e(-pi_f, delta_c2_2) e(-pi_r, delta_c1_2) == 1"""
    if isinstance(vk, (str, bytes)): vk = json.loads(vk)
    if isinstance(public, (str, bytes)): public = json.loads(public)
    if isinstance(proof, (str, bytes)): proof = json.loads(proof)
    a, b = _g1(proof["pi_a"]), _g2(proof["pi_b"])
    fin, rnd = _g1(proof["pi_f"]), _g1(proof["pi_r"])
    if not (g1_on_curve(a) and g1_on_curve(fin) and g1_on_curve(rnd) and g2_on_curve(b)):
        return False
    ic = [_g1(x) for x in vk["IC"]]
    if len(public) + 1 != len(ic):
        return False
    vkx = ic[0]
    for v, pt in zip(public, ic[1:]):
        vkx = g1_add(vkx, g1_mul(pt, int(v) % R))
    vkx = g1_add(vkx, g1_mul(_g1(vk["IC_rand"]), derive_challenge(rnd)))
    f = F12_ONE
    for g1p, g2p in ((a, b), (g1_neg(_g1(vk["vk_alpha_1"])), _g2(vk["vk_beta_2"])), (g1_neg(vkx), _g2(vk["vk_gamma_2"])),
                     (g1_neg(fin), _g2(vk["vk_delta_c2_2"])), (g1_neg(rnd), _g2(vk["vk_delta_c1_2"]))):
        if g1p is None or g2p is None:
            continue
        f = f12_mul(f, miller(g2p, g1p))
    return final_exp(f) == F12_ONE
