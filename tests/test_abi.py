"""The C-ABI library loads without a GPU, exports every symbol include/*.h declares, and its host-only entry points
behave like the reference's (sizes, error codes, error strings). No compute calls here."""
import ctypes as C
import os
import re

import pytest

import ultragroth_amd as ug
from ultragroth_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return ug.load()


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return set(re.findall(r"\b((?:ug|groth16|ultra_groth)_[a-z0-9_]+)\s*\(", txt))


def test_every_declared_symbol_is_exported(lib):
    inner, outer, verifier = _declared("ultragroth_hip.h"), _declared("prover.h"), _declared("verifier.h")
    assert inner == set(_lib.INNER_SYMBOLS)
    assert outer == set(_lib.OUTER_SYMBOLS)
    assert verifier == set(_lib.VERIFIER_SYMBOLS)             # the reference's src/verifier.h:22-38
    for name in inner | outer | verifier:
        assert hasattr(lib, name), name


def test_reference_symbol_set_is_complete(lib):
    """the 18 entry points of the reference's src/prover.h"""
    ref = ["groth16_public_size_for_zkey_buf", "ultra_groth_public_size_for_zkey_buf", "groth16_public_size_for_zkey_file",
           "ultra_groth_public_size_for_zkey_file", "groth16_proof_size", "ultra_groth_proof_size", "groth16_prover_create",
           "ultra_groth_prover_create", "groth16_prover_create_zkey_file", "ultra_groth_prover_create_zkey_file",
           "groth16_prover_prove", "ultra_groth_prover_prove", "groth16_prover_destroy", "ultra_groth_prover_destroy",
           "groth16_prover", "ultra_groth_prover", "groth16_prover_zkey_file", "ultra_groth_prover_zkey_file"]
    for name in ref:
        assert hasattr(lib, name)


def test_sizes(lib, zkey):
    assert ug.groth16_proof_size() == 810                 # src/prover.cpp:55-59
    assert ug.ultra_groth_proof_size() == 1400            # :61-65
    assert ug.groth16_public_size_for_zkey_buf(zkey) == 1 * 82 + 4       # :67-71
    v = C.c_ulonglong()
    err = C.create_string_buffer(256)
    path = os.path.join(ROOT, "tests", "golden", "circuit_final.zkey").encode()
    assert lib.groth16_public_size_for_zkey_file(path, C.byref(v), err, 255) == 0 and v.value == 86


def test_error_strings_match_the_reference(lib, zkey, wtns):
    with pytest.raises(ug.ProverError) as e:
        ug.groth16_public_size_for_zkey_buf(b"zkey")
    assert e.value.code == ug.PROVER_ERROR and e.value.message == "File is too short."           # binfile_utils.cpp:37
    with pytest.raises(ug.ProverError) as e:
        ug.groth16_public_size_for_zkey_buf(wtns)
    assert e.value.message == "Invalid file type. It should be zkey and it is wtns"                # :44
    with pytest.raises(ug.ProverError) as e:
        ug.ultra_groth_public_size_for_zkey_buf(zkey)
    assert e.value.message == "zkey file is not ultragroth"                                        # zkey_utils.cpp:129-131
    bad = bytearray(zkey)
    bad[4] = 9
    with pytest.raises(ug.ProverError) as e:
        ug.groth16_public_size_for_zkey_buf(bytes(bad))
    assert e.value.message == "Invalid version. It should be <=1 and it is 9"                      # :49
    with pytest.raises(ug.ProverError) as e:
        ug.groth16_public_size_for_zkey_buf(zkey[:5000])
    assert e.value.message.startswith("Section #")                                                 # :71-75
    # null-argument checks come before any device work (src/prover.cpp:382-388,514-536)
    err = C.create_string_buffer(256)
    assert lib.groth16_prover_create(None, zkey, len(zkey), err, 255) == ug.PROVER_ERROR
    assert err.value == b"Null prover object"
    h = C.c_void_p()
    assert lib.groth16_prover_create(C.byref(h), None, 0, err, 255) == ug.PROVER_ERROR
    assert err.value == b"Null zkey buffer"
    assert lib.groth16_prover_prove(None, wtns, len(wtns), None, None, None, None, err, 255) == ug.PROVER_ERROR
    assert err.value == b"Null prover object"
    msg = C.create_string_buffer(b"\xff" * 8, 8)          # strncpy semantics: at most maxsize bytes, maybe unterminated
    assert lib.groth16_prover_create(None, zkey, len(zkey), msg, 4) == ug.PROVER_ERROR
    assert msg.raw[:4] == b"Null" and msg.raw[4:] == b"\xff" * 4


def test_section_size_cannot_wrap(lib, zkey):
    """a section size near 2^64 must not wrap the cursor back into the buffer (the reference adds before it checks,
    src/binfile_utils.cpp:60-66; here the check comes first)"""
    import struct
    for huge in ((1 << 64) - 1, (1 << 64) - 12, (1 << 64) - 24 - 10, (1 << 63) + 1):
        bad = b"zkey" + struct.pack("<II", 1, 2) + struct.pack("<IQ", 1, 4) + struct.pack("<I", 1) + struct.pack("<IQ", 2, huge)
        bad += bytes(64)
        with pytest.raises(ug.ProverError) as e:
            ug.groth16_public_size_for_zkey_buf(bad)
        assert e.value.message.startswith("Section #1 is invalid")
    # a size that ends exactly at the end of the buffer is fine
    ok = b"zkey" + struct.pack("<II", 1, 2) + struct.pack("<IQ", 1, 4) + struct.pack("<I", 1) + struct.pack("<IQ", 2, 64) + bytes(64)
    assert ug.groth16_public_size_for_zkey_buf(ok) == 4            # (an all-zero header: nPublic = 0)


def test_blinding_hook_is_gated():
    """ug_test_set_blinding changes nothing unless the process was started with ULTRAGROTH_TEST_HOOKS=1"""
    import subprocess
    import sys
    code = ("import ctypes, sys; sys.path.insert(0, %r); import ultragroth_amd as ug; L = ug.load(); "
            "print(L.ug_test_set_blinding(b'x' * 62, 62), L.ug_test_set_blinding(None, 0))" % ROOT)
    env = dict(os.environ)
    env.pop("ULTRAGROTH_TEST_HOOKS", None)
    assert subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout.split() == ["1", "0"]
    env["ULTRAGROTH_TEST_HOOKS"] = "1"
    assert subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout.split() == ["0", "0"]


def test_product_library_holds_no_measurement_code():
    """the A/B and measurement switches of finished experiments (one of them, UG_GROUP_FOLD_LOG, gives wrong sums) and the
    library sort exist only in the -DUG_MEASURE build (include/ultragroth_hip.h, csrc/dev_common.hpp: measure_env): the
    product library contains neither their names nor a hipcub / rocprim symbol"""
    import subprocess
    blob = open(_lib.LIB_PATH, "rb").read()
    for name in (b"UG_GROUP_FOLD_LOG", b"UG_GROUP_ROTATE", b"UG_SORT_IPT", b"UG_SORT_SPL", b"UG_SORT_LBW", b"UG_SORT_DROP",
                 b"UG_NTT_BATCH", b"UG_MATVEC_TILED", b"UG_SORT\0"):
        assert name not in blob, name
    syms = subprocess.run(["nm", "-C", _lib.LIB_PATH], capture_output=True, text=True).stdout.lower()
    assert "hipcub" not in syms and "rocprim" not in syms
    # ... while the knobs the header's table lists as part of the product are there
    for name in (b"UG_TABLE_C", b"UG_SEG_LANES_LOG", b"ULTRAGROTH_DEVICES", b"ULTRAGROTH_TEST_HOOKS"):
        assert name in blob, name


def test_no_silent_cpu_fallback(lib, zkey):
    """without a GPU the compute entry points fail loudly"""
    if ug.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(ug.DeviceError):
        ug.Device(0)
    with pytest.raises(ug.ProverError) as e:
        ug.Groth16Prover(zkey)
    assert e.value.code == ug.PROVER_ERROR and "HIP" in e.value.message


def test_partials_add_is_host_only_and_exact(lib, zkey):
    """ug_groth16_partials_add (the N>1 exchange step) against the oracle's group law"""
    import oracle as O
    off, _ = O.section(zkey, "zkey", 5)
    off2, _ = O.section(zkey, "zkey", 7)
    g1 = [zkey[off + 64 * i: off + 64 * i + 64] for i in range(2, 12)]
    g2 = [zkey[off2 + 128 * i: off2 + 128 * i + 128] for i in range(2, 4)]
    a = g1[0] + g1[1] + g2[0] + g1[2] + g1[3]
    b = g1[4] + bytes(64) + g2[1] + g1[2] + g1[5]                  # B1 partial at infinity; C partials equal (doubling)
    s = ug.ShardedGroth16Prover.add_partials(a, b)
    assert s[0:64] == O.g1_add(g1[0], g1[4])
    assert s[64:128] == g1[1]
    assert s[128:256] == O.g2_add(g2[0], g2[1])
    assert s[256:320] == O.g1_mul(g1[2], 2)
    assert s[320:384] == O.g1_add(g1[3], g1[5])


def test_headers_are_plain_c(tmp_path):
    """include/*.h are what a cgo / JNI / bindgen user includes: they must compile as C99 on their own, and a C program
    must link against the library with nothing but them"""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    inc = os.path.join(ROOT, "include")
    for h in ("prover.h", "verifier.h", "ultragroth_hip.h"):
        r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", inc, "-x", "c", "-"],
                           input='#include "%s"\n' % h, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    src = tmp_path / "link.c"
    src.write_text('#include <stdio.h>\n#include "prover.h"\n#include "verifier.h"\n#include "ultragroth_hip.h"\n'
                   'int main(void) {\n  unsigned long long n = 0; char err[64] = {0};\n  groth16_proof_size(&n);\n'
                   '  int rc = groth16_verify("{", "[]", "{}", err, 63);\n'
                   '  printf("%llu %d %s %d\\n", n, rc, err, ug_msm_table_window(1u << 24));\n  return 0;\n}\n')
    exe = tmp_path / "link"
    libdir = os.path.dirname(_lib.LIB_PATH)
    r = subprocess.run(["gcc", "-std=c99", "-I", inc, str(src), "-o", str(exe), "-L", libdir, "-lultragroth_hip", "-Wl,-rpath," + libdir],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip() == "810 2 invalid proof data 22"


def test_sort_plan_covers_the_key_bits():
    """csrc/sort.hip: ceil(bits / 8) passes, low bits first, digits of at most 8 bits that tile the key exactly, the narrow
    ones first (the fused first pass has the smallest tiles)"""
    import ctypes as C
    from ultragroth_amd import _lib
    L = _lib.load()
    for bits in range(1, 33):
        shift, width = (C.c_int * 4)(), (C.c_int * 4)()
        passes = L.ug_sort_plan(bits, shift, width)
        assert passes == (bits + 7) // 8
        assert [shift[p] for p in range(passes)] == [sum(width[q] for q in range(p)) for p in range(passes)]
        assert sum(width[p] for p in range(passes)) == bits and all(1 <= width[p] <= 8 for p in range(passes))
        assert [width[p] for p in range(passes)] == sorted(width[p] for p in range(passes))
    assert L.ug_sort_plan(0, (C.c_int * 4)(), (C.c_int * 4)()) == -1 and L.ug_sort_plan(33, (C.c_int * 4)(), (C.c_int * 4)()) == -1


def test_balanced_witness_ranges_tile_the_witness():
    """ug_groth16_balanced_witness_range (what ULTRAGROTH_DEVICES and bench.py split the witness-indexed sections by): the ranks'
    ranges tile [0, nVars) in order for every rank count, ranks that run an NTT chain (0..2) get fewer points than the others,
    and the C ranges that follow from them stay inside the C section"""
    import ultragroth_amd as ug
    for n_vars in (1, 2, 7, 1003, (1 << 20) - 1, (1 << 26) - 1):
        for world in (1, 2, 3, 4, 5, 8, 16):
            prev = 0
            sizes = []
            for k in range(world):
                lo, hi = ug.ShardedGroth16Prover.balanced_witness_range(n_vars, k, world)
                assert lo == prev and hi >= lo
                prev = hi
                sizes.append(hi - lo)
                if n_vars >= 2:
                    w, c, h = ug.ShardedGroth16Prover.shard_ranges(n_vars, 1, 1 << 10, k, world, (lo, hi))
                    assert w == (lo, hi) and 0 <= c[0] <= c[1] <= n_vars - 2
            assert prev == n_vars
            if world > 3 and n_vars > 1000:
                assert max(sizes[:3]) < min(sizes[3:])
    with pytest.raises(ug.ProverError):
        ug.ShardedGroth16Prover.balanced_witness_range(100, 3, 3)


def test_shard_layouts_tile_everything(monkeypatch):
    """ug_groth16_shard_layout (host arithmetic only): for every rank count and every split into P base-point ranges x B bucket-class
    ranks the layouts tile the witness (per group), the residues [0, Q) and the special-bucket scalars of every group, and h; the
    default is the base-point form (P = world, no classes: it measured faster); ULTRAGROTH_SHARD=PxB asks for
    classes; chain ranks get smaller shares, and from five ranks on no part of h; no rank gets more than 31 residues (a result
    block holds 63 points); the ultra ranges tile their four sets"""
    import ultragroth_amd as ug
    S = ug.ShardedGroth16Prover
    monkeypatch.delenv("ULTRAGROTH_SHARD", raising=False)
    for log_n in (14, 20, 24, 26):
        n_vars, dom = (1 << log_n) - 1, 1 << log_n
        for world in (1, 2, 3, 4, 5, 6, 8, 16):
            base = [S.shard_layout(n_vars, 1, dom, k, world) for k in range(world)]
            assert all(L.q_log == 0 for L in base)                               # the library's own choice: base-point ranges
            assert [L.witness for L in base] == [S.balanced_witness_range(n_vars, k, world) for k in range(world)]
            for P in [p for p in (1, 2, 4, 8) if world % p == 0 and p < world]:
                lay = [S.shard_layout(n_vars, 1, dom, k, world, P) for k in range(world)]
                B = world // P
                assert lay[0].h[0] == 0 and lay[-1].h[1] == dom and all(a.h[1] == b.h[0] for a, b in zip(lay, lay[1:]))
                assert lay[0].witness[0] == 0 and lay[-1].witness[1] == n_vars
                for g in range(P):
                    grp = lay[g * B:(g + 1) * B]
                    Q = 1 << grp[0].q_log
                    assert grp[0].q_log > 0 and all(L.witness == grp[0].witness and L.q_log == grp[0].q_log for L in grp)
                    assert grp[0].first_residue == 0 and grp[-1].first_residue + grp[-1].residues == Q
                    assert all(a.first_residue + a.residues == b.first_residue and a.special[1] == b.special[0] for a, b in zip(grp, grp[1:]))
                    assert (grp[0].special[0], grp[-1].special[1]) == grp[0].witness
                    assert all(1 <= L.residues <= 31 for L in grp)
                    if g + 1 < P:
                        assert grp[0].witness[1] == lay[(g + 1) * B].witness[0]
                for k, L in enumerate(lay):
                    assert L.chains == [c for c in range(3) if c % world == k]
                    assert L.c[0] == max(L.witness[0] - 2, 0) and L.c[1] == max(L.witness[1] - 2, 0)
                if world >= 5:
                    assert all(L.h[0] == L.h[1] for L in lay[:3]) and all(L.h[1] > L.h[0] for L in lay[3:])
                    if P == 1 and world == 8:                                     # eight ranks: a chain outweighs a fifth of the H product, so the chain ranks get fewer residues
                        assert max(L.residues for L in lay[:3]) <= min(L.residues for L in lay[3:])
    monkeypatch.setenv("ULTRAGROTH_SHARD", "2x4")
    assert [S.shard_layout((1 << 24) - 1, 1, 1 << 24, k, 8).q_log for k in range(8)] == [6] * 8
    assert S.shard_layout((1 << 24) - 1, 1, 1 << 24, 0, 4).q_log == 0                       # 2 x 4 does not fit four ranks: base-point form
    monkeypatch.setenv("ULTRAGROTH_SHARD", "auto")
    assert S.shard_layout((1 << 24) - 1, 1, 1 << 24, 0, 8).witness == (0, (1 << 24) - 1)      # one group: the tables of 2^24 points fit
    assert S.shard_layout((1 << 26) - 1, 1, 1 << 26, 7, 8).witness[0] > 0                      # 2^26: two groups
    assert S.shard_layout((1 << 24) - 1, 1, 1 << 24, 0, 2).q_log == 0                          # fewer than four ranks: base-point form
    with pytest.raises(ug.ProverError):
        S.shard_layout(100, 1, 128, 4, 4)
    monkeypatch.delenv("ULTRAGROTH_SHARD", raising=False)
    with pytest.raises(ug.ProverError):                # a caller's point_ranges that does not divide the ranks is refused, not replaced
        S.shard_layout((1 << 24) - 1, 1, 1 << 24, 0, 8, 3)
    with pytest.raises(ug.ProverError):                # more class ranks per range than residues (Q is capped at 2^7)
        S.shard_layout((1 << 24) - 1, 1, 1 << 24, 0, 256, 1)
    U = ug.ShardedUltraGrothProver
    for world in (1, 3, 8):
        rs = [U.shard_ranges(1000, 1024, 249, 748, k, world) for k in range(world)]
        for part, end in zip(range(4), (1000, 249, 748, 1024)):
            assert rs[0][part][0] == 0 and rs[-1][part][1] == end and all(a[part][1] == b[part][0] for a, b in zip(rs, rs[1:]))


def test_sliced_creation_checks_its_buffers_before_any_device_work(lib):
    """ug_groth16_prover_create_sharded_slices / _layout and ug_ultra_groth_prover_create_sharded_slices refuse a slice that is shorter
    than the rank's range (or missing) with the reason -- host checks that come before the first HIP call, so they hold without a
    GPU; with buffers of the right size the call gets as far as the device (and fails there on a box without one)"""
    import struct
    from ultragroth_amd import synth
    r_le, q_le = synth.R_MOD.to_bytes(32, "little"), synth.Q_MOD.to_bytes(32, "little")
    n_vars, dom = 1000, 1024
    hdr = struct.pack("<I", 32) + q_le + struct.pack("<I", 32) + r_le + struct.pack("<III", n_vars, 1, dom) + bytes(64 + 64 + 128 + 128 + 64 + 128)
    S = ug.ShardedGroth16Prover
    (w0, w1), (c0, c1), (h0, h1) = S.shard_ranges(n_vars, 1, dom, 1, 4)
    good = [bytes((w1 - w0) * 64), bytes((w1 - w0) * 64), bytes((w1 - w0) * 128), bytes((c1 - c0) * 64), bytes((h1 - h0) * 64)]
    for k, name in enumerate(("points_a", "points_b1", "points_b2", "points_c", "points_h")):
        short = list(good)
        short[k] = short[k][:-1]
        with pytest.raises(ug.ProverError, match=name + " slice is shorter"):
            S.from_slices(hdr, None, 0, short, 0, 1, 4, public_size=86)
    lay = S.shard_layout(n_vars, 1, dom, 1, 4)
    with pytest.raises(ug.ProverError, match="points_h slice is shorter"):
        S.from_slices(hdr, None, 0, good[:4] + [b""], 0, 1, 4, public_size=86, layout=lay)
    bad = ug.ShardLayout(lay.raw[:6] + [9, 0, 1] + lay.raw[9:])
    with pytest.raises(ug.ProverError, match="invalid bucket classes"):
        S.from_slices(hdr, None, 0, good, 0, 1, 4, public_size=86, layout=bad)
    if ug.device_count() < 1:
        with pytest.raises(ug.ProverError, match="HIP"):
            S.from_slices(hdr, None, 0, good, 0, 1, 4, public_size=86)
    # UltraGroth: eight buffers
    n1, n2 = 249, 748
    uh = struct.pack("<I", 32) + q_le + struct.pack("<I", 32) + r_le + struct.pack("<IIIIII", n_vars, 2, dom, n1, n2, 2) + bytes(64 + 64 + 128 + 128 + 64 + 128 + 64 + 128)
    U = ug.ShardedUltraGrothProver
    (w0, w1), (a0, a1), (f0, f1), (h0, h1) = U.shard_ranges(n_vars, dom, n1, n2, 2, 3)
    ugood = [bytes((w1 - w0) * 64), bytes((w1 - w0) * 64), bytes((w1 - w0) * 128), bytes((a1 - a0) * 64), bytes((f1 - f0) * 64), bytes((h1 - h0) * 64),
             bytes((a1 - a0) * 4), bytes((f1 - f0) * 4)]
    for k, name in enumerate(("points_a", "points_b1", "points_b2", "points_round_c", "points_final_c", "points_h", "round_indexes", "final_round_indexes")):
        short = list(ugood)
        short[k] = short[k][:-1]
        with pytest.raises(ug.ProverError, match=name + " slice is shorter"):
            U.from_slices(uh, None, 0, short, 0, 2, 3, public_size=86)
    with pytest.raises(ug.ProverError, match="Invalid section size"):
        U.from_slices(uh[:40], None, 0, ugood, 0, 2, 3, public_size=86)
    if ug.device_count() < 1:
        with pytest.raises(ug.ProverError, match="HIP"):
            U.from_slices(uh, None, 0, ugood, 0, 2, 3, public_size=86)


@pytest.mark.parametrize("order", ["library_first", "torch_first"])
def test_one_hip_runtime_whatever_the_import_order(order):
    """torch bundles a HIP runtime of its own: the loader must leave ONE copy in the process whichever is imported first
    (round-2 review: library first used to give two runtimes and a torch without devices). Fresh interpreter per order."""
    import subprocess
    import sys
    first, second = ("import ultragroth_amd as ug; ug._lib.load()", "import torch") if order == "library_first" else \
                    ("import torch", "import ultragroth_amd as ug; ug._lib.load()")
    code = "%s\n%s\nimport ultragroth_amd._lib as L\nprint('RUNTIMES', len(L.hip_runtimes_loaded()), L.hip_runtimes_loaded())" % (first, second)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "RUNTIMES 1 " in out.stdout, out.stdout


@pytest.mark.gpu
def test_library_first_then_torch_sees_the_device():
    """the same on the GPU box: the library is loaded and used first, torch afterwards still finds the card and both work"""
    import subprocess
    import sys
    code = ("import ultragroth_amd as ug\n"
            "d = ug.Device(0)\n"
            "import torch\n"
            "assert torch.cuda.is_available() and torch.cuda.device_count() >= 1\n"
            "x = torch.arange(8, device='cuda').sum().item()\n"
            "assert x == 28\n"
            "import ultragroth_amd._lib as L\n"
            "assert len(L.hip_runtimes_loaded()) == 1, L.hip_runtimes_loaded()\n"
            "d.close()\nprint('OK')\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "OK" in out.stdout, (out.stdout + out.stderr)[-2000:]
