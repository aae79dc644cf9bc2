"""ultragroth_amd -- MI355X-native drop-in for the prover hot path of rarimo/ultragroth.

Host-side mirror (Python) of the C interfaces in ``include/``:

* :class:`Groth16Prover`, :class:`UltraGrothProver`, :func:`groth16_prover` ... mirror the reference's
  ``extern "C"`` prover API (src/prover.h) -- same names, argument meaning and error behaviour,
  with the status codes raised as :class:`ProverError` carrying ``code`` and the reference's message.
* :class:`Device` exposes the inner ABI (``include/ultragroth_hip.h``): MSM, NTT, H polynomial and
  field ops, in the reference's byte formats.

All compute goes through ``libultragroth_hip.so`` (hand-written HIP for gfx950). There is no CPU
fallback: without the library or without a GPU every call raises.
"""
import ctypes as C

from . import _lib
from ._lib import build, load, LIB_PATH

PROVER_OK = 0
PROVER_ERROR = 1
PROVER_ERROR_SHORT_BUFFER = 2
PROVER_INVALID_WITNESS_LENGTH = 3

FR, FQ = 0, 1
OP_MUL, OP_ADD, OP_SUB, OP_SQR = 0, 1, 2, 3
GROTH16_PARTIALS_SIZE = 384


class ProverError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(message)
        self.code = code
        self.message = message


class DeviceError(RuntimeError):
    pass


def _check(rc):
    if rc != 0:
        raise DeviceError(load().ug_last_error().decode(errors="replace"))


def device_count():
    return load().ug_device_count()


def set_test_blinding(data):
    """Queue bytes for the prover's blinding draws (31 bytes each); b'' restores OS entropy."""
    data = bytes(data)
    if load().ug_test_set_blinding(data if data else None, len(data)) != PROVER_OK:
        raise ProverError(PROVER_ERROR, "test hooks are off: start the process with ULTRAGROTH_TEST_HOOKS=1")


FAULT_HPOLY_RUN, FAULT_SCHEDULE_BUILD = 1, 2


def inject_fault(site, after=1):
    """ug_test_inject_fault: the `after`-th next pass through fault point `site` fails (test processes only)"""
    if load().ug_test_inject_fault(site, after) != 0:
        raise ProverError(PROVER_ERROR, "test hooks are off: start the process with ULTRAGROTH_TEST_HOOKS=1")


def _buf(b):
    return (C.c_char * len(b)).from_buffer_copy(b) if not isinstance(b, C.Array) else b


# ---------------------------------------------------------------------------------------------------
# outer API mirror (src/prover.h)

def groth16_proof_size():
    v = C.c_ulonglong()
    load().groth16_proof_size(C.byref(v))
    return v.value


def ultra_groth_proof_size():
    v = C.c_ulonglong()
    load().ultra_groth_proof_size(C.byref(v))
    return v.value


def _public_size(fn, zkey):
    v = C.c_ulonglong()
    err = C.create_string_buffer(1024)
    rc = fn(zkey, len(zkey), C.byref(v), err, len(err) - 1)
    if rc != PROVER_OK:
        raise ProverError(rc, err.value.decode(errors="replace"))
    return v.value


def groth16_public_size_for_zkey_buf(zkey):
    return _public_size(load().groth16_public_size_for_zkey_buf, zkey)


def ultra_groth_public_size_for_zkey_buf(zkey):
    return _public_size(load().ultra_groth_public_size_for_zkey_buf, zkey)


# ---------------------------------------------------------------------------------------------------
# verifier mirror (src/verifier.h)
VERIFIER_VALID_PROOF, VERIFIER_INVALID_PROOF, VERIFIER_ERROR = 0, 1, 2


class VerifierError(RuntimeError):
    pass


def _verify(fn, proof, inputs, verification_key):
    import json
    enc = lambda v: v if isinstance(v, bytes) else (v if isinstance(v, str) else json.dumps(v)).encode()
    err = C.create_string_buffer(256)
    rc = fn(enc(proof), enc(inputs), enc(verification_key), err, 255)
    if rc == VERIFIER_ERROR:
        raise VerifierError(err.value.decode(errors="replace"))
    return rc == VERIFIER_VALID_PROOF


def groth16_verify(proof, inputs, verification_key):
    """groth16_verify (src/verifier.h:22-29): JSON texts or parsed objects; True / False, VerifierError on bad data"""
    return _verify(load().groth16_verify, proof, inputs, verification_key)


def ultra_groth_verify(proof, inputs, verification_key):
    return _verify(load().ultra_groth_verify, proof, inputs, verification_key)


class _ProverBase:
    _create = _prove = _destroy = _public_size_fn = None
    _proof_size = staticmethod(groth16_proof_size)

    def __init__(self, zkey):
        L = load()
        self._h = C.c_void_p()
        err = C.create_string_buffer(1024)
        rc = getattr(L, self._create)(C.byref(self._h), zkey, len(zkey), err, len(err) - 1)
        if rc != PROVER_OK:
            self._h = None
            raise ProverError(rc, err.value.decode(errors="replace"))
        self._public_size = _public_size(getattr(L, self._public_size_fn), zkey)

    def prove(self, wtns, proof_size=None, public_size=None):
        """Returns (proof_json, public_json) -- the buffers up to their first NUL, as the CLI writes them."""
        L = load()
        psz = C.c_ulonglong(self._proof_size() if proof_size is None else proof_size)
        qsz = C.c_ulonglong(self._public_size if public_size is None else public_size)
        proof = C.create_string_buffer(max(psz.value, 1))
        pub = C.create_string_buffer(max(qsz.value, 1))
        err = C.create_string_buffer(1024)
        rc = getattr(L, self._prove)(self._h, wtns, len(wtns), proof, C.byref(psz), pub, C.byref(qsz), err, len(err) - 1)
        if rc != PROVER_OK:
            raise ProverError(rc, err.value.decode(errors="replace"))
        return proof.raw.split(b"\0", 1)[0].decode(), pub.raw.split(b"\0", 1)[0].decode()

    # the phases of a proof on a witness that stays resident in HBM (include/prover.h: ug_groth16_prover_load_witness / _run /
    # _finish; bench.py times run + finish): available on every prover object
    def load_witness(self, wtns):
        err = C.create_string_buffer(1024)
        rc = load().ug_groth16_prover_load_witness(self._h, wtns, len(wtns), err, len(err) - 1)
        if rc != PROVER_OK:
            raise ProverError(rc, err.value.decode(errors="replace"))

    def prove_resident(self):
        """ug_groth16_prover_prove_resident: one whole proof of the witness load_witness left in HBM"""
        psz = C.c_ulonglong(self._proof_size())
        qsz = C.c_ulonglong(self._public_size)
        proof = C.create_string_buffer(psz.value)
        pub = C.create_string_buffer(max(qsz.value, 1))
        err = C.create_string_buffer(1024)
        rc = load().ug_groth16_prover_prove_resident(self._h, proof, C.byref(psz), pub, C.byref(qsz), err, len(err) - 1)
        if rc != PROVER_OK:
            raise ProverError(rc, err.value.decode(errors="replace"))
        return proof.raw.split(b"\0", 1)[0].decode(), pub.raw.split(b"\0", 1)[0].decode()

    def run(self):
        out = C.create_string_buffer(GROTH16_PARTIALS_SIZE)
        err = C.create_string_buffer(1024)
        rc = load().ug_groth16_prover_run(self._h, out, err, len(err) - 1)
        if rc != PROVER_OK:
            raise ProverError(rc, err.value.decode(errors="replace"))
        return out.raw

    def finish(self, partials_sum):
        psz = C.c_ulonglong(self._proof_size())
        qsz = C.c_ulonglong(self._public_size)
        proof = C.create_string_buffer(psz.value)
        pub = C.create_string_buffer(max(qsz.value, 1))
        err = C.create_string_buffer(1024)
        rc = load().ug_groth16_prover_finish(self._h, bytes(partials_sum), proof, C.byref(psz), pub, C.byref(qsz), err, len(err) - 1)
        if rc != PROVER_OK:
            raise ProverError(rc, err.value.decode(errors="replace"))
        return proof.raw.split(b"\0", 1)[0].decode(), pub.raw.split(b"\0", 1)[0].decode()

    def last_timings(self):
        """(msm_ms, fft_ms, total_ms) of the last prove: device time of the MSM and H-polynomial parts, host wall time."""
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        load().ug_prover_last_timings(self._h, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    def kernel_stats(self, g2=False, reset=False, which=None):
        """(avg launch ms, launches, units) since creation / last reset of the G1 (which 0) / G2 (1) bucket-accumulation
        kernel or the NTT pass kernel (2)."""
        a, l, e = C.c_double(), C.c_ulonglong(), C.c_ulonglong()
        w = (1 if g2 else 0) if which is None else which
        load().ug_prover_kernel_stats(self._h, w, C.byref(a), C.byref(l), C.byref(e), 1 if reset else 0)
        return a.value, l.value, e.value

    def last_upload_ms(self):
        """host wall time the last prove / load_witness spent bringing the witness into HBM"""
        a = C.c_double()
        load().ug_prover_last_upload_ms(self._h, C.byref(a))
        return a.value

    def tables_ready(self, wait=False):
        """ug_prover_tables_ready: True once the fixed-base window tables (built in the background after create) are in use;
        wait=True blocks until they are"""
        rc = load().ug_prover_tables_ready(self._h, 1 if wait else 0)
        if rc < 0:
            raise ProverError(PROVER_ERROR, "ug_prover_tables_ready failed")
        return rc == 1

    def close(self):
        if getattr(self, "_h", None):
            getattr(load(), self._destroy)(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class Groth16Prover(_ProverBase):
    """groth16_prover_create / _prove / _destroy (src/prover.h:80-87,127-138,156-159)."""
    _create, _prove, _destroy = "groth16_prover_create", "groth16_prover_prove", "groth16_prover_destroy"
    _public_size_fn = "groth16_public_size_for_zkey_buf"


class UltraGrothProver(_ProverBase):
    """ultra_groth_prover_create / _prove / _destroy (src/prover.h:89-96,140-151,161-164)."""
    _create, _prove, _destroy = "ultra_groth_prover_create", "ultra_groth_prover_prove", "ultra_groth_prover_destroy"
    _public_size_fn = "ultra_groth_public_size_for_zkey_buf"
    _proof_size = staticmethod(ultra_groth_proof_size)


def _one_shot(fn, public_size_fn, proof_size, zkey, wtns):
    L = load()
    psz = C.c_ulonglong(proof_size)
    qsz = C.c_ulonglong(_public_size(getattr(L, public_size_fn), zkey))
    proof = C.create_string_buffer(max(psz.value, 1))
    pub = C.create_string_buffer(max(qsz.value, 1))
    err = C.create_string_buffer(1024)
    rc = getattr(L, fn)(zkey, len(zkey), wtns, len(wtns), proof, C.byref(psz), pub, C.byref(qsz), err, len(err) - 1)
    if rc != PROVER_OK:
        raise ProverError(rc, err.value.decode(errors="replace"))
    return proof.raw.split(b"\0", 1)[0].decode(), pub.raw.split(b"\0", 1)[0].decode()


def groth16_prover(zkey, wtns):
    """The reference's one-shot entry point groth16_prover (src/prover.h:173-185): create, prove once, destroy -- no window
    tables are built for it (they cannot pay for one proof)."""
    return _one_shot("groth16_prover", "groth16_public_size_for_zkey_buf", groth16_proof_size(), zkey, wtns)


def ultra_groth_prover(zkey, wtns):
    """ultra_groth_prover (src/prover.h:187-199)"""
    return _one_shot("ultra_groth_prover", "ultra_groth_public_size_for_zkey_buf", ultra_groth_proof_size(), zkey, wtns)


class Registry:
    """ug_registry_* (include/prover.h): several resident circuits on one device under an HBM budget -- the GPU form of
    the reference's FullProver map<circuit, Prover> (src/fullprover.cpp:21-63)."""
    NOT_LOADED, RESIDENT, RESIDENT_WITH_TABLES, EVICTED = 0, 1, 2, 3

    def __init__(self, device=0, hbm_budget_bytes=0):
        self._h = C.c_void_p()
        err = C.create_string_buffer(1024)
        rc = load().ug_registry_create(C.byref(self._h), device, hbm_budget_bytes, err, len(err) - 1)
        if rc != PROVER_OK:
            self._h = None
            raise ProverError(rc, err.value.decode(errors="replace"))

    def _call(self, name, *args):
        err = C.create_string_buffer(1024)
        rc = getattr(load(), name)(self._h, *args, err, len(err) - 1)
        if rc != PROVER_OK:
            raise ProverError(rc, err.value.decode(errors="replace"))

    def load(self, circuit, zkey):
        self._call("ug_registry_load", circuit.encode(), zkey, len(zkey))

    def load_file(self, path):
        self._call("ug_registry_load_file", path.encode())

    def prove(self, circuit, wtns, proof_size=1400, public_size=1 << 16):
        psz, qsz = C.c_ulonglong(proof_size), C.c_ulonglong(public_size)
        proof, pub = C.create_string_buffer(psz.value), C.create_string_buffer(qsz.value)
        self._call("ug_registry_prove", circuit.encode(), wtns, len(wtns), proof, C.byref(psz), pub, C.byref(qsz))
        return proof.raw.split(b"\0", 1)[0].decode(), pub.raw.split(b"\0", 1)[0].decode()

    def evict(self, circuit):
        self._call("ug_registry_evict", circuit.encode())

    def info(self, circuit=None):
        """(resident bytes, state, proofs) of a circuit; (bytes in use, resident circuits, proofs) for circuit=None"""
        b, st, n = C.c_ulonglong(), C.c_int(), C.c_ulonglong()
        load().ug_registry_info(self._h, circuit.encode() if circuit else None, C.byref(b), C.byref(st), C.byref(n))
        return b.value, st.value, n.value

    def close(self):
        if getattr(self, "_h", None):
            load().ug_registry_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class ShardLayout:
    """out[12] of ug_groth16_shard_layout (include/prover.h): what one rank of a many-device prover owns"""

    def __init__(self, raw):
        self.raw = list(raw)
        self.ranges = ((raw[0], raw[1]), (raw[2], raw[3]), (raw[4], raw[5]))      # witness, C, H -- as shard_ranges returns them
        self.witness, self.c, self.h = self.ranges
        self.q_log, self.first_residue, self.residues = raw[6], raw[7], raw[8]
        self.special = (raw[9], raw[10])
        self.chains = [k for k in range(3) if (raw[11] >> k) & 1]

    def __repr__(self):
        cls = "classes %d..%d of %d" % (self.first_residue, self.first_residue + self.residues, 1 << self.q_log) if self.q_log else "all buckets"
        return "ShardLayout(witness %s, h %s, %s, chains %s)" % (self.witness, self.h, cls, self.chains)


class ShardedGroth16Prover:
    """One rank of a sharded Groth16 prover (one process per GPU): see include/prover.h."""

    def __init__(self, zkey, device, rank, world, witness_range=None):
        """witness_range = (first, end): this rank's slice of the witness-indexed sections, chosen by the caller
        (ug_groth16_prover_create_sharded_range); None = the even split"""
        L = load()
        self._h = C.c_void_p()
        err = C.create_string_buffer(1024)
        if witness_range is None:
            rc = L.ug_groth16_prover_create_sharded(C.byref(self._h), zkey, len(zkey), device, rank, world, err, len(err) - 1)
        else:
            rc = L.ug_groth16_prover_create_sharded_range(C.byref(self._h), zkey, len(zkey), device, rank, world,
                                                          witness_range[0], witness_range[1], err, len(err) - 1)
        if rc != PROVER_OK:
            self._h = None
            raise ProverError(rc, err.value.decode(errors="replace"))
        self._public_size = groth16_public_size_for_zkey_buf(zkey)

    @classmethod
    def from_slices(cls, header, coefs, n_coefs, slices, device, rank, world, witness_range=None, public_size=None, layout=None):
        """ug_groth16_prover_create_sharded_slices: header = zkey section 2, coefs = section 4 records (None: no chain on
        this rank), slices = (A, B1, B2, C, H) buffers holding this rank's points only (shard_ranges tells which).
        layout (a ShardLayout from shard_layout): ug_groth16_prover_create_sharded_layout -- the rank's part of a many-device
        layout with bucket classes; the slices are those of layout.ranges"""
        self = cls.__new__(cls)
        self._h = C.c_void_p()
        err = C.create_string_buffer(1024)
        wr = (C.c_ulonglong * 2)(*witness_range) if witness_range is not None else None
        sizes = (C.c_ulonglong * 5)(*[len(x) for x in slices])
        if layout is not None:
            rc = load().ug_groth16_prover_create_sharded_layout(C.byref(self._h), header, len(header), coefs, n_coefs, *slices, sizes,
                                                                device, rank, world, (C.c_ulonglong * 12)(*layout.raw), err, len(err) - 1)
        else:
            rc = load().ug_groth16_prover_create_sharded_slices(C.byref(self._h), header, len(header), coefs, n_coefs, *slices, sizes,
                                                                device, rank, world, wr, err, len(err) - 1)
        if rc != PROVER_OK:
            self._h = None
            raise ProverError(rc, err.value.decode(errors="replace"))
        self._public_size = public_size
        return self

    @staticmethod
    def balanced_witness_range(n_vars, rank, world):
        """ug_groth16_balanced_witness_range: the witness range of a rank when ranks 0..2 also run an NTT chain each"""
        out = (C.c_ulonglong * 2)()
        if load().ug_groth16_balanced_witness_range(n_vars, rank, world, out) != PROVER_OK:
            raise ProverError(PROVER_ERROR, "invalid shard rank / count")
        return out[0], out[1]

    @staticmethod
    def shard_layout(n_vars, n_public, domain, rank, world, point_ranges=0, hbm_bytes=0):
        """ug_groth16_shard_layout: the rank's part of the many-device layout the library would choose (point_ranges = 0), or of
        the one with that many base-point ranges (world / point_ranges ranks share a range through bucket classes)"""
        out = (C.c_ulonglong * 12)()
        if load().ug_groth16_shard_layout(n_vars, n_public, domain, rank, world, point_ranges, hbm_bytes, out) != PROVER_OK:
            raise ProverError(PROVER_ERROR, "invalid shard rank / count or layout")
        return ShardLayout(list(out))

    @staticmethod
    def shard_ranges(n_vars, n_public, domain, rank, world, witness_range=None):
        """((witness first, end), (C first, end), (H first, end)) of a rank"""
        out = (C.c_ulonglong * 6)()
        wr = (C.c_ulonglong * 2)(*witness_range) if witness_range is not None else None
        if load().ug_groth16_shard_ranges(n_vars, n_public, domain, rank, world, wr, out) != PROVER_OK:
            raise ProverError(PROVER_ERROR, "invalid shard rank / count or witness range")
        return (out[0], out[1]), (out[2], out[3]), (out[4], out[5])

    def load_witness(self, wtns):
        err = C.create_string_buffer(1024)
        rc = load().ug_groth16_prover_load_witness(self._h, wtns, len(wtns), err, len(err) - 1)
        if rc != PROVER_OK:
            raise ProverError(rc, err.value.decode(errors="replace"))

    def load_witness_part(self, wtns, part):
        """part 0: the scalars of this rank's MSM slice; part 1: the rest (only ranks that run an H-polynomial chain)"""
        err = C.create_string_buffer(1024)
        rc = load().ug_groth16_prover_load_witness_part(self._h, wtns, len(wtns), part, err, len(err) - 1)
        if rc != PROVER_OK:
            raise ProverError(rc, err.value.decode(errors="replace"))

    def run(self):
        """Device part on the resident witness; returns this rank's 384-byte partial sums."""
        out = C.create_string_buffer(GROTH16_PARTIALS_SIZE)
        err = C.create_string_buffer(1024)
        rc = load().ug_groth16_prover_run(self._h, out, err, len(err) - 1)
        if rc != PROVER_OK:
            raise ProverError(rc, err.value.decode(errors="replace"))
        return out.raw

    def _phase(self, name, *args):
        out = C.create_string_buffer(GROTH16_PARTIALS_SIZE)
        err = C.create_string_buffer(1024)
        rc = getattr(load(), name)(self._h, *args, *([out] if name.endswith(("_msm", "_msm_end")) else []), err, len(err) - 1)
        if rc != PROVER_OK:
            raise ProverError(rc, err.value.decode(errors="replace"))
        return out.raw

    def run_witness_msm(self):
        """A, B1, B2, C partial sums of this rank (H record at infinity)"""
        return self._phase("ug_groth16_prover_run_witness_msm")

    def witness_msm_begin(self):
        """run_witness_msm without the wait: the four products are queued on the witness stream; drive the H branch (hpoly_chain,
        the slice exchange, hpoly_combine, run_h_msm) from this thread meanwhile, then witness_msm_end()"""
        self._phase("ug_groth16_prover_witness_msm_begin")

    def witness_msm_end(self):
        """the partial sums of the products witness_msm_begin queued (waits for them)"""
        return self._phase("ug_groth16_prover_witness_msm_end")

    def run_h_msm(self):
        """H partial sum of this rank from the h slice on the device (other records at infinity)"""
        return self._phase("ug_groth16_prover_run_h_msm")

    def hpoly_chain(self, which, device_ptr):
        """coset evaluations of polynomial `which` (0: A.w, 1: B.w, 2: their product) into device memory"""
        self._phase("ug_groth16_prover_hpoly_chain", which, C.c_void_p(device_ptr))

    def hpoly_combine(self, ptr_a, ptr_b, ptr_c):
        """this rank's slices of the three evaluation vectors (device pointers) -> its slice of h"""
        self._phase("ug_groth16_prover_hpoly_combine", C.c_void_p(ptr_a), C.c_void_p(ptr_b), C.c_void_p(ptr_c))

    def h_range(self):
        a, b, c = C.c_ulonglong(), C.c_ulonglong(), C.c_ulonglong()
        load().ug_groth16_prover_h_range(self._h, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    @staticmethod
    def add_partials(acc, other):
        a = C.create_string_buffer(bytes(acc), GROTH16_PARTIALS_SIZE)
        if load().ug_groth16_partials_add(a, bytes(other)) != PROVER_OK:
            raise ProverError(PROVER_ERROR, "partials add failed")
        return a.raw

    def finish(self, partials_sum):
        psz = C.c_ulonglong(groth16_proof_size())
        qsz = C.c_ulonglong(self._public_size)
        proof = C.create_string_buffer(psz.value)
        pub = C.create_string_buffer(max(qsz.value, 1))
        err = C.create_string_buffer(1024)
        rc = load().ug_groth16_prover_finish(self._h, bytes(partials_sum), proof, C.byref(psz), pub, C.byref(qsz), err, len(err) - 1)
        if rc != PROVER_OK:
            raise ProverError(rc, err.value.decode(errors="replace"))
        return proof.raw.split(b"\0", 1)[0].decode(), pub.raw.split(b"\0", 1)[0].decode()

    last_timings = _ProverBase.last_timings
    kernel_stats = _ProverBase.kernel_stats
    last_upload_ms = _ProverBase.last_upload_ms

    def close(self):
        if getattr(self, "_h", None):
            load().groth16_prover_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShardedUltraGrothProver(ShardedGroth16Prover):
    """One rank of a sharded UltraGroth prover (include/prover.h: ug_ultra_groth_prover_create_sharded and the phase
    calls it shares with the Groth16 one). Per proof: load_witness(uwtns) / round_commit() on every rank, the parts
    added (add_records), round_finish(sum) on one rank, apply_commitment(commitment) on every rank, then the Groth16
    phases and finish() on the rank that closed the round."""

    def __init__(self, zkey, device, rank, world):
        L = load()
        self._h = C.c_void_p()
        err = C.create_string_buffer(1024)
        rc = L.ug_ultra_groth_prover_create_sharded(C.byref(self._h), zkey, len(zkey), device, rank, world, err, len(err) - 1)
        if rc != PROVER_OK:
            self._h = None
            raise ProverError(rc, err.value.decode(errors="replace"))
        self._public_size = ultra_groth_public_size_for_zkey_buf(zkey)

    @classmethod
    def from_slices(cls, header, coefs, n_coefs, slices, device, rank, world, public_size):
        """ug_ultra_groth_prover_create_sharded_slices: header = zkey section 2, coefs = section 4 records (None: no chain on this
        rank), slices = (A, B1, B2, C1, C2, H, round_indexes, final_round_indexes) buffers holding this rank's part only
        (shard_ranges tells which)"""
        self = cls.__new__(cls)
        self._h = C.c_void_p()
        err = C.create_string_buffer(1024)
        sizes = (C.c_ulonglong * 8)(*[len(x) for x in slices])
        rc = load().ug_ultra_groth_prover_create_sharded_slices(C.byref(self._h), header, len(header), coefs, n_coefs, *slices, sizes,
                                                                device, rank, world, err, len(err) - 1)
        if rc != PROVER_OK:
            self._h = None
            raise ProverError(rc, err.value.decode(errors="replace"))
        self._public_size = public_size
        return self

    @staticmethod
    def shard_ranges(n_vars, domain, n_round, n_final, rank, world):
        """((witness first, end), (round set first, end), (final set first, end), (H first, end)) of a rank"""
        out = (C.c_ulonglong * 8)()
        if load().ug_ultra_groth_shard_ranges(n_vars, domain, n_round, n_final, rank, world, out) != PROVER_OK:
            raise ProverError(PROVER_ERROR, "invalid shard rank / count")
        return (out[0], out[1]), (out[2], out[3]), (out[4], out[5]), (out[6], out[7])

    def _call(self, name, *args):
        err = C.create_string_buffer(1024)
        rc = getattr(load(), name)(self._h, *args, err, len(err) - 1)
        if rc != PROVER_OK:
            raise ProverError(rc, err.value.decode(errors="replace"))

    def round_commit(self):
        out = C.create_string_buffer(64)
        self._call("ug_ultra_groth_prover_round_commit", out)
        return out.raw

    def round_finish(self, commit_sum):
        out = C.create_string_buffer(64)
        self._call("ug_ultra_groth_prover_round_finish", bytes(commit_sum), out)
        return out.raw

    def apply_commitment(self, commitment):
        self._call("ug_ultra_groth_prover_apply_commitment", bytes(commitment))

    @staticmethod
    def add_records(acc, other):
        a = C.create_string_buffer(bytes(acc), 64)
        if load().ug_g1_record_add(a, bytes(other)) != PROVER_OK:
            raise ProverError(PROVER_ERROR, "record add failed")
        return a.raw

    def finish(self, partials_sum):
        psz = C.c_ulonglong(ultra_groth_proof_size())
        qsz = C.c_ulonglong(self._public_size)
        proof = C.create_string_buffer(psz.value)
        pub = C.create_string_buffer(max(qsz.value, 1))
        err = C.create_string_buffer(1024)
        rc = load().ug_groth16_prover_finish(self._h, bytes(partials_sum), proof, C.byref(psz), pub, C.byref(qsz), err, len(err) - 1)
        if rc != PROVER_OK:
            raise ProverError(rc, err.value.decode(errors="replace"))
        return proof.raw.split(b"\0", 1)[0].decode(), pub.raw.split(b"\0", 1)[0].decode()

    def close(self):
        if getattr(self, "_h", None):
            load().ultra_groth_prover_destroy(self._h)
            self._h = None


# ---------------------------------------------------------------------------------------------------
# inner ABI (include/ultragroth_hip.h)

class Device:
    """A ug_ctx plus convenience wrappers in the reference's byte formats."""

    def __init__(self, device=0):
        self._L = load()
        self._h = C.c_void_p()
        _check(self._L.ug_ctx_create(C.byref(self._h), device))

    def close(self):
        if getattr(self, "_h", None):
            self._L.ug_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- raw handles
    def bases(self, points, n, g2=False, global_first=0, table_c=0):
        """table_c: also precompute the fixed-base window tables of that width (ug_bases_precompute)"""
        h = C.c_void_p()
        fn = self._L.ug_bases_create_g2 if g2 else self._L.ug_bases_create_g1
        _check(fn(self._h, points, n, global_first, C.byref(h)))
        b = _Handle(h, self._L.ug_bases_destroy, self)
        if table_c:
            _check(self._L.ug_bases_precompute(h, table_c))
        return b

    def bases_group(self, members, group_first, slots, table_c=0):
        """ug_bases_create_group_g1: members = [(points bytes, n, first scalar index), ...] (2 or 3 G1 sets sharing their scalars)"""
        k = len(members)
        keep = [_buf(bytes(p)) if not isinstance(p, C.Array) else p for p, _, _ in members]
        hosts = (C.c_void_p * k)(*[C.cast(b, C.c_void_p) for b in keep])
        ns = (C.c_uint64 * k)(*[n for _, n, _ in members])
        firsts = (C.c_uint64 * k)(*[f for _, _, f in members])
        h = C.c_void_p()
        _check(self._L.ug_bases_create_group_g1(self._h, k, hosts, ns, firsts, group_first, slots, table_c, C.byref(h)))
        g = _Handle(h, self._L.ug_bases_destroy, self)
        g.members = k
        return g

    def msm_group(self, group, schedule):
        """the K sums of a base group over a schedule (ug_msm_group_enqueue + ug_ctx_collect): K affine records"""
        outs = [C.create_string_buffer(64) for _ in range(group.members)]
        arr = (C.c_void_p * group.members)(*[C.cast(o, C.c_void_p) for o in outs])
        _check(self._L.ug_msm_group_enqueue(self._h, group.h, schedule.h, arr))
        _check(self._L.ug_ctx_collect(self._h))
        return [o.raw for o in outs]

    def dvec(self, n, data=None):
        h = C.c_void_p()
        _check(self._L.ug_dvec_create(self._h, n, C.byref(h)))
        v = _Handle(h, self._L.ug_dvec_destroy, self)
        if data is not None:
            _check(self._L.ug_dvec_upload(h, data, len(data) // 32))
        return v

    def download(self, dvec, first, n):
        out = C.create_string_buffer(n * 32)
        _check(self._L.ug_dvec_download(dvec.h, out, first, n))
        return out.raw

    def apply_lookup(self, dvec, w_idx, p_idx, chunks, table, lookup_size):
        """ug_dvec_apply_lookup: dvec[w_idx[i]] = push[p_idx[i]] in order (numpy uint32 index arrays, table bytes)"""
        import numpy as np
        w = np.ascontiguousarray(w_idx, dtype=np.uint32); p = np.ascontiguousarray(p_idx, dtype=np.uint32)
        c = np.ascontiguousarray(chunks, dtype=np.uint32)
        _check(self._L.ug_dvec_apply_lookup(dvec.h, w.ctypes.data, p.ctypes.data, len(w), c.ctypes.data, len(c), table, lookup_size))

    def schedule(self, dvec, first, count, table_c=0, classes=None):
        """classes (ug_schedule_set_classes): (q_log, first_residue, residues, specials, special_first, special_count)"""
        h = C.c_void_p()
        _check(self._L.ug_schedule_create(self._h, C.byref(h)))
        s = _Handle(h, self._L.ug_schedule_destroy, self)
        if classes is not None:
            _check(self._L.ug_schedule_set_classes(h, *classes))
        if table_c:
            _check(self._L.ug_schedule_build_tables(h, dvec.h, first, count, table_c))
        else:
            _check(self._L.ug_schedule_build(h, dvec.h, first, count))
        return s

    def table_window(self, n):
        return self._L.ug_msm_table_window(n)

    def mem_info(self):
        f, t = C.c_uint64(), C.c_uint64()
        _check(self._L.ug_ctx_mem_info(self._h, C.byref(f), C.byref(t)))
        return f.value, t.value

    def msm(self, bases, schedule, index_shift=0, g2=False):
        out = C.create_string_buffer(128 if g2 else 64)
        fn = self._L.ug_msm_g2 if g2 else self._L.ug_msm_g1
        _check(fn(self._h, bases.h, schedule.h, index_shift, out))
        return out.raw

    def msm_batch(self, bases_list, schedule, index_shifts=None):
        """ug_msm_batch: several products over one schedule with one host synchronisation; bases_list holds
        (bases handle, is_g2) pairs; returns the affine records"""
        n = len(bases_list)
        outs = [C.create_string_buffer(128 if g2 else 64) for _, g2 in bases_list]
        arr_b = (C.c_void_p * n)(*[b.h for b, _ in bases_list])
        arr_o = (C.c_void_p * n)(*[C.cast(o, C.c_void_p) for o in outs])
        arr_s = (C.c_int64 * n)(*index_shifts) if index_shifts is not None else None
        _check(self._L.ug_msm_batch(self._h, n, arr_b, schedule.h, arr_s, arr_o))
        return [o.raw for o in outs]

    # -- one-shot helpers
    def msm_g1(self, points, scalars, n, table_c=0):
        """sum scalars[i] * points[i]; points n x 64 B zkey records, scalars n x 32 B plain integers."""
        b = self.bases(points, n, table_c=table_c)
        s = self.schedule(self.dvec(max(n, 1), scalars if n else None), 0, n, table_c=table_c)
        return self.msm(b, s)

    def msm_g2(self, points, scalars, n, table_c=0):
        b = self.bases(points, n, g2=True, table_c=table_c)
        s = self.schedule(self.dvec(max(n, 1), scalars if n else None), 0, n, table_c=table_c)
        return self.msm(b, s, g2=True)

    def ntt(self, data, logn, inverse=False):
        buf = C.create_string_buffer(bytes(data), len(data))
        _check(self._L.ug_fr_ntt(self._h, buf, logn, 1 if inverse else 0))
        return buf.raw

    def field_op(self, field, op, a, b):
        n = len(a) // 32
        out = C.create_string_buffer(max(len(a), 1))
        _check(self._L.ug_field_op(self._h, field, op, out, bytes(a), bytes(b), n))
        return out.raw[:len(a)]

    def hpoly(self, coefs, ncoefs, domain, nvars):
        h = C.c_void_p()
        _check(self._L.ug_hpoly_create(self._h, coefs, ncoefs, domain, nvars, C.byref(h)))
        return _HPoly(h, self, domain)

    def timings(self, reset=False):
        a, b = C.c_double(), C.c_double()
        _check(self._L.ug_ctx_timings(self._h, C.byref(a), C.byref(b), 1 if reset else 0))
        return a.value, b.value

    def kernel_stats(self, g2=False, reset=False, which=None):
        a, l, e = C.c_double(), C.c_uint64(), C.c_uint64()
        w = (1 if g2 else 0) if which is None else which
        _check(self._L.ug_ctx_kernel_stats(self._h, w, C.byref(a), C.byref(l), C.byref(e), 1 if reset else 0))
        return a.value, l.value, e.value


class _Handle:
    def __init__(self, h, destroy, owner):
        self.h, self._destroy, self._owner = h, destroy, owner

    def __del__(self):
        try:
            if self.h and getattr(self._owner, "_h", None):
                self._destroy(self.h)
            self.h = None
        except Exception:
            pass


class _HPoly(_Handle):
    def __init__(self, h, dev, domain):
        super().__init__(h, dev._L.ug_hpoly_destroy, dev)
        self.dev, self.domain = dev, domain

    def run(self, wtns_dvec):
        """h vector (domain x 32 B plain integers) as a device vector"""
        out = self.dev.dvec(self.domain)
        _check(self.dev._L.ug_hpoly_run(self.h, wtns_dvec.h, out.h))
        return out

    def debug_abc(self):
        bufs = [C.create_string_buffer(self.domain * 32) for _ in range(3)]
        _check(self.dev._L.ug_hpoly_debug_abc(self.h, *bufs))
        return [b.raw for b in bufs]
