"""ctypes loader for libultragroth_hip.so (built in-tree by ultragroth_amd/csrc/Makefile).

The product path has no CPU fallback: if the shared library is missing or the HIP runtime has no
device, every compute entry point fails loudly.
"""
import ctypes as C
import importlib.util
import os
import subprocess
import sys

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
# ULTRAGROTH_LIB: another build of the same library (tools: the -DUG_MEASURE build, libultragroth_hip_measure.so)
LIB_PATH = os.environ.get("ULTRAGROTH_LIB") or os.path.join(_CSRC, "libultragroth_hip.so")

# every symbol that include/ultragroth_hip.h and include/prover.h declare
INNER_SYMBOLS = [
    "ug_last_error", "ug_device_count", "ug_ctx_create", "ug_ctx_create_priority", "ug_ctx_destroy", "ug_ctx_sync",
    "ug_bases_create_g1", "ug_bases_create_g2", "ug_bases_create_tables_g1", "ug_bases_create_tables_g2", "ug_bases_destroy",
    "ug_msm_table_window", "ug_bases_tables_bytes", "ug_bases_precompute", "ug_ctx_mem_info", "ug_schedule_build_tables",
    "ug_dvec_create", "ug_dvec_upload", "ug_dvec_upload_range", "ug_dvec_upload_idle", "ug_dvec_download", "ug_dvec_gather", "ug_dvec_scatter", "ug_dvec_apply_lookup", "ug_fr_lookup_table", "ug_index_create", "ug_index_destroy", "ug_dvec_gather_index", "ug_dvec_wrap", "ug_dvec_size", "ug_dvec_destroy",
    "ug_schedule_create", "ug_schedule_build", "ug_schedule_destroy", "ug_schedule_set_classes",
    "ug_msm_g1", "ug_msm_g2", "ug_msm_batch", "ug_msm_batch_enqueue", "ug_ctx_collect", "ug_ctx_wait",
    "ug_hpoly_create", "ug_hpoly_run", "ug_hpoly_chain", "ug_hpoly_combine", "ug_hpoly_debug_abc", "ug_hpoly_destroy",
    "ug_fr_ntt", "ug_field_op", "ug_synth_points", "ug_ctx_timings", "ug_ctx_kernel_stats", "ug_ctx_abandon", "ug_test_inject_fault",
    "ug_bases_create_group_g1", "ug_bases_members", "ug_points_all_infinity", "ug_msm_group_enqueue", "ug_msm_witness_enqueue", "ug_dvec_device_ptr", "ug_dvec_copy", "ug_dvec_copy_via", "ug_sort_plan",
    "ug_bases_drop_tables", "ug_bases_table_window", "ug_schedule_trim", "ug_ctx_trim",
    "ug_ctx_defer_tables", "ug_bases_tables_alloc", "ug_bases_tables_adopt", "ug_bases_tables_step", "ug_bases_tables_ready",
    "ug_graph_begin", "ug_graph_end", "ug_graph_abort", "ug_graph_valid", "ug_graph_nodes", "ug_graph_launch", "ug_graph_destroy",
]
VERIFIER_SYMBOLS = ["groth16_verify", "ultra_groth_verify"]        # include/verifier.h
OUTER_SYMBOLS = [
    "groth16_public_size_for_zkey_buf", "ultra_groth_public_size_for_zkey_buf",
    "groth16_public_size_for_zkey_file", "ultra_groth_public_size_for_zkey_file",
    "groth16_proof_size", "ultra_groth_proof_size",
    "groth16_prover_create", "ultra_groth_prover_create",
    "groth16_prover_create_zkey_file", "ultra_groth_prover_create_zkey_file",
    "groth16_prover_prove", "ultra_groth_prover_prove",
    "groth16_prover_destroy", "ultra_groth_prover_destroy",
    "groth16_prover", "ultra_groth_prover",
    "groth16_prover_zkey_file", "ultra_groth_prover_zkey_file",
    "ug_registry_create", "ug_registry_load", "ug_registry_load_file", "ug_registry_prove", "ug_registry_evict", "ug_registry_info",
    "ug_registry_destroy",
    "ug_test_set_blinding", "ug_prover_last_timings", "ug_prover_kernel_stats", "ug_prover_last_upload_ms", "ug_prover_tables_ready",
    "ug_groth16_prover_create_sharded", "ug_groth16_prover_create_sharded_range", "ug_groth16_prover_create_sharded_slices",
    "ug_groth16_shard_ranges", "ug_groth16_balanced_witness_range", "ug_groth16_prover_load_witness_part",
    "ug_groth16_shard_layout", "ug_groth16_prover_create_sharded_layout",
    "ug_ultra_groth_prover_create_sharded", "ug_ultra_groth_shard_ranges", "ug_ultra_groth_prover_create_sharded_slices",
    "ug_ultra_groth_prover_round_commit", "ug_ultra_groth_prover_round_finish",
    "ug_ultra_groth_prover_apply_commitment", "ug_g1_record_add",
    "ug_groth16_prover_load_witness", "ug_groth16_prover_run", "ug_groth16_prover_prove_resident",
    "ug_groth16_partials_add", "ug_groth16_prover_finish",
    "ug_groth16_prover_run_witness_msm", "ug_groth16_prover_run_h_msm", "ug_groth16_prover_hpoly_chain",
    "ug_groth16_prover_hpoly_combine", "ug_groth16_prover_h_range",
    "ug_groth16_prover_witness_msm_begin", "ug_groth16_prover_witness_msm_end",
]


def build(jobs=4):
    """Compile every HIP extension for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    subprocess.check_call(["make", "-s", "-C", _CSRC, "-j%d" % jobs])
    return LIB_PATH


_lib = None


def _one_hip_runtime():
    """One HIP runtime per process, whatever the import order. A torch wheel bundles its own libamdhip64.so, whose SONAME
    (libamdhip64.so.7) is the name this library needs: when torch is imported first the dynamic linker hands that copy to this
    library as well. The other way round used to be a hazard (round-2 review): this library brought in /opt/rocm's copy, torch's
    own NEEDED entry (`libamdhip64.so`, no version) did not match it, a second runtime was loaded and torch saw no devices. So
    when a torch with a bundled runtime is installed, that runtime is loaded here, before the library, and both orders end with
    the same single copy. Programs that never touch Python (the `prover` CLI, cgo / JNI callers) use the system's runtime;
    ULTRAGROTH_HIP_RUNTIME=system keeps that one here too (a process that then imports torch is on its own)."""
    if os.environ.get("ULTRAGROTH_HIP_RUNTIME") == "system" or "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    bundled = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(bundled):
        C.CDLL(bundled, mode=C.RTLD_GLOBAL)


def hip_runtimes_loaded():
    """paths of every libamdhip64 mapped into this process (tests: exactly one)"""
    seen = []
    with open("/proc/self/maps") as f:
        for line in f:
            path = line.rsplit(" ", 1)[-1].strip()
            if "libamdhip64" in os.path.basename(path) and path not in seen:
                seen.append(path)
    return seen


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libultragroth_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C ultragroth_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    _one_hip_runtime()
    L = C.CDLL(LIB_PATH)
    vp, u64, i64, u32 = C.c_void_p, C.c_uint64, C.c_int64, C.c_uint32
    pp = C.POINTER(C.c_void_p)
    ull, pull = C.c_ulonglong, C.POINTER(C.c_ulonglong)
    L.ug_last_error.restype = C.c_char_p
    L.ug_device_count.restype = C.c_int
    L.ug_ctx_create.argtypes = [pp, C.c_int]
    L.ug_ctx_create_priority.argtypes = [pp, C.c_int, C.c_int]
    L.ug_ctx_destroy.argtypes = [vp]; L.ug_ctx_destroy.restype = None
    L.ug_ctx_sync.argtypes = [vp]
    L.ug_ctx_abandon.argtypes = [vp]; L.ug_ctx_abandon.restype = None
    L.ug_test_inject_fault.argtypes = [C.c_int, C.c_int]
    L.ug_bases_create_group_g1.argtypes = [vp, C.c_int, vp, vp, vp, u64, u64, C.c_int, pp]
    L.ug_bases_members.argtypes = [vp]
    L.ug_points_all_infinity.argtypes = [vp, u64, u64]
    L.ug_msm_group_enqueue.argtypes = [vp, vp, vp, vp]
    L.ug_ctx_collect.argtypes = [vp]
    L.ug_dvec_device_ptr.argtypes = [vp]; L.ug_dvec_device_ptr.restype = vp
    L.ug_dvec_copy.argtypes = [vp, u64, vp, u64, u64]
    L.ug_sort_plan.argtypes = [C.c_int, vp, vp]
    for n in ("ug_bases_create_g1", "ug_bases_create_g2"):
        getattr(L, n).argtypes = [vp, vp, u64, u64, pp]
    L.ug_bases_destroy.argtypes = [vp]; L.ug_bases_destroy.restype = None
    L.ug_msm_table_window.argtypes = [u64]
    L.ug_bases_tables_bytes.argtypes = [u64, C.c_int, C.c_int]; L.ug_bases_tables_bytes.restype = u64
    L.ug_bases_precompute.argtypes = [vp, C.c_int]
    L.ug_ctx_mem_info.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
    L.ug_schedule_build_tables.argtypes = [vp, vp, u64, u64, C.c_int]
    L.ug_dvec_create.argtypes = [vp, u64, pp]
    L.ug_dvec_upload.argtypes = [vp, vp, u64]
    L.ug_dvec_upload_idle.argtypes = [vp, vp, u64]
    L.ug_dvec_download.argtypes = [vp, vp, u64, u64]
    L.ug_dvec_gather.argtypes = [vp, vp, vp, u64]
    L.ug_dvec_scatter.argtypes = [vp, vp, vp, u64]
    L.ug_index_create.argtypes = [vp, vp, u64, pp]
    L.ug_index_destroy.argtypes = [vp]; L.ug_index_destroy.restype = None
    L.ug_dvec_gather_index.argtypes = [vp, vp, vp]
    L.ug_dvec_apply_lookup.argtypes = [vp, vp, vp, u64, vp, u64, vp, u64]
    L.ug_dvec_wrap.argtypes = [vp, vp, u64, pp]
    L.ug_dvec_size.argtypes = [vp]; L.ug_dvec_size.restype = u64
    L.ug_dvec_destroy.argtypes = [vp]; L.ug_dvec_destroy.restype = None
    L.ug_schedule_create.argtypes = [vp, pp]
    L.ug_schedule_build.argtypes = [vp, vp, u64, u64]
    L.ug_schedule_set_classes.argtypes = [vp, C.c_int, u32, u32, u32, u64, u64]
    L.ug_schedule_destroy.argtypes = [vp]; L.ug_schedule_destroy.restype = None
    L.ug_msm_g1.argtypes = [vp, vp, vp, i64, vp]
    L.ug_msm_g2.argtypes = [vp, vp, vp, i64, vp]
    L.ug_msm_batch.argtypes = [vp, C.c_int, vp, vp, vp, vp]
    L.ug_hpoly_create.argtypes = [vp, vp, u64, u32, u32, pp]
    L.ug_hpoly_run.argtypes = [vp, vp, vp]
    L.ug_hpoly_chain.argtypes = [vp, vp, C.c_int, vp]
    L.ug_hpoly_combine.argtypes = [vp, vp, vp, vp, u64, u64, vp]
    L.ug_hpoly_debug_abc.argtypes = [vp, vp, vp, vp]
    L.ug_hpoly_destroy.argtypes = [vp]; L.ug_hpoly_destroy.restype = None
    L.ug_fr_ntt.argtypes = [vp, vp, C.c_int, C.c_int]
    L.ug_field_op.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, u64]
    L.ug_synth_points.argtypes = [vp, C.c_int, vp, u64, u64, vp]
    L.ug_ctx_timings.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int]
    L.ug_ctx_kernel_stats.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(u64), C.POINTER(u64), C.c_int]
    for n in VERIFIER_SYMBOLS:
        getattr(L, n).argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_ulong]
    # outer API (include/prover.h)
    for n in ("groth16_public_size_for_zkey_buf", "ultra_groth_public_size_for_zkey_buf"):
        getattr(L, n).argtypes = [vp, ull, pull, vp, ull]
    for n in ("groth16_public_size_for_zkey_file", "ultra_groth_public_size_for_zkey_file"):
        getattr(L, n).argtypes = [C.c_char_p, pull, vp, ull]
    for n in ("groth16_proof_size", "ultra_groth_proof_size"):
        getattr(L, n).argtypes = [pull]; getattr(L, n).restype = None
    for n in ("groth16_prover_create", "ultra_groth_prover_create"):
        getattr(L, n).argtypes = [pp, vp, ull, vp, ull]
    for n in ("groth16_prover_create_zkey_file", "ultra_groth_prover_create_zkey_file"):
        getattr(L, n).argtypes = [pp, C.c_char_p, vp, ull]
    for n in ("groth16_prover_prove", "ultra_groth_prover_prove"):
        getattr(L, n).argtypes = [vp, vp, ull, vp, pull, vp, pull, vp, ull]
    for n in ("groth16_prover_destroy", "ultra_groth_prover_destroy"):
        getattr(L, n).argtypes = [vp]; getattr(L, n).restype = None
    for n in ("groth16_prover", "ultra_groth_prover"):
        getattr(L, n).argtypes = [vp, ull, vp, ull, vp, pull, vp, pull, vp, ull]
    for n in ("groth16_prover_zkey_file", "ultra_groth_prover_zkey_file"):
        getattr(L, n).argtypes = [C.c_char_p, vp, ull, vp, pull, vp, pull, vp, ull]
    L.ug_test_set_blinding.argtypes = [vp, ull]; L.ug_test_set_blinding.restype = C.c_int
    L.ug_prover_last_timings.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.ug_prover_kernel_stats.argtypes = [vp, C.c_int, C.POINTER(C.c_double), pull, pull, C.c_int]
    L.ug_prover_last_upload_ms.argtypes = [vp, C.POINTER(C.c_double)]
    L.ug_prover_tables_ready.argtypes = [vp, C.c_int]
    L.ug_registry_create.argtypes = [pp, C.c_int, ull, vp, ull]
    L.ug_registry_load.argtypes = [vp, C.c_char_p, vp, ull, vp, ull]
    L.ug_registry_load_file.argtypes = [vp, C.c_char_p, vp, ull]
    L.ug_registry_prove.argtypes = [vp, C.c_char_p, vp, ull, vp, pull, vp, pull, vp, ull]
    L.ug_registry_evict.argtypes = [vp, C.c_char_p, vp, ull]
    L.ug_registry_info.argtypes = [vp, C.c_char_p, pull, C.POINTER(C.c_int), pull]
    L.ug_registry_destroy.argtypes = [vp]; L.ug_registry_destroy.restype = None
    L.ug_groth16_prover_create_sharded.argtypes = [pp, vp, ull, C.c_int, C.c_int, C.c_int, vp, ull]
    L.ug_ultra_groth_prover_create_sharded.argtypes = [pp, vp, ull, C.c_int, C.c_int, C.c_int, vp, ull]
    L.ug_ultra_groth_prover_round_commit.argtypes = [vp, vp, vp, ull]
    L.ug_ultra_groth_prover_round_finish.argtypes = [vp, vp, vp, vp, ull]
    L.ug_ultra_groth_prover_apply_commitment.argtypes = [vp, vp, vp, ull]
    L.ug_g1_record_add.argtypes = [vp, vp]
    L.ug_groth16_prover_create_sharded_range.argtypes = [pp, vp, ull, C.c_int, C.c_int, C.c_int, ull, ull, vp, ull]
    L.ug_groth16_prover_load_witness.argtypes = [vp, vp, ull, vp, ull]
    L.ug_groth16_prover_load_witness_part.argtypes = [vp, vp, ull, C.c_int, vp, ull]
    L.ug_groth16_shard_ranges.argtypes = [ull, ull, ull, C.c_int, C.c_int, vp, vp]
    L.ug_groth16_shard_layout.argtypes = [ull, ull, ull, C.c_int, C.c_int, C.c_int, ull, vp]
    L.ug_ultra_groth_shard_ranges.argtypes = [ull, ull, ull, ull, C.c_int, C.c_int, vp]
    L.ug_ultra_groth_prover_create_sharded_slices.argtypes = [pp, vp, ull, vp, ull] + [vp] * 9 + [C.c_int, C.c_int, C.c_int, vp, ull]
    L.ug_groth16_prover_create_sharded_layout.argtypes = [pp, vp, ull, vp, ull, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, ull]
    L.ug_groth16_balanced_witness_range.argtypes = [ull, C.c_int, C.c_int, vp]
    L.ug_groth16_prover_create_sharded_slices.argtypes = [pp, vp, ull, vp, ull, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, ull]
    L.ug_dvec_upload_range.argtypes = [vp, vp, u64, u64, vp]
    L.ug_fr_lookup_table.argtypes = [vp, vp, vp, u64, vp]
    L.ug_groth16_prover_run.argtypes = [vp, vp, vp, ull]
    L.ug_groth16_prover_prove_resident.argtypes = [vp, vp, pull, vp, pull, vp, ull]
    L.ug_groth16_partials_add.argtypes = [vp, vp]
    L.ug_groth16_prover_run_witness_msm.argtypes = [vp, vp, vp, ull]
    L.ug_groth16_prover_run_h_msm.argtypes = [vp, vp, vp, ull]
    L.ug_groth16_prover_witness_msm_begin.argtypes = [vp, vp, ull]
    L.ug_groth16_prover_witness_msm_end.argtypes = [vp, vp, vp, ull]
    L.ug_groth16_prover_hpoly_chain.argtypes = [vp, C.c_int, vp, vp, ull]
    L.ug_groth16_prover_hpoly_combine.argtypes = [vp, vp, vp, vp, vp, ull]
    L.ug_groth16_prover_h_range.argtypes = [vp, pull, pull, pull]
    L.ug_groth16_prover_finish.argtypes = [vp, vp, vp, pull, vp, pull, vp, ull]
    _lib = L
    return L
