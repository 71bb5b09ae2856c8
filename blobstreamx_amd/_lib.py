"""ctypes loader for the product library blobstreamx_amd/lib/libbsx.so (C ABI: include/bsx.h).

The library is HIP-only: `context()` raises BsxError(ERR_NO_DEVICE) when no GPU is visible — there is no CPU
fallback anywhere in this package.
"""
import ctypes as C
import os
import subprocess
import sys
import threading

from . import types as T

_DIR = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("BSX_LIB_OVERRIDE") or os.path.join(_DIR, "lib", "libbsx.so")   # override: kernel A/B experiments
_lib = None
_lock = threading.Lock()
_ctx = {}

# every symbol include/bsx.h declares (checked by tests/test_abi.py without touching a GPU)
SYMBOLS = [
    "bsx_version", "bsx_init", "bsx_shutdown", "bsx_last_error", "bsx_status_str", "bsx_device_count",
    "bsx_map_witness_layout", "bsx_reduce_witness_layout", "bsx_witness_manifest",
    "bsx_encode_data_root_tuple", "bsx_get_data_commitment", "bsx_header_hashes", "bsx_data_commitment_inputs",
    "bsx_prove_subchain", "bsx_reduce", "bsx_prove_data_commitment", "bsx_prove_next_header_data_commitment",
    "bsx_verify_commits", "bsx_header_range", "bsx_next_header",
    "bsx_dev_alloc", "bsx_dev_free", "bsx_set_tuning", "bsx_trim", "bsx_dev_header_merkle", "bsx_dev_assemble_inputs", "bsx_dev_prove_subchain", "bsx_dev_reduce", "bsx_dev_reduce_strided", "bsx_dev_finalize",
    "bsx_dev_expand_witness", "bsx_dev_fill_end_hash", "bsx_dev_sha512_challenge", "bsx_dev_ed25519_verify",
    "bsx_dev_commit_tally", "bsx_dev_skip_check",
    "bsx_ed25519_keytable_bytes", "bsx_dev_ed25519_keytable", "bsx_dev_ed25519_verify_keyed", "bsx_ed25519_keytable_bytes_w", "bsx_dev_ed25519_keytable_w",
    "bsx_dev_ed25519_verify_keyed_w", "bsx_ed25519_verify_scratch_bytes",
    "bsx_dev_skip_eval", "bsx_find_block_to_request",
    "bsx_poseidon_permute", "bsx_poseidon_hash_no_pad", "bsx_poseidon_two_to_one", "bsx_poseidon_tree_digests",
    "bsx_witness_leaf_count", "bsx_poseidon_merkle_tree", "bsx_witness_merkle_caps",
    "bsx_dev_poseidon_permute", "bsx_dev_poseidon_leaf_hashes", "bsx_dev_witness_leaf_hashes", "bsx_dev_poseidon_merkle_caps",
    "bsx_ingest_last_error", "bsx_ingest_header_json", "bsx_ingest_signed_block_json", "bsx_ingest_data_commitment_json",
    "bsx_pipeline_create", "bsx_pipeline_destroy", "bsx_pipeline_upload", "bsx_pipeline_enable_input_streaming", "bsx_pipeline_step",
    "bsx_pipeline_join", "bsx_pipeline_set_allgather", "bsx_pipeline_get_results", "bsx_pipeline_buffer", "bsx_pipeline_set_timing",
    "bsx_pipeline_timing", "bsx_pipeline_timing2", "bsx_pipeline_autotune", "bsx_calibrate", "bsx_dev_verify_commits", "bsx_dev_verify_commits_scratch_bytes",
    "bsx_ed25519_decoded_r_bytes", "bsx_dev_ed25519_decode_r", "bsx_dev_ed25519_verify_keyed_r",
    "bsx_witness_manifest_section", "bsx_commit_witness_layout", "bsx_skip_witness_layout", "bsx_step_witness_layout",
    "bsx_header_range_witness_elements", "bsx_next_header_witness_elements", "bsx_prepare_process",
    "bsx_pipeline_set_rccl", "bsx_rccl_get_unique_id", "bsx_rccl_comm_init_rank", "bsx_rccl_comm_destroy", "bsx_pipeline_check_allgather",
    "bsx_batcher_create", "bsx_batcher_destroy", "bsx_submit_header_range", "bsx_submit_data_commitment_inputs", "bsx_submit_prove_subchain",
    "bsx_wait", "bsx_poll", "bsx_enable_coalescing", "bsx_batcher_get_stats", "bsx_context_batcher", "bsx_batcher_cork", "bsx_submit_map_job", "bsx_map_job", "bsx_header_range_cap", "bsx_next_header_cap", "bsx_verify_commits_cap",
    "bsx_submit_header_range_ex", "bsx_header_range_packed", "bsx_packed_headers_bound", "bsx_pack_headers", "bsx_unpack_headers", "bsx_host_register", "bsx_host_unregister",
]


class BsxError(RuntimeError):
    def __init__(self, status, message):
        self.status = status
        name = T.STATUS_NAMES[status] if 0 <= status < len(T.STATUS_NAMES) else str(status)
        super().__init__(f"{name}: {message}")


def build(force=False):
    """Compile libbsx.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    src = os.path.join(_DIR, "csrc")
    cmd = ["make", "-C", src, "-j4"] + (["-B"] if force else [])
    subprocess.run(cmd, check=True, capture_output=True)
    return _SO


def lib():
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(_SO):
                raise BsxError(T.ERR_NO_DEVICE, f"{_SO} is not built; run `python -c 'import __graft_entry__ as g; g.build()'`")
            # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64.so.7; loading it FIRST makes
            # libbsx.so (NEEDED libamdhip64.so.7) bind to the same copy, so torch tensors / streams / RCCL and our
            # kernels share one runtime.  The reverse order leaves torch without visible GPUs.
            if "torch" not in sys.modules and os.environ.get("BSX_NO_TORCH") != "1":
                try:
                    import torch  # noqa: F401
                except Exception:  # torch is plumbing, not a requirement of the C ABI
                    pass
            # An override (kernel A/B builds) is loaded RTLD_GLOBAL: the native caller harness (tests/hostcheck/libconcdrive.so, NEEDED
            # libbsx.so) then binds ITS calls to the override too — loaded locally, the harness pulled in the product library beside it and
            # the A/B compared the product with itself (round 6: the lone-caller A/B was void for that reason, profiles/r6_lone_caller_ab.txt)
            L = C.CDLL(_SO, mode=C.RTLD_GLOBAL) if os.environ.get("BSX_LIB_OVERRIDE") else C.CDLL(_SO)
            # the pipeline's 16 streams need their own hardware queues: asked for BEFORE the HIP runtime initialises (it usually has
            # not: importing torch does not initialise HIP).  An explicit, documented call — the library itself never touches the
            # environment (bsx.h bsx_prepare_process); BSX_KEEP_ENV=1 skips it
            if os.environ.get("BSX_KEEP_ENV") != "1":
                L.bsx_prepare_process()
            L.bsx_version.restype = C.c_uint32
            L.bsx_ed25519_keytable_bytes.restype = C.c_uint64
            L.bsx_ed25519_keytable_bytes_w.restype = C.c_uint64
            L.bsx_ed25519_verify_scratch_bytes.restype = C.c_uint64
            L.bsx_poseidon_tree_digests.restype = C.c_uint64
            L.bsx_witness_leaf_count.restype = C.c_uint32
            L.bsx_last_error.restype = C.c_char_p
            L.bsx_status_str.restype = C.c_char_p
            L.bsx_ingest_last_error.restype = C.c_char_p
            L.bsx_pipeline_destroy.restype = None
            L.bsx_batcher_destroy.restype = None
            L.bsx_context_batcher.restype = C.c_void_p
            L.bsx_dev_verify_commits_scratch_bytes.restype = C.c_uint64
            L.bsx_ed25519_decoded_r_bytes.restype = C.c_uint64
            L.bsx_header_range_witness_elements.restype = C.c_uint64
            L.bsx_next_header_witness_elements.restype = C.c_uint64
            for s in SYMBOLS:
                getattr(L, s)   # AttributeError here = header/library drift
            _lib = L
    return _lib


def check(rc, allow=()):
    if rc != T.OK and rc not in allow:
        raise BsxError(rc, lib().bsx_last_error().decode(errors="replace"))
    return rc


def last_error():
    return lib().bsx_last_error().decode(errors="replace")


def context(device=0):
    """One bsx_ctx per device, created on first use.  Raises when no GPU is present."""
    with _lock:
        c = _ctx.get(device)
    if c is not None:
        return c
    L = lib()
    h = C.c_void_p()
    rc = L.bsx_init(C.c_int(device), C.byref(h))
    if rc != T.OK:
        raise BsxError(rc, L.bsx_last_error().decode(errors="replace"))
    with _lock:
        _ctx[device] = h
    return h


def p(a):
    """numpy array -> void* (keeps the array alive through the returned object)."""
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def dp(t):
    """torch device tensor / int address -> void*"""
    if t is None:
        return None
    if isinstance(t, int):
        return C.c_void_p(t)
    return C.c_void_p(t.data_ptr())


class DeviceBuffer:
    """A bsx_dev_alloc block (VMM-backed device memory for the large streaming buffers) seen as a torch int64 tensor
    through __cuda_array_interface__ — zero copy; the block is released when the last tensor view is gone."""

    def __init__(self, n_int64, device=0):
        self.device, self.n = device, int(n_int64)
        self.ptr = C.c_void_p()
        check(lib().bsx_dev_alloc(context(device), C.c_uint64(self.n * 8), C.byref(self.ptr)))
        self.__cuda_array_interface__ = {"shape": (self.n,), "typestr": "<i8", "data": (int(self.ptr.value), False), "version": 2}

    def tensor(self):
        import torch
        return torch.as_tensor(self, device=f"cuda:{self.device}")

    def __del__(self):
        try:
            if self.ptr and self.ptr.value:
                lib().bsx_dev_free(context(self.device), self.ptr)
                self.ptr = C.c_void_p()
        except Exception:
            pass
