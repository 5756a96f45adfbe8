"""ctypes binding of libsclens_hip.so (include/sclens_hip.h). Fails loudly: there is no CPU fallback.

The shared library is built in-tree by `__graft_entry__.build()` / `make -C sclens_amd/csrc`.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsclens_hip.so")

ERR_NAMES = {1: "ARG", 2: "NO_DEVICE", 3: "OOM", 4: "HIP", 5: "NOCONV", 6: "NAN", 7: "STATE"}

c_f32p = C.POINTER(C.c_float)
c_f64p = C.POINTER(C.c_double)
c_i64p = C.POINTER(C.c_int64)
c_i32p = C.POINTER(C.c_int32)
c_u8p = C.POINTER(C.c_uint8)
c_u32p = C.POINTER(C.c_uint32)
vp = C.c_void_p
# int allreduce(void* user, void* dev_ptr, int64_t count, int dtype): see sclens_hip_session_create_sharded
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int)
# int reduce(void* user, void* dev_ptr, int64_t count, int dtype, int root): the sum lands on `root` only
REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int)
i64 = C.c_int64

# name -> (restype, argtypes); every symbol declared in include/sclens_hip.h
SIGNATURES = {
    "sclens_hip_create": (C.c_int, [C.POINTER(vp), C.c_int]),
    "sclens_hip_destroy": (None, [vp]),
    "sclens_hip_last_error": (C.c_char_p, [vp]),
    "sclens_hip_version": (C.c_char_p, []),
    "sclens_hip_set_timing": (C.c_int, [vp, C.c_int]),
    "sclens_hip_get_timing": (C.c_int, [vp, C.c_char_p, c_f64p, c_i64p]),
    "sclens_hip_reset_timing": (C.c_int, [vp]),
    "sclens_hip_set_option": (C.c_int, [vp, C.c_char_p, i64]),
    "sclens_hip_get_option": (C.c_int, [vp, C.c_char_p, c_i64p]),
    "sclens_hip_copy_options": (C.c_int, [vp, vp]),
    "sclens_hip_stream": (vp, [vp]),
    "sclens_hip_trim": (C.c_int, [C.c_int]),
    "sclens_hip_pool_set_cap": (C.c_int, [C.c_int, i64]),
    "sclens_hip_pool_peak": (i64, [C.c_int, C.c_int]),
    "sclens_hip_release_scratch": (C.c_int, [vp, C.c_char_p]),
    "sclens_hip_pool_stats": (C.c_int, [C.c_int, c_i64p, c_i64p, c_i64p, c_i64p]),
    "sclens_hip_symv_probe": (C.c_int, [vp, i64, c_i64p, c_f64p, c_f64p]),
    "sclens_hip_symv_profile": (C.c_int, [vp, C.c_int]),
    "sclens_hip_symv_profile_read": (C.c_int, [vp, c_i64p, c_f64p, c_f64p]),
    "sclens_hip_session_set_int": (C.c_int, [vp, C.c_char_p, i64]),
    "sclens_hip_session_get_int": (C.c_int, [vp, C.c_char_p, c_i64p]),
    "sclens_hip_session_slot_ld": (i64, [vp]),
    "sclens_hip_session_export_slot": (C.c_int, [vp, i64, i64, vp]),
    "sclens_hip_session_import_slot": (C.c_int, [vp, i64, i64, i64, vp]),
    "sclens_hip_wishart_matrix_f32": (C.c_int, [vp, c_f32p, i64, i64, C.c_int, c_f32p]),
    "sclens_hip_get_eigen_f32": (C.c_int, [vp, c_f32p, i64, c_f32p, c_f32p]),
    "sclens_hip_corr_mat_f32": (C.c_int, [vp, c_f32p, i64, i64, c_f32p, i64, c_f32p]),
    "sclens_hip_preprocess_csc": (C.c_int, [vp, i64, i64, c_i64p, c_i32p, c_f32p, c_u8p, c_u8p, C.c_double, C.c_double, C.c_double,
                                            C.c_double, i64, i64, i64, C.c_double, C.c_double, c_u8p, c_i64p, c_i64p, c_i64p,
                                            c_i64p]),
    "sclens_hip_preprocess_gather": (C.c_int, [vp, c_i64p, c_i32p, c_f32p]),
    "sclens_hip_preprocess_keep": (C.c_int, [vp, C.POINTER(vp)]),
    "sclens_hip_counts_upload": (C.c_int, [vp, i64, i64, c_i64p, c_i32p, c_f32p, C.POINTER(vp)]),
    "sclens_hip_counts_info": (C.c_int, [vp, c_i64p, c_i64p, c_i64p]),
    "sclens_hip_counts_download": (C.c_int, [vp, vp, c_i64p, c_i32p, c_f32p]),
    "sclens_hip_counts_destroy": (None, [vp]),
    "sclens_hip_host_alloc": (C.c_int, [i64, C.POINTER(vp)]),
    "sclens_hip_host_free": (None, [vp]),
    "sclens_hip_session_create_from_counts": (C.c_int, [vp, vp, C.POINTER(vp)]),
    "sclens_hip_pattern_create_drawn_from_counts": (C.c_int, [vp, vp, C.c_uint64, C.POINTER(vp), c_i64p]),
    "sclens_hip_session_create_sharded": (C.c_int, [vp, i64, i64, i64, i64, c_i64p, c_i32p, c_f32p, i64, c_u32p, c_u32p, ALLREDUCE_FN,
                                                    vp, C.POINTER(vp)]),
    "sclens_hip_session_set_reducer": (C.c_int, [vp, ALLREDUCE_FN, vp]),
    "sclens_hip_session_set_reduce_to": (C.c_int, [vp, REDUCE_FN, vp]),
    "sclens_hip_session_create_sharded_drawn": (C.c_int, [vp, i64, i64, i64, i64, c_i64p, c_i32p, c_f32p, i64, C.c_uint64, ALLREDUCE_FN, vp,
                                                          C.POINTER(vp), c_i64p]),
    "sclens_hip_session_set_candidate_range": (C.c_int, [vp, i64, i64]),
    "sclens_hip_session_create_chunked": (C.c_int, [vp, i64, i64, C.c_int, i64, C.c_uint64, C.POINTER(vp)]),
    "sclens_hip_session_chunk_add": (C.c_int, [vp, C.c_int, C.c_int, i64, i64, c_i64p, c_i32p, c_f32p]),
    "sclens_hip_session_chunk_commit": (C.c_int, [vp]),
    "sclens_hip_session_null_spectrum_chunked": (C.c_int, [vp, c_f64p]),
    "sclens_hip_session_local_candidates": (C.c_int, [vp, c_u32p, c_u32p]),
    "sclens_hip_session_search_round_seeded": (C.c_int, [vp, C.POINTER(C.c_uint64), c_i64p, c_i32p, C.c_int, C.c_int, i64, c_f64p, c_i64p]),
    "sclens_hip_session_perturb_round_seeded": (C.c_int, [vp, c_i64p, C.POINTER(C.c_uint64), c_i64p, c_i32p, C.c_int, C.c_int, i64, c_f64p,
                                                          c_i64p]),
    "sclens_hip_session_shared_buffer": (C.c_int, [vp, C.c_int, i64, i64, c_f64p, C.POINTER(vp), c_i64p, c_i64p, c_i64p]),
    "sclens_hip_pattern_create": (C.c_int, [vp, i64, i64, c_i64p, c_i32p, c_f32p, i64, c_u32p, c_u32p, C.POINTER(vp)]),
    "sclens_hip_pattern_create_drawn": (C.c_int, [vp, i64, i64, c_i64p, c_i32p, c_f32p, C.c_uint64, C.POINTER(vp), c_i64p]),
    "sclens_hip_pattern_candidates": (C.c_int, [vp, vp, c_u32p, c_u32p]),
    "sclens_hip_pattern_download": (C.c_int, [vp, vp, C.c_int, vp]),
    "sclens_hip_pattern_destroy": (None, [vp]),
    "sclens_hip_session_set_pattern": (C.c_int, [vp, vp]),
    "sclens_hip_session_null_spectrum_pattern": (C.c_int, [vp, vp, c_f64p]),
    "sclens_hip_scale_csc_f32": (C.c_int, [vp, i64, i64, c_i64p, c_i32p, c_f32p, C.c_int, C.c_int, c_f32p, c_f64p, c_f64p, c_f64p,
                                           c_f64p, c_f64p]),
    "sclens_hip_corr_colmax_f32": (C.c_int, [vp, c_f32p, i64, i64, c_f32p, i64, C.c_int, c_f32p]),
    "sclens_hip_gram_binary_f32": (C.c_int, [vp, i64, i64, c_i64p, c_i32p, c_f32p, C.c_int, C.c_float, c_f32p]),
    "sclens_hip_gram_counts_f32": (C.c_int, [vp, i64, i64, c_i64p, c_i32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_float, c_f32p]),
    "sclens_hip_get_eigvec_f32": (C.c_int, [vp, c_f32p, i64, i64, i64, c_f32p, c_f32p, c_i64p]),
    "sclens_hip_get_denoised_f32": (C.c_int, [vp, c_f32p, i64, i64, c_f32p, i64, c_f64p, c_f64p, c_f64p, c_f64p, c_f64p, c_f32p]),
    "sclens_mp_calculation": (C.c_int, [c_f64p, i64, c_f64p, i64, c_f64p, c_f64p, c_u8p]),
    "sclens_tw": (C.c_int, [i64, c_f64p, i64, c_f64p, c_f64p, c_f64p, c_f64p]),
    "sclens_mp_check": (C.c_int, [c_f64p, i64, C.c_double, c_f64p, C.POINTER(C.c_int)]),
    "sclens_robust_scores": (C.c_int, [c_f64p, i64, i64, c_f64p, c_f64p]),
    "sclens_noise_baseline_exact": (C.c_double, [i64]),
    "sclens_draw_zero_candidates": (C.c_int, [i64, i64, c_i64p, c_i32p, C.c_uint64, c_u32p, c_u32p, c_i64p]),
    "sclens_draw_null_matrix": (C.c_int, [i64, i64, c_i64p, c_f32p, C.c_uint64, c_i32p, c_f32p]),
    "sclens_sample_without_replacement": (C.c_int, [C.c_uint64, i64, C.c_uint64, c_u32p]),
    "sclens_hip_session_search_step_seeded": (C.c_int, [vp, C.c_uint64, i64, i64, c_f64p, c_i64p]),
    "sclens_hip_session_perturb_seeded": (C.c_int, [vp, i64, C.c_uint64, i64, i64, c_f64p, c_i64p]),
    "sclens_hip_session_create": (C.c_int, [vp, i64, i64, c_i64p, c_i32p, c_f32p, i64, c_u32p, c_u32p, C.POINTER(vp)]),
    "sclens_hip_session_destroy": (None, [vp]),
    "sclens_hip_session_clone": (C.c_int, [vp, vp, C.POINTER(vp)]),
    "sclens_hip_session_spectrum": (C.c_int, [vp, c_i64p, c_i32p, c_f32p, c_f64p, c_f64p, c_f64p, c_f64p, c_f64p, c_f64p, c_f64p]),
    "sclens_hip_session_null_spectrum": (C.c_int, [vp, c_i64p, c_i32p, c_f32p, c_f64p]),
    "sclens_hip_session_data_spectrum": (C.c_int, [vp, c_f64p, c_f64p, c_f64p, c_f64p, c_f64p, c_f64p]),
    "sclens_hip_session_adopt": (C.c_int, [vp, vp, C.c_int]),
    "sclens_hip_session_signal_vectors": (C.c_int, [vp, i64, c_f32p]),
    "sclens_hip_session_refine_eigenvalues": (C.c_int, [vp, i64, i64, c_f64p]),
    "sclens_hip_session_binary_basis": (C.c_int, [vp, c_f64p, c_i64p]),
    "sclens_hip_session_search_step": (C.c_int, [vp, c_u32p, i64, i64, c_f64p, c_i64p]),
    "sclens_hip_session_perturb": (C.c_int, [vp, i64, c_u32p, i64, i64, c_f64p, c_i64p]),
    "sclens_hip_session_get_perturbed": (C.c_int, [vp, i64, c_f32p]),
    "sclens_hip_session_robustness": (C.c_int, [vp, i64, c_i32p, c_f64p]),
    "sclens_hip_session_gene_basis": (C.c_int, [vp, c_f64p, c_f32p]),
    "sclens_hip_comm_unique_id": (C.c_int, [vp, c_u8p]),
    "sclens_hip_comm_create": (C.c_int, [vp, c_u8p, C.c_int, C.c_int, C.POINTER(vp)]),
    "sclens_hip_comm_destroy": (None, [vp]),
    "sclens_hip_comm_info": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "sclens_hip_comm_stats": (C.c_int, [vp, c_i64p, c_f64p]),
    "sclens_hip_comm_last_error": (C.c_char_p, [vp]),
    "sclens_hip_comm_allreduce": (C.c_int, [vp, vp, i64, C.c_int]),
    "sclens_hip_comm_broadcast": (C.c_int, [vp, vp, i64, C.c_int]),
    "sclens_hip_comm_allgather": (C.c_int, [vp, vp, vp, i64]),
    "sclens_hip_comm_allgather_host": (C.c_int, [vp, vp, vp, i64]),
    "sclens_hip_comm_broadcast_host": (C.c_int, [vp, vp, i64, C.c_int]),
    "sclens_hip_comm_allreduce_cb": (C.c_int, [vp, vp, i64, C.c_int]),
    "sclens_hip_comm_reduce_cb": (C.c_int, [vp, vp, i64, C.c_int, C.c_int]),
    "sclens_hip_dev_gemm_f32": (C.c_int, [vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, C.c_float, C.c_float, C.c_int, C.c_int, vp]),
    "sclens_hip_dev_gram_f32": (C.c_int, [vp, vp, i64, i64, i64, C.c_float, vp, i64]),
    "sclens_hip_dev_sy2sb_f32": (C.c_int, [vp, vp, i64, i64, vp, C.POINTER(C.c_int)]),
    "sclens_hip_dev_sbr_apply_q1_f32": (C.c_int, [vp, vp, i64, i64, vp, vp, i64, i64]),
    "sclens_hip_dev_sbr_apply_q2_f32": (C.c_int, [vp, i64, vp, i64, i64]),
    "sclens_hip_dev_sb2st_f32": (C.c_int, [vp, vp, i64, i64, vp, vp]),
    "sclens_hip_dev_sytrd_f32": (C.c_int, [vp, vp, i64, i64, vp, vp, vp]),
    "sclens_hip_dev_stebz_f64": (C.c_int, [vp, vp, vp, i64, vp]),
    "sclens_hip_dev_eigh_f32": (C.c_int, [vp, vp, i64, i64, vp, i64, i64, vp, i64]),
    "sclens_hip_dev_malloc": (vp, [vp, i64]),
    "sclens_hip_dev_free": (None, [vp, vp]),
    "sclens_hip_dev_memcpy": (C.c_int, [vp, vp, vp, i64, C.c_int]),
    "sclens_hip_dev_memset": (C.c_int, [vp, vp, C.c_int, i64]),
    "sclens_hip_dev_sync": (C.c_int, [vp]),
}

_lib = None


class SclensHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libsclens_hip: SCLENS_ERR_{ERR_NAMES.get(code, code)}: {msg}")
        self.code = code


def load():
    """Load the shared library (no GPU needed for loading or for the host statistics)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C sclens_amd/csrc`). sclens_amd has no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def ptr(a: np.ndarray, ctype):
    return a.ctypes.data_as(C.POINTER(ctype))


class Context:
    """Owns one sclens_hip_ctx (one GPU, one stream)."""

    def __init__(self, device: int = 0):
        self.lib = load()
        # PyTorch-ROCm wheels bundle their own HIP runtime next to the system one this library links: both can live in one
        # process, but only if torch's is initialised first (measured: the other order leaves torch with "No HIP GPUs").
        # A process that has imported torch (the multi-GPU hosts do, for torch.distributed) gets that order here.
        if "torch" in sys.modules:
            try:
                sys.modules["torch"].cuda.is_available()
            except Exception:
                pass
        h = vp()
        rc = self.lib.sclens_hip_create(C.byref(h), int(device))
        if rc != 0:
            raise SclensHipError(rc, "sclens_hip_create failed (no usable HIP device?)")
        self.h = h
        self.device = int(device)
        # objects created on this context (sessions, patterns, device matrices, communicators) hold it open: close() with
        # live children only marks it, the last child's close() destroys it. (The garbage collector finalises the members
        # of a dead cycle in any order: a session must never be destroyed after its context.)
        self._children = 0
        self._deferred = False
        self._lock = threading.Lock()

    def check(self, rc):
        if rc != 0:
            raise SclensHipError(rc, self.lib.sclens_hip_last_error(self.h).decode())

    def _adopt(self):
        with self._lock:
            self._children += 1

    def _release(self):
        with self._lock:
            self._children -= 1
            last = self._children <= 0 and self._deferred
        if last:
            self.close()

    def close(self):
        if getattr(self, "h", None):
            with self._lock:
                if self._children > 0:
                    self._deferred = True
                    return
            self.lib.sclens_hip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def trim_pool(self):
        """hand the idle blocks of the library's memory pool on this device back to the driver (sclens_hip_trim)"""
        self.lib.sclens_hip_trim(self.device)

    def release_scratch(self, family: str):
        """hand one family of this context's idle scratch back to the pool: "eigensolver", "gram", "chefsi", "corr" or "all" """
        self.check(self.lib.sclens_hip_release_scratch(self.h, family.encode()))

    def pool_peak(self, reset: bool = False) -> int:
        """largest number of live (handed-out) device bytes of the library's pool on this device since the last reset"""
        return int(self.lib.sclens_hip_pool_peak(self.device, 1 if reset else 0))

    def set_option(self, name: str, value: int):
        """a named tunable of this context (include/sclens_hip.h: "precision", "two_stage", "gram_bits", ...; csrc/common.h has the table)"""
        self.check(self.lib.sclens_hip_set_option(self.h, name.encode(), int(value)))

    def get_option(self, name: str) -> int:
        v = C.c_int64(0)
        self.check(self.lib.sclens_hip_get_option(self.h, name.encode(), C.byref(v)))
        return int(v.value)

    def copy_options_from(self, other: "Context"):
        """worker contexts decompose the way the caller's context does"""
        self.check(self.lib.sclens_hip_copy_options(self.h, other.h))
        return self

    def options(self, **kw):
        """`with ctx.options(precision=0, q2_variant=14): ...` -- set, run, restore"""
        import contextlib

        @contextlib.contextmanager
        def scope():
            old = {k: self.get_option(k) for k in kw}
            try:
                for k, v in kw.items():
                    self.set_option(k, v)
                yield self
            finally:
                for k, v in old.items():
                    self.set_option(k, v)

        return scope()

    # ---- timing
    def set_timing(self, on: bool):
        self.check(self.lib.sclens_hip_set_timing(self.h, int(on)))

    def reset_timing(self):
        self.check(self.lib.sclens_hip_reset_timing(self.h))

    def timing(self, stage: str):
        ms, calls = C.c_double(0), C.c_int64(0)
        self.check(self.lib.sclens_hip_get_timing(self.h, stage.encode(), C.byref(ms), C.byref(calls)))
        return ms.value, calls.value

    # ---- raw device memory helpers (tests / bench)
    def malloc(self, nbytes: int):
        p = self.lib.sclens_hip_dev_malloc(self.h, int(nbytes))
        if not p:
            raise SclensHipError(3, "dev_malloc")
        return p

    def free(self, p):
        self.lib.sclens_hip_dev_free(self.h, p)

    def h2d(self, dst, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        self.check(self.lib.sclens_hip_dev_memcpy(self.h, dst, arr.ctypes.data, arr.nbytes, 1))

    def d2h(self, arr: np.ndarray, src):
        assert arr.flags.c_contiguous
        self.check(self.lib.sclens_hip_dev_memcpy(self.h, arr.ctypes.data, src, arr.nbytes, 2))

    def memset(self, dst, value, nbytes):
        self.check(self.lib.sclens_hip_dev_memset(self.h, dst, int(value), int(nbytes)))

    def sync(self):
        self.check(self.lib.sclens_hip_dev_sync(self.h))


class Comm:
    """sclens_hip_comm: this rank's RCCL communicator, created and used inside the library (csrc/comm.hip). The host only
    ships the 128-byte unique id from rank 0 to the other ranks (`ship`: bytes-or-None -> bytes, any channel)."""

    ID_BYTES = 128

    def __init__(self, ctx: Context, rank: int, world: int, ship):
        self.ctx, self.lib = ctx, ctx.lib
        uid = np.zeros(self.ID_BYTES, dtype=np.uint8)
        payload = None
        if rank == 0:
            try:
                ctx.check(self.lib.sclens_hip_comm_unique_id(ctx.h, ptr(uid, C.c_uint8)))
                payload = uid.tobytes()
            except Exception as e:  # the other ranks are waiting for the id: `ship` delivers the failure and raises everywhere
                payload = e
        got = ship(payload)
        uid = np.frombuffer(got, dtype=np.uint8).copy()
        h = vp()
        ctx.check(self.lib.sclens_hip_comm_create(ctx.h, ptr(uid, C.c_uint8), int(rank), int(world), C.byref(h)))
        self.h = h
        self.ctx._adopt()
        w, r, v = C.c_int(0), C.c_int(0), C.c_int(0)
        self.check(self.lib.sclens_hip_comm_info(self.h, C.byref(w), C.byref(r), C.byref(v)))
        self.world, self.rank, self.rccl_version = w.value, r.value, v.value  # as RCCL reports them
        if (self.world, self.rank) != (int(world), int(rank)):
            raise SclensHipError(4, f"RCCL reports rank {self.rank} of {self.world}, expected {rank} of {world}")

    def check(self, rc):
        if rc != 0:
            raise SclensHipError(rc, self.lib.sclens_hip_comm_last_error(self.h).decode())

    def reducer(self):
        """(function pointer, user pointer) for Session.create_sharded: the all-reduce of a row-sharded session is then an
        RCCL call made by the library itself"""
        fn = C.cast(self.lib.sclens_hip_comm_allreduce_cb, ALLREDUCE_FN)
        return fn, self.h

    def reducer_to(self):
        """(function pointer, user pointer) of the sum onto ONE rank (ncclReduce): Session.set_reduce_to"""
        return C.cast(self.lib.sclens_hip_comm_reduce_cb, REDUCE_FN), self.h

    def allreduce(self, dev_ptr: int, count: int, dtype: int):
        self.check(self.lib.sclens_hip_comm_allreduce(self.h, vp(dev_ptr), int(count), int(dtype)))

    def broadcast(self, dev_ptr: int, nbytes: int, root: int):
        self.check(self.lib.sclens_hip_comm_broadcast(self.h, vp(dev_ptr), int(nbytes), int(root)))

    def allgather(self, send_ptr: int, recv_ptr: int, nbytes: int):
        self.check(self.lib.sclens_hip_comm_allgather(self.h, vp(send_ptr), vp(recv_ptr), int(nbytes)))

    def allgather_host(self, arr: np.ndarray) -> np.ndarray:
        arr = np.ascontiguousarray(arr)
        out = np.empty((self.world,) + arr.shape, dtype=arr.dtype)
        self.check(self.lib.sclens_hip_comm_allgather_host(self.h, arr.ctypes.data, out.ctypes.data, arr.nbytes))
        return out

    def broadcast_host(self, arr: np.ndarray, root: int) -> np.ndarray:
        buf = np.ascontiguousarray(arr).copy()
        self.check(self.lib.sclens_hip_comm_broadcast_host(self.h, buf.ctypes.data, buf.nbytes, int(root)))
        return buf

    def stats(self):
        calls, nbytes = C.c_int64(0), C.c_double(0)
        self.check(self.lib.sclens_hip_comm_stats(self.h, C.byref(calls), C.byref(nbytes)))
        return {"calls": calls.value, "bytes": nbytes.value}

    def close(self):
        if getattr(self, "h", None):
            self.lib.sclens_hip_comm_destroy(self.h)
            self.h = None
            self.ctx._release()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
