"""ctypes loader for libsumcheck_hip.so -- the C ABI declared in include/sumcheck_hip.h.

There is no CPU fallback: if the shared library is missing this raises, and on a machine without a
HIP device every compute entry point returns SC_ERR_HIP (surfaced as SumcheckError).
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# SC_LIB_VARIANT=exp (tests/test_gpu_variants.py only): the -DSC_EXPERIMENTS build with the cross-check kernels and their knobs
SO_PATH = os.path.join(HERE, "libsumcheck_hip_exp.so" if os.environ.get("SC_LIB_VARIANT") == "exp" else "libsumcheck_hip.so")
if os.environ.get("SC_LIB_PATH"):  # an explicit library file (A/B runs of two builds on one box: tools/ab.sh)
    SO_PATH = os.environ["SC_LIB_PATH"]

u64p = C.POINTER(C.c_uint64)
u32p = C.POINTER(C.c_uint32)
u8p = C.POINTER(C.c_uint8)

SC_OK = 0
SC_ERR_CONSTANT_POLY = 1
SC_ERR_FIRST_ROUND_HAS_MSG = 2
SC_ERR_MISSING_MSG = 3
SC_ERR_NOT_ACTIVE = 4
SC_ERR_BAD_ARG = 5
SC_ERR_HIP = 6
SC_ERR_OOM = 7
SC_ERR_REJECT = 8

SC_TABLES_ON_DEVICE = 1
SC_TABLES_BORROW = 2
SC_TABLES_STREAM = 4
SC_NO_DEVICE_POLLING = 8


class PolyDesc(C.Structure):
    _fields_ = [
        ("num_vars", C.c_uint32),
        ("max_multiplicands", C.c_uint32),
        ("n_products", C.c_uint32),
        ("coeffs", u64p),
        ("prod_offsets", u32p),
        ("prod_indices", u32p),
        ("n_tables", C.c_uint32),
        ("tables", C.POINTER(C.c_void_p)),
        ("flags", C.c_uint32),
    ]


# every symbol include/sumcheck_hip.h declares: (restype, argtypes)
_V = C.c_void_p
SIGNATURES = {
    "sc_abi_version": (C.c_int, []),
    "sc_last_error": (C.c_char_p, []),
    "sc_device_count": (C.c_int, []),
    "sc_set_device": (C.c_int, [C.c_int]),
    "sc_prover_init": (C.c_int, [C.POINTER(PolyDesc), C.POINTER(_V)]),
    "sc_prover_init_streamed": (C.c_int, [C.POINTER(PolyDesc), C.c_uint32, C.POINTER(_V)]),
    "sc_prove_round": (C.c_int, [_V, _V, _V]),
    "sc_prover_push_randomness": (C.c_int, [_V, _V]),
    "sc_prover_state": (C.c_int, [_V, _V, u32p, _V, u32p]),
    "sc_prover_free": (None, [_V]),
    "sc_prover_set_stream": (C.c_int, [_V, _V, C.c_int]),
    "sc_prover_set_polling": (C.c_int, [_V, C.c_int]),
    "sc_prover_set_resident": (C.c_int, [_V, C.c_uint32]),
    "sc_set_cache_limit": (C.c_int, [C.c_uint64]),
    "sc_library_stats": (C.c_int, [u64p, C.c_uint32]),
    "sc_set_policy": (C.c_int, [C.c_char_p, C.c_int64]),
    "sc_get_policy": (C.c_int, [C.c_char_p, C.POINTER(C.c_int64)]),
    "sc_plan_count": (C.c_uint32, []),
    "sc_plan_name": (C.c_char_p, [C.c_uint32]),
    "sc_plan_stats": (C.c_int, [u64p, C.c_uint32]),
    "sc_prove_round_partial": (C.c_int, [_V, _V, _V]),
    "sc_wide_reduce": (C.c_int, [_V, C.c_uint32, _V]),
    "sc_prover_bind_final": (C.c_int, [_V, _V, _V]),
    "sc_comm_unique_id": (C.c_int, [_V]),
    "sc_comm_init": (C.c_int, [_V, C.c_int, C.c_int, C.POINTER(_V)]),
    "sc_comm_init_host": (C.c_int, [C.c_int, C.c_int, _V, _V, _V, C.POINTER(_V)]),
    "sc_comm_init_p2p": (C.c_int, [C.c_uint64, C.c_int, C.c_int, C.POINTER(_V)]),
    "sc_comm_selftest": (C.c_int, [_V]),
    "sc_comm_free": (None, [_V]),
    "sc_comm_info": (C.c_int, [_V, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "sc_comm_exchange_bench": (C.c_int, [_V, C.c_uint32, C.c_uint32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "sc_set_publish_timeout_ms": (C.c_int, [C.c_uint32]),
    "sc_ml_prove_sharded": (C.c_int, [_V, _V, _V, C.c_uint32, _V, _V]),
    "sc_ml_prove_sharded_rounds": (C.c_int, [_V, _V, _V, C.c_uint32, C.c_uint32, _V, _V]),
    "sc_fix_variables": (C.c_int, [_V, C.c_uint32, _V, C.c_uint32, _V, C.c_uint32]),
    "sc_poly_evaluate": (C.c_int, [_V, _V, _V, _V]),
    "sc_ml_prove_rounds": (C.c_int, [_V, _V, C.c_uint32, _V, _V]),
    "sc_release_caches": (C.c_int, []),
    "sc_sparse_evaluate": (C.c_int, [_V, _V, C.c_uint64, C.c_uint32, _V, _V]),
    "sc_rng_setup": (_V, []),
    "sc_rng_free": (None, [_V]),
    "sc_rng_feed_bytes": (None, [_V, C.c_char_p, C.c_size_t]),
    "sc_rng_fill_bytes": (None, [_V, _V, C.c_size_t]),
    "sc_rng_feed_poly_info": (None, [_V, C.c_uint64, C.c_uint64]),
    "sc_rng_feed_prover_msg": (None, [_V, _V, C.c_uint32]),
    "sc_rng_sample_fr": (None, [_V, _V]),
    "sc_ml_prove": (C.c_int, [C.POINTER(PolyDesc), _V, _V, C.POINTER(_V)]),
    "sc_ml_prove_handle": (C.c_int, [_V, _V, _V]),
    "sc_interpolate_uni_poly": (C.c_int, [_V, C.c_uint32, _V, _V]),
    "sc_ml_verify": (C.c_int, [C.c_uint32, C.c_uint32, _V, _V, C.c_uint64, _V, _V, _V]),
    "sc_gkr_phase_one": (C.c_int, [_V, _V, C.c_uint64, C.c_uint32, _V, _V, C.c_uint32, _V, _V, _V, u64p]),
    "sc_gkr_phase_two": (C.c_int, [_V, _V, C.c_uint64, C.c_uint32, _V, C.c_uint32, _V]),
    "sc_gkr_prove": (C.c_int, [_V, _V, _V, C.c_uint64, C.c_uint32, _V, _V, _V, C.c_uint32, _V, _V]),
    "sc_dense_scale": (C.c_int, [_V, C.c_uint64, _V, _V, C.c_uint32]),
    "sc_gkr_phase_one_sharded": (C.c_int, [_V, _V, _V, C.c_uint64, C.c_uint32, _V, _V, C.c_uint32, _V, _V, _V, _V, u64p]),
    "sc_gkr_phase_two_sharded": (C.c_int, [_V, _V, _V, C.c_uint64, C.c_uint32, _V, C.c_uint32, _V, _V]),
    "sc_gkr_prove_sharded": (C.c_int, [_V, _V, _V, _V, C.c_uint64, C.c_uint32, _V, _V, _V, C.c_uint32, _V, _V]),
    "sc_wide_reduce_table": (C.c_int, [_V, C.c_uint64, _V, C.c_uint32]),
    "sc_synth_table_device": (C.c_int, [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, _V]),
    "sc_prover_last_round_ms": (C.c_int, [_V, C.POINTER(C.c_float)]),
    "sc_prover_set_timing": (C.c_int, [_V, C.c_int]),
    "sc_prover_get_timing": (C.c_int, [_V, C.POINTER(C.c_double), u64p, C.POINTER(C.c_double)]),
    "sc_prover_get_round_timing": (C.c_int, [_V, C.POINTER(C.c_double), u64p]),
    "sc_prover_reset": (C.c_int, [_V, _V, C.c_uint32]),
    "sc_claim_weights": (C.c_int, [C.c_uint32, _V, _V]),
    "sc_fr_elementwise": (C.c_int, [C.c_int, _V, _V, _V, C.c_uint64]),
    "sc_bench_modmul": (C.c_int, [C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_float), u64p]),
}

ABI_VERSION = 5  # SC_ABI_VERSION of include/sumcheck_hip.h as declared above
_lib = None


def _missing_symbol(name):
    def stub(*_a, **_k):
        raise SumcheckError(SC_ERR_BAD_ARG, f"{SO_PATH} does not export {name} (an older build loaded with SC_AB_ALLOW_MISSING=1)")
    return stub


class SumcheckError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"[sc_status {code}] {msg}")
        self.code = code
        self.msg = msg


def lib():
    """Load the HIP library (raises if it has not been built: there is no fallback path)."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise ImportError(
                f"{SO_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(sumcheck_amd has no CPU fallback)")
        # PyTorch-ROCm ships its own libamdhip64.so; two HIP runtimes in one process do not see each other's devices or
        # allocations.  Loading torch first makes the dynamic loader resolve our libamdhip64.so.N to the copy torch
        # already mapped, so tensors' device pointers are valid inside the library.  (Pure C / Rust callers never load
        # torch and use the system runtime.)
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        L = C.CDLL(SO_PATH)
        # An A/B run against an OLDER build (tools/ab.sh) sets SC_AB_ALLOW_MISSING=1 next to SC_LIB_PATH: entry points that build lacks
        # become stubs that raise when called.  Without it every declared symbol must resolve, whatever file SC_LIB_PATH names.
        allow_missing = os.environ.get("SC_AB_ALLOW_MISSING") == "1"
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(L, name)  # AttributeError if the .so does not export a declared symbol
            except AttributeError:
                if not allow_missing:
                    raise
                setattr(L, name, _missing_symbol(name))
                continue
            fn.restype = res
            fn.argtypes = args
        got = L.sc_abi_version()
        if got != ABI_VERSION:
            msg = f"{SO_PATH} reports ABI version {got}, this loader declares {ABI_VERSION}: signatures may not match"
            if not allow_missing:
                raise ImportError(msg + " (rebuild: python -c 'import __graft_entry__ as g; g.build()')")
            import warnings
            warnings.warn(msg)
        _lib = L
    return _lib


def check(rc: int):
    if rc != SC_OK:
        raise SumcheckError(rc, lib().sc_last_error().decode("utf-8", "replace"))


def set_policy(key: str, value: int) -> None:
    """sc_set_policy: process-wide library policy (include/sumcheck_hip.h lists the keys)"""
    check(lib().sc_set_policy(key.encode(), int(value)))


def get_policy(key: str) -> int:
    v = C.c_int64()
    check(lib().sc_get_policy(key.encode(), C.byref(v)))
    return int(v.value)


class policy:
    """with _lib.policy(tail_slices=0): ...  -- set for the block, restored afterwards (tests, A/B runs)"""

    def __init__(self, **kv):
        self.kv, self.old = kv, {}

    def __enter__(self):
        for k, v in self.kv.items():
            self.old[k] = get_policy(k)
            set_policy(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            set_policy(k, v)
        return False


def plan_stats() -> dict:
    """sc_plan_stats as {plan name: launches so far in this process}"""
    L = lib()
    n = L.sc_plan_count()
    out = (C.c_uint64 * n)()
    check(L.sc_plan_stats(out, n))
    return {L.sc_plan_name(i).decode(): int(out[i]) for i in range(n)}
