"""ctypes binding of ntcard_amd/lib/libntcard_hip.so (include/ntcard_hip.h).

There is no CPU fallback: loading fails loudly if the library was not built, and every compute
entry point fails with NTC_ERR_DEVICE when no HIP device is present.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libntcard_hip.so")

# every symbol include/ntcard_hip.h declares
ABI_SYMBOLS = [
    "ntc_abi_version", "ntc_max_k", "ntc_last_error", "ntc_create", "ntc_destroy", "ntc_reset",
    "ntc_submit", "ntc_submit_spans", "ntc_submit_device", "ntc_sync", "ntc_finish", "ntc_device_state",
    "ntc_hash_dump_device", "ntc_hash_dump_k1_device", "ntc_gen_reads_device", "ntc_estimate", "ntc_write_hist",
    "ntc_kernel_time", "ntc_apply_time", "ntc_fixup_time", "ntc_merge_allocations", "ntc_update_mode", "ntc_flush", "ntc_set_profiling", "ntc_merge_counters", "ntc_merge_devices", "ntc_value_hist_device", "ntc_hll_create", "ntc_hll_finish", "ntc_hll_estimate",
    "ntc_submit_tiled_device", "ntc_submit_tiled_ragged_device", "ntc_submit_tiled_bins_device", "ntc_tiled_bytes", "ntc_gen_reads_tiled_device",
    "ntc_narrow_u16_device", "ntc_sum_slices_u16_device", "ntc_value_hist_u16_device",
    "ntc_log_export_device", "ntc_log_replace_device",
]


class NtcConfig(C.Structure):
    _fields_ = [
        ("n_k", C.c_uint32),
        ("k", C.POINTER(C.c_uint32)),
        ("gap", C.c_uint32),
        ("r_bits", C.c_uint32),
        ("s_bits", C.c_uint32),
        ("device", C.c_int32),
        ("stream", C.c_void_p),
        ("ext_sketch", C.c_void_p),
        ("ext_f1", C.c_void_p),
        ("flags", C.c_uint32),
        ("log_entries", C.c_uint64),
    ]


class NtcError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"ntCard: {msg} (status {code})")
        self.code = code


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C ntcard_amd/csrc`). The HIP extension is mandatory; there is no CPU fallback.")
    # PyTorch-ROCm bundles its own HIP runtime; if this process uses torch on the GPU as well, torch has to bring its
    # runtime up before ours touches the device (the other order leaves torch without visible GPUs).
    import sys
    torch = sys.modules.get("torch")
    if torch is not None and torch.cuda.is_available():
        torch.cuda.init()
    L = C.CDLL(LIB_PATH)
    u32, u64, i32, p = C.c_uint32, C.c_uint64, C.c_int32, C.c_void_p
    L.ntc_abi_version.restype = u32
    L.ntc_max_k.restype = u32
    L.ntc_last_error.restype = C.c_char_p
    L.ntc_create.argtypes = [C.POINTER(NtcConfig), C.POINTER(p)]
    L.ntc_destroy.argtypes = [p]
    L.ntc_destroy.restype = None
    L.ntc_reset.argtypes = [p]
    L.ntc_submit.argtypes = [p, p, p, u64]
    L.ntc_submit_spans.argtypes = [p, p, p, p, u64]
    L.ntc_submit_device.argtypes = [p, p, u64, u32, u32]
    L.ntc_submit_tiled_device.argtypes = [p, p, u64, u32]
    L.ntc_submit_tiled_ragged_device.argtypes = [p, p, u64, u32, p]
    L.ntc_submit_tiled_bins_device.argtypes = [p, u32, p, p, p, p]
    L.ntc_tiled_bytes.argtypes = [u64, u32]
    L.ntc_tiled_bytes.restype = u64
    L.ntc_gen_reads_tiled_device.argtypes = [i32, p, p, u64, u64, u64, u32, u32, u64]
    L.ntc_sync.argtypes = [p]
    L.ntc_finish.argtypes = [p, p, p, p]
    L.ntc_merge_counters.argtypes = [p, p, p]
    L.ntc_merge_devices.argtypes = [C.POINTER(p), i32]
    L.ntc_value_hist_device.argtypes = [i32, p, p, u64, p]
    L.ntc_narrow_u16_device.argtypes = [i32, p, p, u64, p]
    L.ntc_sum_slices_u16_device.argtypes = [i32, p, p, u64, u32, u64]
    L.ntc_value_hist_u16_device.argtypes = [i32, p, p, u64, p]
    L.ntc_device_state.argtypes = [p, C.POINTER(p), C.POINTER(u64), C.POINTER(p)]
    L.ntc_log_export_device.argtypes = [p, u32, p, C.POINTER(u64), C.POINTER(u64)]
    L.ntc_log_replace_device.argtypes = [p, p, u64]
    L.ntc_hash_dump_device.argtypes = [i32, p, p, u64, u32, u32, u32, u32, u32, p, p]
    L.ntc_hash_dump_k1_device.argtypes = [i32, p, p, u64, u32, u32, u32, u32, u32, p, p]
    L.ntc_gen_reads_device.argtypes = [i32, p, p, u64, u64, u64, u32, u32, u32, u64]
    L.ntc_estimate.argtypes = [p, u32, u32, u32, C.POINTER(C.c_double), p]
    L.ntc_write_hist.argtypes = [C.c_char_p, u64, C.c_double, p, u32]
    L.ntc_kernel_time.argtypes = [p, C.POINTER(C.c_double), C.POINTER(u64)]
    L.ntc_apply_time.argtypes = [p, C.POINTER(C.c_double), C.POINTER(u64)]
    L.ntc_fixup_time.argtypes = [p, C.POINTER(C.c_double)]
    L.ntc_merge_allocations.argtypes = [p, C.POINTER(u64)]
    L.ntc_flush.argtypes = [p]
    L.ntc_update_mode.argtypes = [p, C.POINTER(u32)]
    L.ntc_set_profiling.argtypes = [p, C.c_int]
    L.ntc_hll_create.argtypes = [u32, u32, i32, p, C.POINTER(p)]
    L.ntc_hll_finish.argtypes = [p, p, p]
    L.ntc_hll_estimate.argtypes = [p, u32, C.POINTER(C.c_double)]
    for name in ABI_SYMBOLS:
        fn = getattr(L, name)
        if name not in ("ntc_abi_version", "ntc_max_k", "ntc_last_error", "ntc_destroy", "ntc_tiled_bytes"):
            fn.restype = C.c_int
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise NtcError(rc, lib().ntc_last_error().decode("utf-8", "replace"))
