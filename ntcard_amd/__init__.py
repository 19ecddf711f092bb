"""ntcard_amd — MI355X (gfx950) engine for ntCard's ntHash -> sample -> count hot path.

Product code lives in ntcard_amd/csrc (HIP kernels + the C ABI of include/ntcard_hip.h); this
package is the thin Python mirror of that ABI used by tests and bench.py.
"""
from ._abi import ABI_SYMBOLS, LIB_PATH, NtcError  # noqa: F401
from .engine import (FLAG_ALWAYS_LOG, FLAG_PARTITION_ALWAYS, FLAG_LANE_KERNEL, FLAG_DIRECT_ATOMICS, FLAG_REQUIRE_TILED, FLAG_DEFER_REDO, FLAG_SIMPLE_KERNEL, Engine, HllEngine, hll_estimate, estimate, merge_devices, gen_reads_device, gen_reads_tiled_device, tiled_bytes, tile_reads, tile_reads_ragged, hash_dump_device, s_bits_for_input, value_hist_device, narrow_u16_device, sum_slices_u16_device, value_hist_u16_device,  # noqa: F401
                     write_hist)
