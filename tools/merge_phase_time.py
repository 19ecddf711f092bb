"""tools/merge_phase_time.py — the on-device phases of the one-process-per-GPU merge (parallel.merge_to_value_histograms) on ONE MI355X, at the
geometry of N = 2, 4, 8 ranks and rBits = 27: narrow the uint32 sketch to uint16 (ntc_narrow_u16_device), add the N received slices
(ntc_sum_slices_u16_device), value histogram of the summed slice (ntc_value_hist_u16_device).  The exchange itself (N - 1 slices out and in over
point-to-point xGMI links) cannot be measured on a one-GPU box; DESIGN §7 prices it."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ntcard_amd as nt  # noqa: E402,F401
from ntcard_amd import engine as E  # noqa: E402

dev = torch.device("cuda:0")
n = 2 << 27  # counters of one k at rBits = 27
sketch = torch.randint(0, 4, (n,), dtype=torch.int32, device=dev)
u16 = torch.empty(n, dtype=torch.int16, device=dev)
hist = torch.zeros(2 * 65536, dtype=torch.int32, device=dev)


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


st = torch.cuda.current_stream().cuda_stream
t_n = timed(lambda: E.narrow_u16_device(sketch.data_ptr(), n, u16.data_ptr(), 0, st))
print("narrow 2^28 uint32 -> uint16: %.3f ms (%.2f TB/s of 1.5 GiB)" % (t_n, 1.5 * 2**30 / t_n / 1e9))
for N in (2, 4, 8):
    ln = n // N
    buf = torch.randint(0, 4, (N * ln,), dtype=torch.int16, device=dev)  # (counters of a 200 M-read run: small values, which K2 counts in LDS)
    t_s = timed(lambda: E.sum_slices_u16_device(buf.data_ptr(), ln, N, ln, 0, st))
    t_h = timed(lambda: E.value_hist_u16_device(buf.data_ptr(), ln, hist.data_ptr(), 0, st))
    mib = ln * 2 / 2**20
    print("N = %d: slice %4.0f MiB; sum of %d slices %.3f ms; value histogram of the slice %.3f ms; exchange = %d slices out + in per GPU, one per link" % (N, mib, N, t_s, t_h, N - 1))
