import sys, numpy as np, torch
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import orc, ntcard_amd as nt
n, L, stride = 20_000, 150, 152
d = torch.empty(n * stride + 16, dtype=torch.uint8, device="cuda")
nt.gen_reads_device(d.data_ptr(), 9, 0, n, L, stride, 1, genome_len=300_000)
torch.cuda.synchronize()
host = d[: n * stride].cpu().numpy().reshape(n, stride)
reads = [host[i, :L].tobytes() for i in range(n)]
for klist in ([16],[24],[48],[16,24],[24,32],[16,24,32,48]):
    with nt.Engine(klist, r_bits=19, s_bits=7) as e:
        e.submit_device(d.data_ptr(), n, L, stride)
        tc, ph, f1 = e.finish(counters=True)
    oc, of1 = orc.sketch_reads(reads, klist, 0, 19, 7)
    for ki,k in enumerate(klist):
        diff = (tc[ki].astype(np.int64) - oc[ki].astype(np.int64))
        print(klist, k, "f1 ok" if f1[ki]==of1[ki] else "F1 BAD", "mismatch", np.count_nonzero(diff), "sum diff", diff.sum(), "hits", oc[ki].sum())
