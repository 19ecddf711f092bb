"""quick GPU check of K1h + K1f against the oracle (development aid)"""
import sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch
import ntcard_amd as nt, orc

def tile_array(arr):
    n, L = arr.shape
    C16, ntl = (L + 15) // 16, (n + 2047) // 2048
    a = np.full((ntl * 2048, C16 * 16), ord("A"), dtype=np.uint8)
    a[:n, :L] = arr
    return np.ascontiguousarray(a.reshape(ntl, 2048, C16, 16).transpose(0, 2, 1, 3)).reshape(-1)

def run(arr, k, r_bits, flags):
    n, L = arr.shape
    t = torch.from_numpy(tile_array(arr)).cuda()
    with nt.Engine([k], r_bits=r_bits, s_bits=7, flags=flags | nt.FLAG_REQUIRE_TILED) as e:
        e.submit_tiled_device(t.data_ptr(), n, L)
        return e.finish(counters=True)

def check(n, L, k=32, p_bad=0.0, r_bits=18, seed=1):
    rng = np.random.default_rng(seed)
    alpha = np.frombuffer(b"ACGTacgtUuNnRYKM.-*", dtype=np.uint8)
    arr = alpha[rng.integers(0, 4, size=(n, L))]
    if p_bad:
        arr = np.where(rng.random((n, L)) < p_bad, alpha[rng.integers(4, len(alpha), size=(n, L))], arr).astype(np.uint8)
    tc, ph, f1 = run(arr, k, r_bits, 0)
    counters = np.zeros((1, 2, 1 << r_bits), dtype=np.uint16)
    offs = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
    of1 = orc.sketch_update(counters, np.ascontiguousarray(arr).reshape(-1), offs, [k], 0, r_bits, 7)
    ok = bool(np.array_equal(f1, of1) and np.array_equal(tc, counters))
    print(f"n={n} L={L} k={k} bad={p_bad}: f1 {int(f1[0])} vs {int(of1[0])}, counters differ at {int((tc != counters).sum())} -> {'OK' if ok else 'FAIL'}", flush=True)
    return ok

if __name__ == "__main__":
    allok = True
    for args in [(2048, 40, 32), (5000, 150, 32, 0.002), (1, 150), (70000, 150, 32, 0.001), (4097, 47, 32, 0.02), (3_200_000, 33, 32, 0.01), (1_000_000, 150, 32, 0.0005)]:
        allok &= check(*args)
    print("ALL OK" if allok else "SOME FAILED")
