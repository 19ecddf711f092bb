#!/usr/bin/env python3
"""Generates tools/ubench4.hip: issue cost of a LONE wave's instructions on gfx950 with explicit physical registers
(K1h runs 1 or 2 waves per SIMD, so what a single wave can issue per clock is what bounds it).  Each probe is a loop over
an unrolled body in one asm statement; clocks by s_memtime inside the kernel (wave 0) and by HIP events outside.
Questions: does the VGPR bank (register number mod 4) of the three sources of v_bitop3_b32 matter?  what do the
non-bit-op instructions of the pack / push / pass code cost next to a full-rate op?  is a scalar instruction between two
vector ones free?"""
import random

random.seed(7)
N = 96           # instructions per loop body
ITERS = 4000


def bank_regs(bank, lo=8, hi=120):
    return [r for r in range(lo, hi) if r % 4 == bank]


def body_bitop3(pattern):
    """pattern: banks of (dst, s0, s1, s2) relative to a rotating base"""
    out = []
    for i in range(N):
        base = i % 4
        d, a, b, c = [(base + x) % 4 for x in pattern]
        rd = bank_regs(d)[(i * 5) % 28]
        ra = bank_regs(a)[(i * 7 + 1) % 28]
        rb = bank_regs(b)[(i * 11 + 2) % 28]
        rc = bank_regs(c)[(i * 13 + 3) % 28]
        if len({ra, rb, rc}) < 3:  # keep three DISTINCT registers
            rc = bank_regs(c)[(i * 13 + 4) % 28]
            if len({ra, rb, rc}) < 3:
                rb = bank_regs(b)[(i * 11 + 9) % 28]
        out.append(f"v_bitop3_b32 v{rd}, v{ra}, v{rb}, v{rc} bitop3:0x96")
    return out


def body_walk(fbase, rbase, tbase, stride=1):
    """the shape of K1h's walk: F'[j] = F[j-1] ^ x ^ y in place from the top, R'[j] = R[j+1] ^ x ^ y from the bottom; x, y from a pool of 16"""
    out = []
    F = [fbase + stride * j for j in range(31)]
    R = [rbase + stride * j for j in range(31)]
    pool = [tbase + i for i in range(16)]
    rnd = random.Random(3)
    for j in range(30, 0, -1):
        x, y = rnd.sample(pool, 2)
        out.append(f"v_bitop3_b32 v{F[j]}, v{F[j-1]}, v{x}, v{y} bitop3:0x96")
    for j in range(0, 30):
        x, y = rnd.sample(pool, 2)
        out.append(f"v_bitop3_b32 v{R[j]}, v{R[j+1]}, v{x}, v{y} bitop3:0x96")
    return out


def body_walk_banked(split):
    """the same walk with the state and the function planes kept in different banks: F in banks {0,1} (registers 4i, 4i+1), R in banks {2,3};
    the function planes an F update reads sit in banks 2 and 3, those of an R update in banks 0 and 1"""
    F = [r for r in range(8, 200) if r % 4 in (0, 1)][:31]
    R = [r for r in range(8, 200) if r % 4 in (2, 3)][:31]
    pf_x = [r for r in range(140, 200) if r % 4 == 2][:8]
    pf_y = [r for r in range(140, 200) if r % 4 == 3][:8]
    pr_x = [r for r in range(140, 200) if r % 4 == 0][:8]
    pr_y = [r for r in range(140, 200) if r % 4 == 1][:8]
    rnd = random.Random(3)
    out = []
    for j in range(30, 0, -1):
        out.append(f"v_bitop3_b32 v{F[j]}, v{F[j-1]}, v{rnd.choice(pf_x)}, v{rnd.choice(pf_y)} bitop3:0x96")
    for j in range(0, 30):
        out.append(f"v_bitop3_b32 v{R[j]}, v{R[j+1]}, v{rnd.choice(pr_x)}, v{rnd.choice(pr_y)} bitop3:0x96")
    return out


def body_simple(fmt, n=N, nd=24):
    return [fmt.format(d=8 + (i % nd) * 2, a=60 + (i * 3) % 40, b=101 + (i * 7) % 20, c=124 + (i * 5) % 20) for i in range(n)]


PROBES = [
    ("bitop3: 3 sources in 3 different banks, dst in the 4th", body_bitop3((0, 1, 2, 3))),
    ("bitop3: 3 sources in ONE bank", body_bitop3((0, 1, 1, 1))),
    ("bitop3: two sources share a bank", body_bitop3((0, 1, 1, 2))),
    ("bitop3: dst shares a bank with a source", body_bitop3((1, 1, 2, 3))),
    ("walk shape, contiguous state (K1h today)", body_walk(18, 49, 212)),
    ("walk shape, banked state / function planes", body_walk_banked(True)),
    ("v_xor_b32 (2 sources)", body_simple("v_xor_b32 v{d}, v{a}, v{b}")),
    ("v_and_b32 with a 32-bit literal", body_simple("v_and_b32 v{d}, 0x06060606, v{a}")),
    ("v_mul_lo_u32", body_simple("v_mul_lo_u32 v{d}, v{a}, v{b}")),
    ("v_mul_u32_u24", body_simple("v_mul_u32_u24 v{d}, v{a}, v{b}")),
    ("v_perm_b32 (3 VGPRs)", body_simple("v_perm_b32 v{d}, v{a}, v{b}, v{c}")),
    ("v_perm_b32 (SGPR, VGPR, VGPR)", body_simple("v_perm_b32 v{d}, s40, v{b}, v{c}")),
    ("v_lshl_or_b32", body_simple("v_lshl_or_b32 v{d}, v{a}, 1, v{b}")),
    ("v_and_or_b32", body_simple("v_and_or_b32 v{d}, v{a}, v{b}, v{c}")),
    ("v_bfi_b32", body_simple("v_bfi_b32 v{d}, v{a}, v{b}, v{c}")),
    ("v_alignbit_b32", body_simple("v_alignbit_b32 v{d}, v{a}, v{b}, v{c}")),
    ("v_mov_b32", body_simple("v_mov_b32 v{d}, v{a}")),
    ("v_swap_b32", [f"v_swap_b32 v{8 + 2 * (i % 24)}, v{60 + (i * 3) % 40}" for i in range(N)]),
    ("v_add_co_u32 + v_addc_co_u32 pair (vcc)", sum(([f"v_add_co_u32 v{8 + 2 * (i % 24)}, vcc, -1, v{60 + (i * 3) % 40}",
                                                      f"v_addc_co_u32 v7, vcc, v7, v7, vcc"] for i in range(N // 2)), [])),
    ("v_min_u32 + v_lshl_or_b32 pair", sum(([f"v_min_u32 v{8 + 2 * (i % 24)}, 1, v{60 + (i * 3) % 40}",
                                             f"v_lshl_or_b32 v7, v7, 1, v{8 + 2 * (i % 24)}"] for i in range(N // 2)), [])),
    ("v_cmp_ne_u32 -> vcc", body_simple("v_cmp_ne_u32_e32 vcc, 0, v{a}")),
    ("v_cmp_ne_u32 -> sgpr pair", body_simple("v_cmp_ne_u32_e64 s[42:43], 0, v{a}")),
    ("v_mbcnt_lo + v_mbcnt_hi", sum(([f"v_mbcnt_lo_u32_b32 v{8 + 2 * (i % 24)}, s42, 0",
                                      f"v_mbcnt_hi_u32_b32 v{8 + 2 * (i % 24)}, s43, v{8 + 2 * (i % 24)}"] for i in range(N // 2)), [])),
    ("bitop3 with a scalar op after each", sum(([x, f"s_add_u32 s{44 + i % 4}, s{44 + i % 4}, 1"] for i, x in enumerate(body_bitop3((0, 1, 2, 3))[: N // 2])), [])),
    ("bitop3 with TWO scalar ops after each", sum(([x, f"s_add_u32 s{44 + i % 4}, s{44 + i % 4}, 1", f"s_lshl_b32 s{48 + i % 4}, s{48 + i % 4}, 1"]
                                                   for i, x in enumerate(body_bitop3((0, 1, 2, 3))[: N // 3])), [])),
    ("s_add_u32 alone", [f"s_add_u32 s{44 + i % 8}, s{44 + i % 8}, 1" for i in range(N)]),
    ("exec write + ds_write_b64 + exec restore", sum(([f"s_mov_b64 exec, s[42:43]", f"ds_write_b64 v6, v[{8 + 2 * (i % 24)}:{9 + 2 * (i % 24)}]", "s_mov_b64 exec, -1"]
                                                      for i in range(N // 3)), []) + ["s_waitcnt lgkmcnt(0)"]),
    ("ds_read_b32 (wave-wide, no conflicts) x8 + wait", sum(([f"ds_read_b32 v{8 + 2 * (i % 24)}, v6 offset:{256 * (i % 8)}" for i in range(8 * g, 8 * g + 8)] + ["s_waitcnt lgkmcnt(0)"]
                                                             for g in range(N // 9)), [])),
]

src = ['// ubench4.hip — what one wave can issue per clock on gfx950, with explicit registers (generated by tools/gen_ubench4.py)',
       '#include <hip/hip_runtime.h>', '#include <cstdio>', '#include <cstdint>', '#include <cstdlib>', '#include <vector>',
       '#define CHECK(x) do{hipError_t e=(x); if(e!=hipSuccess){fprintf(stderr,"%s: %s\\n",#x,hipGetErrorString(e)); exit(1);} }while(0)',
       f'constexpr int ITERS={ITERS};']
clob = ",".join(f'"v{i}"' for i in range(6, 232)) + "," + ",".join(f'"s{i}"' for i in range(38, 60)) + ',"vcc","memory"'
for pi, (name, body) in enumerate(PROBES):
    text = "\\n".join(body)
    src += [f'__global__ __launch_bounds__(512) void probe{pi}(unsigned long long* out){{',
            ' extern __shared__ uint32_t lds[]; lds[threadIdx.x]=threadIdx.x; __syncthreads();',
            ' unsigned long long t0, t1;',
            f' asm volatile("v_lshlrev_b32 v6, 2, %2\\ns_mov_b64 s[42:43], 0x55\\ns_mov_b32 s40, 0x03020100\\nv_mov_b32 v7, 0\\ns_memtime %0\\ns_waitcnt lgkmcnt(0)\\ns_mov_b32 s38, {ITERS}\\n'
            f'L_{pi}_%=:\\n{text}\\ns_sub_u32 s38, s38, 1\\ns_cmp_lg_u32 s38, 0\\ns_cbranch_scc1 L_{pi}_%=\\ns_memtime %1\\ns_waitcnt lgkmcnt(0)\\n"',
            f'  : "=&s"(t0), "=&s"(t1) : "v"(threadIdx.x & 63) : {clob});',
            ' if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;', ' if (lds[threadIdx.x] == 0xffffffffu) out[0] = 0;', '}']
src += ['typedef void (*kern_t)(unsigned long long*);',
        'static void run(const char* name, kern_t k, int n_instr, unsigned long long* d_out){',
        ' CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 120*1024));',
        ' for (int threads : {256, 384, 512}) {',
        '  hipEvent_t a,b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));',
        '  hipLaunchKernelGGL(k, dim3(256), dim3(threads), 120*1024, 0, d_out); CHECK(hipDeviceSynchronize());',
        '  CHECK(hipEventRecord(a)); hipLaunchKernelGGL(k, dim3(256), dim3(threads), 120*1024, 0, d_out); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));',
        '  float ms; CHECK(hipEventElapsedTime(&ms,a,b)); std::vector<unsigned long long> h(256); CHECK(hipMemcpy(h.data(), d_out, 256*8, hipMemcpyDeviceToHost));',
        '  double s=0; for (auto x: h) s += (double)x; s /= 256.0;',
        '  const double per = (double)ITERS * n_instr;',
        '  printf("%-52s %d waves/CU: %6.2f memtime-ticks/instr/wave   %7.3f ms = %5.2f ns/instr/wave\\n", name, threads/64, s/per, ms, ms*1e6/per);',
        ' }', '}',
        'int main(){ unsigned long long* d; CHECK(hipMalloc(&d, 256*8));']
for pi, (name, body) in enumerate(PROBES):
    src.append(f' run("{name}", probe{pi}, {len(body)}, d);')
src += [' return 0; }']
open('tools/ubench4.hip', 'w').write("\n".join(src) + "\n")
print("probes:", len(PROBES))
