// ntc_tile_bits.hpp — byte-level helpers shared by the kernels that read the TILED slot layout (ntc_sketch_ts.hip, ntc_sketch_k1h.hip):
// 16 raw bytes -> 2 bits per base, and the exact mask of the bytes that are no ACGTU letter (nthash.hpp:31-64 restricted to letters).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ntc {
namespace tilebits {

__device__ __forceinline__ uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbit(hi, lo, sh); }
__device__ __forceinline__ uint32_t perm(uint32_t s0, uint32_t s1, uint32_t sel) { return __builtin_amdgcn_perm(s0, s1, sel); }
__device__ __forceinline__ uint64_t ballot(bool b) { return __builtin_amdgcn_ballot_w64(b); }
__device__ __forceinline__ uint32_t bfi(uint32_t m, uint32_t x, uint32_t y) { return (x & m) | (y & ~m); } // v_bfi_b32
__device__ __forceinline__ uint32_t mbcnt(uint64_t m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }
__device__ __forceinline__ uint32_t rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

typedef uint32_t v4u32 __attribute__((ext_vector_type(4)));

// v_perm table indexed by (byte & 7): the only letter each index may stand for (nthash.hpp:16,32 trick), 0xff: none
constexpr uint32_t kExpS0 = 0x47ff5554u; // idx 7:'G' 6:- 5:'U' 4:'T'
constexpr uint32_t kExpS1 = 0x43ff41ffu; // idx 3:'C' 2:- 1:'A' 0:-

// 16 raw bytes -> 32 bits (2 per base, code2 = (ascii >> 1) & 3: A=0 C=1 T/U=2 G=3, base q in bits 2q+1:2q);
// bad != 0 iff some byte is not ACGTU/acgtu
__device__ __forceinline__ uint32_t pack16(const v4u32 v, uint32_t& bad)
{
	// code2 of 4 bytes lands in the top byte of (w & 0x06060606) * 0x00820820 (fields 2 bits wide, no carries)
	const uint32_t p0 = (v.x & 0x06060606u) * 0x00820820u, p1 = (v.y & 0x06060606u) * 0x00820820u;
	const uint32_t p2 = (v.z & 0x06060606u) * 0x00820820u, p3 = (v.w & 0x06060606u) * 0x00820820u;
	const uint32_t lo = perm(p1, p0, 0x0c0c0703u);
	const uint32_t hi = perm(p3, p2, 0x07030c0cu);
	// the letter (byte & 7) may stand for, XORed with the byte: zero (or the case bit) for a base letter
	uint32_t x = perm(kExpS0, kExpS1, v.x & 0x07070707u) ^ v.x;
	x = (uint32_t)__builtin_amdgcn_bitop3_b32(perm(kExpS0, kExpS1, v.y & 0x07070707u), v.y, x, 0xbe); // (a ^ b) | c
	x = (uint32_t)__builtin_amdgcn_bitop3_b32(perm(kExpS0, kExpS1, v.z & 0x07070707u), v.z, x, 0xbe);
	x = (uint32_t)__builtin_amdgcn_bitop3_b32(perm(kExpS0, kExpS1, v.w & 0x07070707u), v.w, x, 0xbe);
	bad = x & 0xdfdfdfdfu;
	return lo | hi;
}

// exact 16-bit mask of the non-ACGTU bytes of a 16-byte piece (bit q = byte q); rare paths only
__device__ __forceinline__ uint32_t inv4(uint32_t v)
{
	uint32_t x = (perm(kExpS0, kExpS1, v & 0x07070707u) ^ v) & 0xdfdfdfdfu;
	x |= x >> 4;
	x |= x >> 2;
	x |= x >> 1;
	x &= 0x01010101u;
	return ((x * 0x01020408u) >> 24) & 0xfu;
}
__device__ __forceinline__ uint32_t inv16(const v4u32 v) { return inv4(v.x) | (inv4(v.y) << 4) | (inv4(v.z) << 8) | (inv4(v.w) << 12); }


} // namespace tilebits
} // namespace ntc
