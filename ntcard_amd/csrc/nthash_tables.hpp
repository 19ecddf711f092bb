// nthash_tables.hpp — host-side derivation of the per-k lookup tables the HIP kernels stage in LDS.
//
// Everything is derived from the four base seeds (nthash.hpp:25-28) and the split-rotate rule
// (nthash.hpp:186-217: low 33 bits and high 31 bits rotate independently); the reference's
// msTab33r/msTab31l tables (nthash.hpp:66-183) are exactly srol^i(seed) and are NOT copied.
//
// Device representation of one strand's 64-bit hash x = L | H<<33 (L: 33 bits, H: 31 bits):
//   lo : L[0..31]
//   B  : one bit, L[32]   (forward strand keeps it in bit 31 of a scratch register, reverse in bit 0)
//   Hd : (H << 1) | H[30] (bit 0 duplicates bit 31, so both 31-bit rotates are 2 VALU ops and
//                          unsigned compares of Hd order exactly like H)
// One table entry per (in-base, out-base) pair:
//   A[slot] = { Tf.lo, Tf.Hd, Tr.lo, Tr.Hd }   (one ds_read_b128)
//   B[slot] = bit31: Tf.L[32], bit0: Tr.L[32]  (one ds_read_b32)
// with  Tf = seed(in) ^ srol^k(seed(out))            (NTF64 roll, nthash.hpp:242-248)
//       Tr = comp(out) ^ srol^k(comp(in))            (NTR64 roll, nthash.hpp:251-257, XOR-before-rotate)
#pragma once
#include <cstdint>
#include <cstring>

namespace ntc {

constexpr uint64_t kSeed[4] = { 0x3c8bfbb395c60474ULL,   // A
	                            0x3193c18562a02b4cULL,   // C
	                            0x20323ed082572324ULL,   // G
	                            0x295549f54be24456ULL }; // T (and U)

// code: A=0 C=1 G=2 T=3; complement = 3 - code
inline uint64_t seed_of(unsigned code) { return kSeed[code & 3]; }
inline uint64_t comp_of(unsigned code) { return kSeed[3 - (code & 3)]; }

inline uint64_t srol(uint64_t x, unsigned n)
{
	const uint64_t M33 = (1ULL << 33) - 1, M31 = (1ULL << 31) - 1;
	uint64_t lo = x & M33, hi = x >> 33;
	unsigned a = n % 33u, b = n % 31u;
	if (a) lo = ((lo << a) | (lo >> (33u - a))) & M33;
	if (b) hi = ((hi << b) | (hi >> (31u - b))) & M31;
	return lo | (hi << 33);
}

inline uint32_t lo_of(uint64_t x) { return (uint32_t)x; }
inline uint32_t b32_of(uint64_t x) { return (uint32_t)((x >> 32) & 1u); }
inline uint32_t hd_of(uint64_t x)
{
	uint32_t h = (uint32_t)(x >> 33); // 31 bits
	return (h << 1) | (h >> 30);
}

// slots 0..15: in*4+out (steady state); slots 16..19: in with "no out base yet" (window filling)
constexpr int kMainSlots = 16;
constexpr int kSlots = 20;

struct alignas(16) HashTables {
	uint32_t A[kSlots][4]; // {Tf.lo, Tf.Hd, Tr.lo, Tr.Hd}
	uint32_t B[kSlots][4]; // [0] = bit31:Tf.L[32] | bit0:Tr.L[32]; [1..3] spare (gap tables reuse)
};

inline void build_tables(unsigned k, HashTables& t)
{
	std::memset(&t, 0, sizeof t);
	for (int slot = 0; slot < kSlots; ++slot) {
		unsigned in = slot < kMainSlots ? (unsigned)slot >> 2 : (unsigned)slot - kMainSlots;
		bool has_out = slot < kMainSlots;
		unsigned out = (unsigned)slot & 3;
		uint64_t tf = seed_of(in) ^ (has_out ? srol(seed_of(out), k) : 0);
		uint64_t tr = (has_out ? comp_of(out) : 0) ^ srol(comp_of(in), k);
		t.A[slot][0] = lo_of(tf);
		t.A[slot][1] = hd_of(tf);
		t.A[slot][2] = lo_of(tr);
		t.A[slot][3] = hd_of(tr);
		t.B[slot][0] = (b32_of(tf) << 31) | b32_of(tr);
	}
}

// Strand registers (device layout) of the hash of k consecutive 'A' bases — the state the tuned
// kernel starts every read from (see ntc_sketch_fast.hip): {flo, fB, fHd, rlo, rB, rHd}.
inline void poly_a_state(unsigned k, uint32_t out[6])
{
	uint64_t fh = 0, rh = 0;
	for (unsigned i = 0; i < k; ++i) {
		fh ^= srol(seed_of(0), i);
		rh ^= srol(comp_of(0), i);
	}
	out[0] = lo_of(fh);
	out[1] = b32_of(fh) << 31;
	out[2] = hd_of(fh);
	out[3] = lo_of(rh);
	out[4] = b32_of(rh);
	out[5] = hd_of(rh);
}

// Closed-form table of the resolve stage, two bases per entry: entry (j, a, b) covers window positions 2j, 2j+1
//   fwd = srol^(k-1-2j)(seed(a)) ^ srol^(k-2-2j)(seed(b)),  rev = srol^(2j)(comp(a)) ^ srol^(2j+1)(comp(b))
// as {fwd.lo, fwd.hi, rev.lo, rev.hi} (nthash.hpp:220-239: fh = XOR_i srol^(k-1-i) seed(c_i), rh = XOR_i srol^i comp(c_i)).
// For odd k the last pair has no second base (its b term is dropped).  Entry offset: j*256 + (a*4+b)*16 bytes.
inline unsigned t2_pairs(unsigned k) { return (k + 1) / 2; }
inline void build_t2(unsigned k, uint32_t* out /* t2_pairs(k)*16*4 dwords */, unsigned gap_first = 0, unsigned gap = 0)
{
	auto dc = [&](unsigned i) { return i >= gap_first && i < gap_first + gap; }; // don't-care position of a spaced seed
	for (unsigned j = 0; j < t2_pairs(k); ++j)
		for (unsigned a = 0; a < 4; ++a)
			for (unsigned b = 0; b < 4; ++b) {
				const unsigned i0 = 2 * j, i1 = 2 * j + 1;
				uint64_t f = srol(seed_of(a), k - 1 - i0), r = srol(comp_of(a), i0);
				if (dc(i0)) f = r = 0; // spaced seed: this base does not enter the hash (nthash.hpp:641-646)
				if (i1 < k && !dc(i1)) {
					f ^= srol(seed_of(b), k - 1 - i1);
					r ^= srol(comp_of(b), i1);
				}
				uint32_t* e = out + ((j * 16) + a * 4 + b) * 4;
				e[0] = (uint32_t)f;
				e[1] = (uint32_t)(f >> 32);
				e[2] = (uint32_t)r;
				e[3] = (uint32_t)(r >> 32);
			}
}

// Closed-form table of K1b's resolve stage, FOUR bases per entry, indexed by a packed byte of 2-bit codes in
// code2 order (code2 = (ascii >> 1) & 3: A=0 C=1 T/U=2 G=3; base t of the group in bits 2t+1:2t):
// entry (g, v) = XOR over t < 4, i = 4 g + t < k of  srol^(k-1-i)(seed(c_t))  and  srol^i(comp(c_t))   (nthash.hpp:220-239)
inline unsigned t4_groups(unsigned k) { return (k + 3) / 4; }
inline void build_t4(unsigned k, uint32_t* out /* t4_groups(k)*256*4 dwords */, unsigned gap_first = 0, unsigned gap = 0)
{
	static const unsigned code_of_code2[4] = { 0, 1, 3, 2 }; // code2 -> the A C G T numbering of seed_of()
	for (unsigned g = 0; g < t4_groups(k); ++g)
		for (unsigned v = 0; v < 256; ++v) {
			uint64_t f = 0, r = 0;
			for (unsigned t = 0; t < 4; ++t) {
				const unsigned i = 4 * g + t;
				if (i >= k) break;
				if (i >= gap_first && i < gap_first + gap) continue; // spaced seed: a don't-care position (nthash.hpp:641-646)
				const unsigned c = code_of_code2[(v >> (2 * t)) & 3u];
				f ^= srol(seed_of(c), k - 1 - i);
				r ^= srol(comp_of(c), i);
			}
			uint32_t* e = out + ((size_t)g * 256 + v) * 4;
			e[0] = (uint32_t)f;
			e[1] = (uint32_t)(f >> 32);
			e[2] = (uint32_t)r;
			e[3] = (uint32_t)(r >> 32);
		}
}

// Spaced-seed filter table: for the p-th PAIR of don't-care positions (i, i+1), entry (a, b) holds the H halves
// (Hd layout) of  srol^(k-1-i)(seed(a)) ^ srol^(k-2-i)(seed(b))  and  srol^i(comp(a)) ^ srol^(i+1)(comp(b)),
// i.e. what NTMSM64 XORs out of fh / rh (nthash.hpp:641-646).  16-byte stride: {f.Hd, r.Hd, 0, 0}.
inline void build_gap_table(unsigned k, unsigned gap_first, unsigned gap, uint32_t* out /* ceil(gap/2)*16*4 dwords */)
{
	for (unsigned p = 0; p < (gap + 1) / 2; ++p)
		for (unsigned a = 0; a < 4; ++a)
			for (unsigned b = 0; b < 4; ++b) {
				const unsigned i0 = gap_first + 2 * p, i1 = i0 + 1;
				uint64_t f = srol(seed_of(a), k - 1 - i0), r = srol(comp_of(a), i0);
				if (i1 < gap_first + gap) {
					f ^= srol(seed_of(b), k - 1 - i1);
					r ^= srol(comp_of(b), i1);
				}
				uint32_t* e = out + ((p * 16) + a * 4 + b) * 4;
				e[0] = hd_of(f);
				e[1] = hd_of(r);
				e[2] = e[3] = 0;
			}
}

// Rolling form of the spaced seed (equal-length waves): shifting the window by one base changes four terms — the
// incoming and outgoing base (the ordinary step table) plus the base that leaves the don't-care block at its low
// end (a, window index gap_first -> gap_first-1, now cared for) and the one that enters it at its high end
// (b, index gap_first+gap -> gap_first+gap-1).  Entry (a, b): {f, r} H halves (Hd layout) of
//   f: srol^(k-gap_first)(seed(a)) ^ srol^(k-gap_first-gap)(seed(b))        (XORed in after the forward rotate)
//   r: srol^(gap_first)(comp(a))   ^ srol^(gap_first+gap)(comp(b))          (XORed in before the reverse rotate)
inline void build_gap_roll_table(unsigned k, unsigned gap_first, unsigned gap, uint32_t out[16][2])
{
	for (unsigned a = 0; a < 4; ++a)
		for (unsigned b = 0; b < 4; ++b) {
			const uint64_t f = srol(seed_of(a), k - gap_first) ^ srol(seed_of(b), k - gap_first - gap);
			const uint64_t r = srol(comp_of(a), gap_first) ^ srol(comp_of(b), gap_first + gap);
			out[a * 4 + b][0] = hd_of(f);
			out[a * 4 + b][1] = hd_of(r);
		}
}

} // namespace ntc
