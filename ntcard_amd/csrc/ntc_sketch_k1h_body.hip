// ntc_sketch_k1h_body.hip — K1h's kernels: compiled K1H_GEN_PARTS times (-DK1H_PART=p: the variants with k % parts == p), see ntc_sketch_k1h.hip
// for what the kernel pair computes.  C++ here only stages the closed-form table in LDS, shares the workgroup's blocks out among its eight waves
// and hands seven scalars to the generated assembly (gen_k1h.py: explicit physical registers, exactly 255 VGPRs).
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "ntc_kernels.hpp"

#ifndef K1H_PART
#error "compile with -DK1H_PART=0 .. K1H_GEN_PARTS - 1"
#endif

namespace ntc {

namespace {

#include "ntc_k1h_gen_defs.inc"
#include "ntc_k1h_gen.inc" // (only the strings of this part: #if K1H_PART == ...)

static_assert(offsetof(K1hArgs, tiles) == 0 && offsetof(K1hArgs, log) == 8 && offsetof(K1hArgs, log_fill) == 16 && offsetof(K1hArgs, sketch0) == 24 &&
                  offsetof(K1hArgs, f1) == 32 && offsetof(K1hArgs, dirty) == 40 && offsetof(K1hArgs, tie) == 48 && offsetof(K1hArgs, n_tiles) == 56 &&
                  offsetof(K1hArgs, n_chunks) == 60 && offsetof(K1hArgs, read_len) == 64 && offsetof(K1hArgs, nv_last) == 68 && offsetof(K1hArgs, key_base) == 72 &&
                  offsetof(K1hArgs, rmask2) == 76 && offsetof(K1hArgs, log_regions) == 80 && offsetof(K1hArgs, log_region_cap) == 84 && offsetof(K1hArgs, table) == 88 &&
                  offsetof(K1hArgs, blocks_per_wave) == 104 && offsetof(K1hArgs, nb_magic) == 108 && offsetof(K1hArgs, sus) == 112 &&
                  offsetof(K1hArgs, sus_count) == 120 && offsetof(K1hArgs, sus_cap) == 128 && offsetof(K1hArgs, tails) == 144 && offsetof(K1hArgs, sk_dirty) == 160,
              "gen_k1h.KARG");
constexpr uint32_t kK1hWaves = K1H_GEN_WAVES;
constexpr uint32_t kK1hWArea = K1H_GEN_WAREA;
static_assert(kK1hWaves == 8, "the launch (512 threads: two waves on every SIMD) assumes eight waves per workgroup");
constexpr uint32_t kK1hThreads = kK1hWaves * 64u;
#ifndef K1H_OLD_SHARE
#define K1H_OLD_SHARE 555u // (of 1024: the share of a SIMD pair's blocks that its older wave takes)
#endif
constexpr uint32_t kK1hTableOff = kK1hWaves * kK1hWArea;
constexpr uint32_t k1h_table_bytes(uint32_t k) { return 2u * ((k + 2u) / 3u) * 256u; }
constexpr uint32_t k1h_lds_bytes(uint32_t k) { return kK1hTableOff + k1h_table_bytes(k); } // the wave areas and the table

#define K1H_CLOBBERS_V                                                                                                                 \
	"v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20",   \
	    "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39",   \
	    "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58",   \
	    "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77",   \
	    "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96",   \
	    "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113",  \
	    "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129",       \
	    "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145",       \
	    "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161",       \
	    "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177",       \
	    "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193",       \
	    "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209",       \
	    "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225",       \
	    "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241",       \
	    "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254"
#define K1H_CLOBBERS_S "s26", "s27", "s28", "s29", "s30", "s31", "s34", "s35", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99", "vcc", "memory"

// gen_k1h.py emits one body per (k, gap) and s_bits class (7, >= 8): an assembly string with explicit registers
template <int K, int SB, int GAP> struct K1hBody;
#define K1H_BODY_SPEC(k, g, sb)                                                                                                                              \
	template <> struct K1hBody<k, sb, g> {                                                                                                                  \
		static __device__ __forceinline__ void run(uint32_t karg_lo, uint32_t karg_hi, uint32_t wave_gid, uint32_t n_waves, uint32_t lds_wbase,             \
		                                           uint32_t first_block, uint32_t end_block)                                                                \
		{                                                                                                                                                   \
			asm volatile(K1H_ASM_K##k##_G##g##_S##sb ::"s"(karg_lo), "s"(karg_hi), "s"(wave_gid), "s"(n_waves), "s"(lds_wbase), "s"(first_block),           \
			             "s"(end_block)                                                                                                                     \
			             : K1H_CLOBBERS_V, K1H_CLOBBERS_S);                                                                                                 \
		}                                                                                                                                                   \
	};
#define K1H_BODY_SPECS(k, g) K1H_BODY_SPEC(k, g, 7) K1H_BODY_SPEC(k, g, 8)
#define K1H_CAT2(a, b) a##b
#define K1H_CAT(a, b) K1H_CAT2(a, b)
#define K1H_MY_VARIANTS K1H_CAT(K1H_VARIANTS_P, K1H_PART)
K1H_MY_VARIANTS(K1H_BODY_SPECS)

} // namespace

#ifdef K1H_WAVE_CLOCKS // timing experiment (tools/k1h_variant.sh, K1H_CXXFLAGS=-DK1H_WAVE_CLOCKS): every wave of the LAST launch leaves its first and last clock (100 MHz)
static __device__ unsigned long long g_k1h_wave_clocks[2 * 4096];
#endif

template <int K, int SB, int GAP>
__global__ __launch_bounds__(kK1hThreads) void sketch_k1h_kernel(const K1hMulti m)
{
#ifdef K1H_WAVE_CLOCKS
	const unsigned long long wc_t0 = __builtin_amdgcn_s_memrealtime();
#endif
	extern __shared__ __align__(16) unsigned char smem[];
	// the segment (batch) this workgroup walks: the segments own consecutive workgroup ranges (K1hMulti; a launch over one batch has one segment)
	uint32_t s = 0;
	while (s + 1u < m.n_segs && blockIdx.x >= m.seg[s + 1u].first_wg)
		++s;
	s = (uint32_t)__builtin_amdgcn_readfirstlane((int)s);
	const K1hArgs& a = m.seg[s];
	{ // the closed-form table: [2 strands][ceil(k / 3)][64] dwords behind the eight wave areas
		constexpr uint32_t n = 2u * ((K + 2) / 3) * 64u;
		uint32_t* dst = reinterpret_cast<uint32_t*>(smem + kK1hTableOff);
		const uint32_t* tab = m.seg[0].table; // (the same for every segment)
		for (uint32_t i = threadIdx.x; i < n; i += kK1hThreads)
			dst[i] = tab[i];
	}
	const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	__syncthreads();
	// Eight waves, two on every SIMD (round 4 ran six — two pairs and two lone waves — and weighed their shares by who sat alone): the workgroup's blocks
	// are shared out as contiguous ranges of the flat sequence tile * NB + block of ITS segment, a larger one for the first wave of every SIMD (below).
	const uint32_t bpw = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.blocks_per_wave);
	const uint32_t wg_local = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x - a.first_wg));
	// Round 6: the two waves of a SIMD are NOT equals — the SIMD issues the older wave first, and with equal shares waves 0 .. 3 (the first on their SIMDs)
	// finished 13 % before waves 4 .. 7 in every workgroup of every launch (per-wave clocks of a -DK1H_WAVE_CLOCKS build, profiles/r06_k1h_wave_clocks.txt:
	// 2775 against 3204 us of a 3.2 ms launch), which then ran alone at half a pair's throughput (the older wave of a pair runs as fast as a wave alone, the younger at 0.83 of that).  The older waves take K1H_OLD_SHARE / 1024 of a pair's blocks.
	const uint32_t quota = bpw * kK1hWaves, wg0 = wg_local * quota;
	const uint32_t b_old = (2u * bpw * K1H_OLD_SHARE + 512u) >> 10, b_young = 2u * bpw - b_old;
	const uint32_t my0 = wave < 4u ? wave * b_old : 4u * b_old + (wave - 4u) * b_young;
	const uint32_t first_block = (uint32_t)__builtin_amdgcn_readfirstlane((int)(wg0 + my0));
	const uint32_t end_block = (uint32_t)__builtin_amdgcn_readfirstlane((int)(wg0 + my0 + (wave < 4u ? b_old : b_young)));
	const uint32_t wave_gid = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * kK1hWaves + wave));
#ifdef K1H_STATIC_PRIO // timing experiment (tools/k1h_variant.sh): the second wave of every SIMD runs at a fixed higher priority
	if (wave >= 4) __builtin_amdgcn_s_setprio(K1H_STATIC_PRIO);
#endif
	const uint32_t n_waves = (uint32_t)__builtin_amdgcn_readfirstlane((int)(gridDim.x * kK1hWaves));
	const uint32_t lds_wbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)(wave * kK1hWArea));
	// the body reads its arguments with scalar loads from the segment's K1hArgs inside the kernel argument segment
	const uint64_t karg = (uint64_t)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(K1hMulti, seg) + (uint64_t)s * sizeof(K1hArgs);
	const uint32_t karg_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)karg);
	const uint32_t karg_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(karg >> 32));
	K1hBody<K, SB, GAP>::run(karg_lo, karg_hi, wave_gid, n_waves, lds_wbase, first_block, end_block);
#ifdef K1H_WAVE_CLOCKS
	if ((threadIdx.x & 63u) == 0u && blockIdx.x * kK1hWaves + (threadIdx.x >> 6) < 4096u) {
		g_k1h_wave_clocks[2u * (blockIdx.x * kK1hWaves + (threadIdx.x >> 6))] = wc_t0;
		g_k1h_wave_clocks[2u * (blockIdx.x * kK1hWaves + (threadIdx.x >> 6)) + 1u] = __builtin_amdgcn_s_memrealtime();
	}
#endif
}

// ---- what ntc_sketch_k1h.hip calls: launch / shared-memory attribute of this part's kernels ----
#define K1H_LAUNCH_CASE(kk, gg)                                                                                                                              \
	if (k == kk && gap == gg) {                                                                                                                             \
		*found = true;                                                                                                                                      \
		if (sb7) hipLaunchKernelGGL((sketch_k1h_kernel<kk, 7, gg>), dim3(grid), dim3(kK1hThreads), lds, st, b);                                                      \
		else hipLaunchKernelGGL((sketch_k1h_kernel<kk, 8, gg>), dim3(grid), dim3(kK1hThreads), lds, st, b);                                                          \
		return hipGetLastError();                                                                                                                           \
	}
#define K1H_SMEM_CASE(kk, gg)                                                                                                                                \
	if (rc == hipSuccess) rc = hipFuncSetAttribute(reinterpret_cast<const void*>(&sketch_k1h_kernel<kk, 7, gg>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)k1h_lds_bytes(kk)); \
	if (rc == hipSuccess) rc = hipFuncSetAttribute(reinterpret_cast<const void*>(&sketch_k1h_kernel<kk, 8, gg>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)k1h_lds_bytes(kk));

hipError_t K1H_CAT(k1h_launch_part, K1H_PART)(uint32_t k, uint32_t gap, bool sb7, unsigned grid, uint32_t lds, hipStream_t st, const K1hMulti& b, bool* found)
{
	K1H_MY_VARIANTS(K1H_LAUNCH_CASE)
	*found = false;
	return hipSuccess;
}

#ifdef K1H_WAVE_CLOCKS
} // namespace ntc
extern "C" int K1H_CAT(ntc_dbg_k1h_wave_clocks_p, K1H_PART)(unsigned long long* out)
{
	return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ntc::g_k1h_wave_clocks), sizeof(unsigned long long) * 2 * 4096);
}
namespace ntc {
#endif

hipError_t K1H_CAT(k1h_set_smem_part, K1H_PART)()
{
	hipError_t rc = hipSuccess;
	K1H_MY_VARIANTS(K1H_SMEM_CASE)
	return rc;
}

} // namespace ntc
