// ntc_instr.hpp — the ONE place where instrumentation builds are switched on.  Product builds define none of these macros and every hook
// below expands to nothing; tools/dbg and tools/ab_build.sh build private copies of the library with
//   -DTS_TIMERS      K1c (ntc_sketch_ts.hip): cycles per role and per wait site, summed over waves into TsArgs::dbg (uint64 [4][8])
//   -DTS_DEBUG       K1c: every resolved candidate dumped to TsArgs::dbg (tools/dbg/ts_dbg.hip)
//   -DNTC_BS_TIMERS  K1b (ntc_sketch_bs.hip): per-phase cycle counts, summed over waves into BsArgs::dbg
// K1h (gen_k1h.py) has its own switch, the generator's K1H_EXP environment variable (tools/k1h_variant.sh): timers, and the timing
// experiments noload / nopack / nopass / noflags / nosched, whose results are wrong on purpose.
#pragma once

#ifdef TS_TIMERS // instrumentation build (tools/dbg): cycles per role and per wait site, summed over waves into a.dbg (uint64 [4][8])
#define TS_T(var) const uint64_t var = __builtin_readcyclecounter()
#define TS_ACC(slot, t0, t1) tacc[slot] += (t1) - (t0)
#define TS_WAIT(slot, p, v)                          \
	do {                                             \
		const uint64_t w0__ = __builtin_readcyclecounter(); \
		lds_wait_ge(p, v);                           \
		tacc[slot] += __builtin_readcyclecounter() - w0__;  \
	} while (0)
#define TS_FLUSH(role)                                                                                            \
	do {                                                                                                          \
		tacc[0] = __builtin_readcyclecounter() - t_start;                                                         \
		if (lane == 0 && a.dbg)                                                                                   \
			for (int i = 0; i < 8; ++i)                                                                           \
				atomicAdd(reinterpret_cast<unsigned long long*>(a.dbg) + (role) * 8 + i, (unsigned long long)tacc[i]); \
	} while (0)
#else
#define TS_T(var)
#define TS_ACC(slot, t0, t1)
#define TS_WAIT(slot, p, v) lds_wait_ge(p, v)
#define TS_FLUSH(role)
#endif
