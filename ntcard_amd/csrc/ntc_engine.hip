// ntc_engine.hip — C-ABI shim (include/ntcard_hip.h) over the gfx950 kernels.
//
// Host-side mirror of the reference seam B2: ntc_create = the allocation/zeroing
// main() does (ntcard.cpp:433-439), ntc_submit = a batch of ntRead/stRead calls
// (ntcard.cpp:147-171), ntc_finish = the state compEst reads (ntcard.cpp:237-247) + F1
// (ntcard.cpp:464-466).  No CPU fallback exists: every entry point needs a live HIP device.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <unordered_set>
#include <vector>

#include "../../include/ntcard_hip.h"
#include "ntc_kernels.hpp"

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...)
{
	char buf[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof buf, fmt, ap);
	va_end(ap);
	g_err = buf;
	return code;
}

} // namespace

// the same for the other translation units of the library (ntc_estimator.cpp)
int ntc_internal_fail(int code, const char* fmt, ...)
{
	char buf[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof buf, fmt, ap);
	va_end(ap);
	g_err = buf;
	return code;
}

namespace {

#define HIP_TRY(expr)                                                                               \
	do {                                                                                            \
		hipError_t e__ = (expr);                                                                    \
		if (e__ != hipSuccess)                                                                      \
			return fail(NTC_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e__));          \
	} while (0)

constexpr uint32_t kMaxK = 600; // the closed-form table (k x 128 B) and one wave of host slots (64 x ~2k B) share the 160 KiB of LDS; 600 is tested, 640 no longer fits
constexpr uint32_t kSlotCapMin = 256; // host packing: slot capacity (bytes) for ragged batches

struct DevInfo {
	int cus = 0;
};

int device_info(int dev, DevInfo& di)
{
	static std::mutex mu;
	static std::vector<int> cus; // per device, queried once (hipGetDeviceProperties is slow)
	std::lock_guard<std::mutex> lk(mu);
	if (dev < 0) return fail(NTC_ERR_ARG, "bad device %d", dev);
	if ((size_t)dev >= cus.size()) cus.resize(dev + 1, 0);
	if (cus[dev] == 0) {
		hipDeviceProp_t p;
		HIP_TRY(hipGetDeviceProperties(&p, dev));
		cus[dev] = p.multiProcessorCount;
	}
	di.cus = cus[dev];
	return 0;
}

// kernel kinds: the simple validation kernel (K1s/K1d) and the production kernels (K1)
enum { KIND_SIMPLE = 0, KIND_HF = 2 };

// dynamic LDS per block of the simple kernel: its seed tables + the 4 waves' slots
size_t smem_simple(uint32_t stride) { return (size_t)ntc::kTableBytes + (size_t)ntc::kWavesPerBlock * 64u * stride; }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-device property of a kernel function, not of a launch: it
// is raised ONCE per device to the most any plan can ask for (under a process-wide lock), so that engines driven from
// different threads never lower each other's limit between "set" and "launch".
constexpr size_t kMaxDynLds = 160 * 1024 - 2048; // 160 KiB minus the largest static LDS of any kernel here
int ensure_kernel_attrs(int dev)
{
	static std::mutex mu;
	static std::vector<char> done;
	std::lock_guard<std::mutex> lk(mu);
	if ((size_t)dev >= done.size()) done.resize(dev + 1, 0);
	if (done[dev]) return 0;
	HIP_TRY(ntc::set_sketch_hf_smem_limit(kMaxDynLds));
	HIP_TRY(ntc::set_hash_smem_limit(kMaxDynLds));
	HIP_TRY(ntc::set_apply_smem_limit());
	HIP_TRY(ntc::set_sketch_k1h_smem_limit());
	done[dev] = 1;
	return 0;
}

// K1 (sketch_hf_kernel) launch shape.  Every wave parks its 64 slots in LDS and the block shares
// the closed-form tables, so the waves a CU can hold are bounded by its 160 KiB of LDS; pick the block size
// (1..16 waves) that packs the most waves per CU (a fixed 4-wave block loses a third of them at k = 64).
struct HfPlan {
	unsigned grid = 0, wpb = 0, waves_per_cu = 0;
	size_t smem = 0;
};

// per-k block of the K1 argument struct
void fill_hfk(ntc::HfK& o, uint32_t k, uint32_t* sketch, unsigned long long* f1, const void* t1, uint32_t key_base = 0)
{
	ntc::HashTables tab;
	uint32_t init[6];
	ntc::build_tables(k, tab);
	ntc::poly_a_state(k, init);
	o.k = k;
	o.init_f = init[2];
	o.init_r = init[5];
	o.pad_ = 0;
	o.key_base = key_base;
	o.pad2_ = 0;
	o.sketch = sketch;
	o.f1 = f1;
	o.t1 = t1;
	for (int slot = 0; slot < ntc::kMainSlots; ++slot) {
		o.tabh[slot][0] = tab.A[slot][1];
		o.tabh[slot][1] = tab.A[slot][3];
	}
}
// block shape only (no device call): waves per CU and waves per block for a slot stride, 0 waves = does not fit
void hf_shape(uint32_t stride, const uint32_t* ks, uint32_t n_k, uint32_t gap, HfPlan& p, size_t& shared_out);

int hf_plan(int dev, uint64_t n_slots, uint32_t stride, const uint32_t* ks, uint32_t n_k, uint32_t gap, HfPlan& p)
{
	DevInfo di;
	if (int rc = device_info(dev, di)) return rc;
	size_t shared = 0;
	hf_shape(stride, ks, n_k, gap, p, shared);
	if (p.waves_per_cu == 0)
		return fail(NTC_ERR_ARG, "slot stride %u with k=%u needs more than 160 KiB of LDS per wave", stride, ks[0]);
	p.smem = shared + p.wpb * (64u * (size_t)stride);
	if (p.smem > kMaxDynLds) return fail(NTC_ERR_ARG, "slot stride %u with k=%u needs %zu B of LDS per block", stride, ks[0], p.smem);
	const unsigned per_cu = std::max(1u, p.waves_per_cu / p.wpb);
	const uint64_t need = (n_slots + 64ull * p.wpb - 1) / (64ull * p.wpb);
	p.grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(need, (uint64_t)di.cus * per_cu));
	return 0;
}

void hf_shape(uint32_t stride, const uint32_t* ks, uint32_t n_k, uint32_t gap, HfPlan& p, size_t& shared_out)
{
	const size_t per_wave = 64u * (size_t)stride; // the wave's 64 decoded slots; hit masks and the compaction queue are registers
	size_t shared = 16 + (size_t)((gap + 1u) / 2u) * 256u;
	for (uint32_t j = 0; j < n_k; ++j)
		shared += (size_t)ntc::t2_pairs(ks[j]) * 256u; // the closed-form tables of every fused k are resident
	// A workgroup's LDS (dynamic + the kernel's static tables) is allocated in granules of 1280 B, 128 of them per
	// CU (measured: 3 blocks of 43 granules do not co-reside, 3 of 42 do).  A plan that overestimates the resident
	// blocks leaves part of the persistent grid waiting for a second round, which costs far more than a wave less.
	const size_t granule = 1280, granules_per_cu = 128, static_lds = 1024 + 256 + 64;
	unsigned best_waves = 0;
	auto consider = [&](unsigned w) {
		const size_t alloc = (shared + w * per_wave + static_lds + granule - 1) / granule;
		if (alloc > granules_per_cu || shared + w * per_wave > kMaxDynLds) return;
		const unsigned waves = std::min<unsigned>(16, (unsigned)(granules_per_cu / alloc) * w); // 128 VGPRs: 4 waves per SIMD
		if (waves >= best_waves) { // ties: the larger block (fewer table copies)
			best_waves = waves;
			p.wpb = w;
		}
	};
	// whole multiples of the 4 SIMDs keep them evenly loaded (measured: 6 or 13 waves per block cost 5-12 %,
	// 3 blocks of 3 waves lose to 2 blocks of 4); smaller blocks only when not even 4 waves fit
	for (unsigned w = 4; w <= 16; w += 4)
		consider(w);
	for (unsigned w = 3; best_waves == 0 && w >= 1; --w)
		consider(w);
	p.waves_per_cu = best_waves;
	shared_out = shared;
}

// Slot stride the host packer uses for reads of up to `maxlen` bytes: a multiple of 4; an ODD number of dwords keeps
// the 64 lanes of a wave on distinct LDS banks when they read the same column of their slots (160 B = 40 dwords is
// an 8-way conflict, measured 8 % slower than 156 B), taken whenever it does not cost a wave of occupancy.
uint32_t pick_stride(uint64_t maxlen, const std::vector<uint32_t>& klist, uint32_t gap)
{
	const uint32_t s0 = (uint32_t)((maxlen + 3) & ~3ull);
	if ((s0 / 4) & 1u) return s0;
	HfPlan a, b;
	size_t sh;
	hf_shape(s0, klist.data(), (uint32_t)std::min<size_t>(klist.size(), ntc::kMaxFusedK), gap, a, sh);
	hf_shape(s0 + 4, klist.data(), (uint32_t)std::min<size_t>(klist.size(), ntc::kMaxFusedK), gap, b, sh);
	return b.waves_per_cu >= a.waves_per_cu && b.waves_per_cu > 0 ? s0 + 4 : s0;
}

// grid for the simple (validation) kernel: enough blocks to fill the chip, not more than the work
int hash_grid(int dev, uint64_t n_slots, uint32_t stride, unsigned& grid, size_t& smem)
{
	DevInfo di;
	if (int rc = device_info(dev, di)) return rc;
	smem = smem_simple(stride);
	if (smem > kMaxDynLds) return fail(NTC_ERR_ARG, "slot stride %u needs %zu B of LDS per block (> 160 KiB)", stride, smem);
	unsigned per_cu = (unsigned)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / smem));
	uint64_t need = (n_slots + 64 * ntc::kWavesPerBlock - 1) / (64 * ntc::kWavesPerBlock);
	uint64_t cap = (uint64_t)di.cus * per_cu;
	grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(need, cap));
	return 0;
}

// geometry of the partition passes (plan_log); A/B builds override them (tools/ab_build.sh <name> -DNTC_AB_G1=512): the product has no run-time knob
#ifndef NTC_AB_G1
#define NTC_AB_G1 256
#endif
#ifndef NTC_AB_PARTS2
#define NTC_AB_PARTS2 4
#endif
#ifndef NTC_AB_SLICE_BITS
#define NTC_AB_SLICE_BITS 15
#endif
constexpr uint32_t kApplyG1 = NTC_AB_G1, kApplyParts2 = NTC_AB_PARTS2, kApplySliceBits = NTC_AB_SLICE_BITS;

uint32_t ceil_log2(uint64_t x)
{
	uint32_t b = 0;
	while ((1ull << b) < x) ++b;
	return b;
}

} // namespace

struct ntc_engine {
	int device = 0;
	hipStream_t stream = nullptr;
	std::vector<uint32_t> klist;
	uint32_t gap = 0, r_bits = 27, s_bits = 7;
	std::vector<ntc::HfK> hfk;    // per-k argument blocks of K1 (tables derived once at create)
	uint32_t* d_sketch = nullptr; // [nk][2][1<<r_bits]
	unsigned long long* d_f1 = nullptr;
	bool own_sketch = false, own_f1 = false;
	uint32_t* d_phist = nullptr; // [nk][2][65536]
	uint16_t* d_out16 = nullptr; // [2][1<<r_bits] scratch for finish
	int kernel_kind = 2;         // KIND_HF unless NTC_FLAG_SIMPLE_KERNEL
	// Hit log (ntc_apply.hip): K1 appends the counter index of every sampled k-mer instead of incrementing; the
	// log is applied to d_sketch when it fills up and whenever the sketch itself is needed (finish, merge, ...).
	uint32_t* d_log = nullptr;      // [log_regions][log_region_cap]
	uint32_t* d_logfill = nullptr;  // [log_regions]
	uint32_t log_regions = 0, log_region_cap = 0;
	uint32_t klog_regions = 0;      // K1f's own regions BEHIND the hash kernels' log_regions (round 6: the suspects it counts are log entries too); the apply reads all of them
	uint64_t log_cap = 0;           // entries (of the hash kernels' regions)
	// round 6: does the sketch still hold the zeros of the last reset?  Then the first apply writes its counts instead of adding them (count_kernel).  The
	// host knows about applies, merges and pointers it has handed out (sk_host_dirty); kernels that increment the sketch themselves — K1 in direct mode, any
	// wave out of log regions, K1f's slow path, a partition run that overflowed — set the device word.
	uint32_t* d_skdirty = nullptr;
	bool sk_host_dirty = true;
	bool sk_exposed = false;        // ntc_device_state has handed the counters' address out: the caller may add to them whenever it likes
	uint32_t all_log_regions() const { return log_regions + klog_regions; }
	double log_est = 0.0;           // host-side upper estimate of the entries logged since the last apply
	bool log_pending = false;
	// log or direct atomics: decided ON THE DEVICE from a sample of what the first sizeable batch after a reset logged
	// (repeated keys -> the counters stay cached -> direct atomics are cheaper; ntc_apply.hip, log_probe_kernel)
	uint32_t* d_logmode = nullptr;          // 0 = log, 1 = direct atomics
	bool partition_always = false;          // NTC_FLAG_PARTITION_ALWAYS
	unsigned long long* d_logstats = nullptr; // {keys sampled, repeats among them}
	uint32_t* d_probe = nullptr;            // 2^20-slot hash table of the probe
	bool adaptive = true, probed = false;
	struct ApplyPlan {
		uint32_t key_bits = 0, slice_bits = 0, b1 = 0, b2 = 0; // key = [b1 | b2 | slice_bits]
		uint32_t g1 = 0, parts2 = 0, cap1 = 0, cap2 = 0, n_slices = 0;
		// bytes per key in the runs of partition pass 1 / 2: the LAST pass writes uint16 (the run implies the slice; A3 reads the low slice_bits <= 15 bits)
		size_t key_bytes(int pass) const { return (pass == 2 || b2 == 0) ? 2 : 4; }
	} ap;
	uint32_t *d_s1 = nullptr, *d_c1 = nullptr, *d_s2 = nullptr, *d_c2 = nullptr; // partition scratch (allocated at the first apply)
	std::vector<void*> d_t4s;       // K1f: closed-form table, 4 bases per entry, per k of the list (empty: the engine has no tiled kernel)
	std::vector<uint32_t*> d_k1h_tabs; // K1h: closed-form table (3 bases per entry) per k of the list
	// What K1h hands to K1f (two bit arrays, the suspect list, a little state): one set per K1h launch whose K1f is still to come.  K1f's kernels
	// wait on memory (a few dependent loads per dirty piece / suspect, ~45 us per kernel whatever the batch), so for a caller that promised to leave
	// its batches alone until ntc_sync (NTC_FLAG_DEFER_REDO) the engine collects up to kK1fBatch K1h launches and sends ONE K1f over all of them;
	// without the promise K1f follows its K1h launch at once (set 0).
	struct K1hSet {
		uint32_t *d_dirty = nullptr, *d_tie = nullptr;
		size_t dirty_cap = 0, tie_cap = 0;
		uint4* d_sus = nullptr;            // sus_cap entries per wave of a launch
		uint32_t sus_cap = 0;
		uint32_t *d_sus_count = nullptr, *d_fix_state = nullptr;
	} k1h_set[ntc::kK1fBatch];
	ntc::K1fBatch k1f_batch;        // the launches waiting for K1f (k1f_batch.item[i] uses k1h_set[i])
	uint32_t k1f_n = 0;
	// Round 6: with NTC_FLAG_DEFER_REDO the HASH launches wait as well — up to eight device-resident tiled batches are hashed by ONE K1h launch per k, as
	// segments that share its workgroups (K1hMulti, built for the length bins of a ragged read set).  A K1h wave that starts inside a tile walks two masked
	// blocks first to fill its window: 8 % of a 10 M-read launch (24 blocks per wave), 1 % of an 80 M-read one; and a launch's ramp and tail are paid once.
	// The caller's promise is the same as for K1f: the batches stay unchanged until ntc_sync.  Everything that reads counters or F1, or ends the promise,
	// goes through join_k1f, which launches what waits here first.
	struct DeferredSeg {
		const unsigned char* d_tiles;
		uint64_t n_reads;
		uint32_t read_len;
		const uint32_t* d_tails;
	};
	std::vector<DeferredSeg> deferred;
	bool in_flush = false;
	// ntc_merge_devices: exchange buffers, copy streams and events, kept between merges (grow-only)
	struct MergeCache {
		uint16_t *narrow = nullptr, *recv = nullptr;
		size_t narrow_cap = 0, recv_cap = 0; // elements
		std::vector<hipStream_t> lanes;
		std::vector<hipEvent_t> arrived;
		hipEvent_t narrowed = nullptr, summed = nullptr;
	} mc;
	uint64_t merge_allocs = 0; // device allocations + streams + events ntc_merge_devices has created for this engine
	uint32_t k1h_launch_id = 0;
	// profiling of the tiled path: ONE pair of events brackets a RUN of hash launches (a pair per launch costs 10 - 20 us of stream bubbles per
	// launch, measured); the run ends when anything else is about to enter the stream (K1f, an apply, another kind of batch, a sync)
	hipEvent_t run_ev0 = nullptr;
	std::vector<uint64_t> pending_runs; // index-aligned with `pending`: 0 = one launch; n + 1 = a bracket of tiled launches that counts as n submits
	uint64_t run_submits = 0;
	std::vector<std::pair<hipEvent_t, hipEvent_t>> k1f_events; // profiling: deferred K1f launches (outside the hash kernels' events)
	// An event record is a stream bubble of its own, so where two timed spans touch — hash bracket | K1f | hash bracket, K1f | apply — the end event of the first
	// IS the start event of the second; the second pair then BORROWS its first event (it is destroyed as the other pair's second)
	std::unordered_set<hipEvent_t> borrowed;
	double k1f_ms = 0.0;
	bool ts_ok = false;             // the tiled kernel pair K1h + K1f is built for SOME k of this configuration (k_tiled says which) ...
	bool ts_all = false;            // ... for every k (then nothing of a tiled batch is left to K1)
	std::vector<uint8_t> k_tiled;   // per k of the list: K1h + K1f take it from tiled batches (the others are K1's, which stages the same tiles: round 5)
	bool ts_required = false;       // NTC_FLAG_REQUIRE_TILED
	bool defer_redo = false;        // NTC_FLAG_DEFER_REDO
	unsigned char* d_untile = nullptr; // row-major scratch for tiled batches of configurations K1h is not built for
	size_t untile_cap = 0;
	uint32_t* d_tmeta = nullptr;       // K1's slot table (len | len << 16 per read) of a RAGGED tiled batch under a list of which a part is K1's (round 6)
	size_t tmeta_cap = 0;              // reads
	double apply_ms = 0.0;
	uint64_t applies = 0;
	uint32_t hll_bits = 0;       // != 0: nthll engine (d_sketch holds uint32 M[1<<hll_bits])
	uint32_t* d_hll_thr = nullptr;
	uint64_t hll_reads_seen = 0;
	void* d_gapt = nullptr;      // spaced seed: filter table of the don't-care positions
	std::vector<void*> d_t1;     // per k: closed-form table of the H-filter kernel's resolve stage
	// host-submit staging (grow-only)
	// ntc_submit staging: a small pool of pinned host + device buffer pairs.  A caller packs its reads into a free
	// pair WITHOUT holding the engine lock (the reference's per-file parser threads pack in parallel), then enqueues
	// copy + kernels under the lock; `done` marks the point on the stream after which the pair may be reused.
	struct StageSlot {
		unsigned char* h_stage = nullptr;
		uint32_t* h_meta = nullptr;
		size_t h_stage_cap = 0, h_meta_cap = 0;
		unsigned char* d_stage = nullptr;
		uint32_t* d_meta = nullptr;
		size_t d_stage_cap = 0;
		hipEvent_t done = nullptr;
		bool busy = false, used = false;
	};
	static constexpr int kStageSlots = 4;
	StageSlot stage[kStageSlots];
	std::mutex stage_mu;
	std::condition_variable stage_cv;
	std::mutex mu;
	// profiling of the hash kernel (HIP events on the engine stream)
	bool profiling = false;
	std::vector<std::pair<hipEvent_t, hipEvent_t>> pending, apply_pending;
	double ms_total = 0.0;
	uint64_t launches = 0;

	uint64_t plane_elems() const { return 2ull << r_bits; }
};

namespace {

int close_run(ntc_engine* e, hipEvent_t* recorded);
inline int close_run(ntc_engine* e) { return close_run(e, nullptr); }

int drain_events(ntc_engine* e)
{
	if (int rc = close_run(e)) return rc;
	// every span first (a pair may borrow its first event from another pair: ntc_engine::borrowed), then the events go
	for (auto& pr : e->pending) {
		float ms = 0.f;
		HIP_TRY(hipEventSynchronize(pr.second));
		HIP_TRY(hipEventElapsedTime(&ms, pr.first, pr.second));
		e->ms_total += ms;
		const size_t idx = (size_t)(&pr - e->pending.data());
		e->launches += idx < e->pending_runs.size() && e->pending_runs[idx] ? e->pending_runs[idx] - 1 : 1; // (a bracket: submits + 1; 0: a single launch)
	}
	for (auto& pr : e->apply_pending) {
		float ms = 0.f;
		HIP_TRY(hipEventSynchronize(pr.second));
		HIP_TRY(hipEventElapsedTime(&ms, pr.first, pr.second));
		e->apply_ms += ms;
	}
	for (auto& pr : e->k1f_events) {
		float ms = 0.f;
		HIP_TRY(hipEventSynchronize(pr.second));
		HIP_TRY(hipEventElapsedTime(&ms, pr.first, pr.second));
		e->k1f_ms += ms;
	}
	for (auto* v : { &e->pending, &e->apply_pending, &e->k1f_events }) {
		for (auto& pr : *v) {
			if (!e->borrowed.count(pr.first)) (void)hipEventDestroy(pr.first);
			(void)hipEventDestroy(pr.second);
		}
		v->clear();
	}
	e->pending_runs.clear();
	e->borrowed.clear();
	return 0;
}

// Geometry of the hit log and of its partition passes for this engine's key space (keys are indices into the whole
// sketch array: k index, sample and bucket).  Returns false when the keys do not fit (then ntComp's increments stay
// direct atomics): more than 2^32 counters, or more than two 8-bit partition passes above a 2^15-counter slice.
bool plan_log(ntc_engine* e, uint64_t want_entries)
{
	const uint64_t counters = e->klist.size() * e->plane_elems();
	if (counters > (1ull << 32)) return false;
	auto& ap = e->ap;
	ap.key_bits = ceil_log2(counters);
	ap.slice_bits = std::min<uint32_t>(kApplySliceBits, ap.key_bits);
	const uint32_t pb = ap.key_bits - ap.slice_bits;
	if (pb > 16) return false;
	ap.b1 = pb <= 8 ? pb : (pb + 1) / 2; // two passes: balanced fan-out (longer runs per digit coalesce better than 256-way + 32-way); with the second
	                                     // pass's uint16 runs 7 + 6 bits still beat 6 + 7 and 5 + 8 (0.101 / 0.103 / 0.131 ms per step, profiles/r05_apply_geometry_sweep.txt)
	ap.b2 = pb - ap.b1;
	ap.n_slices = (uint32_t)((counters + (1ull << ap.slice_bits) - 1) >> ap.slice_bits);
	// default: four entries per counter, at most 2^30 (4 GiB at rBits = 27 and one k: the apply's sweep over the whole sketch is then
	// paid once per ~900 M sampled k-mers; 288 GB of HBM have room for the log and its two partition work areas, 12 GiB in all)
	uint64_t cap = want_entries ? want_entries : std::min<uint64_t>(1ull << 30, std::max<uint64_t>(1ull << 18, 4 * counters));
	cap = std::max<uint64_t>(cap, 1ull << 14);
	e->log_region_cap = (uint32_t)std::min<uint64_t>(32768, std::max<uint64_t>(256, cap / 8192)); // <= 65535: one run fits a 16-bit count pass
	e->log_regions = (uint32_t)std::max<uint64_t>(1, cap / e->log_region_cap);
	e->log_cap = (uint64_t)e->log_regions * e->log_region_cap;
	e->klog_regions = std::max<uint32_t>(1, std::min<uint32_t>(1024, e->log_regions / 8)); // 32 Mi entries at the default geometry; a full region falls back to atomics
	// pass 1: g1 workgroups, each owns every g1-th region and writes 2^b1 private runs; a run holds its expected
	// share of a FULL log + 25 % (+64); what does not fit is applied directly (exact), so the margin is about speed only
	ap.g1 = std::min<uint32_t>(e->log_regions, kApplyG1); // one 1024-thread workgroup per CU: few, long private runs (measured 128 … 4096)
	const uint64_t share1 = (uint64_t)((e->all_log_regions() + ap.g1 - 1) / ap.g1) * e->log_region_cap;
	ap.cap1 = (uint32_t)(((share1 >> ap.b1) * 5 / 4 + 64 + 7) & ~7ull); // (multiples of 8 keys: the count pass reads uint16 runs 16 bytes at a time)
	// pass 2: bucket b of pass 1 is split again by `parts2` workgroups
	ap.parts2 = kApplyParts2;
	const uint64_t share2 = ((((uint64_t)e->all_log_regions() * e->log_region_cap) >> ap.b1) * 5 / 4) / ap.parts2 + 1;
	ap.cap2 = (uint32_t)(((share2 >> ap.b2) * 13 / 10 + 64 + 7) & ~7ull);
	return true;
}

// K1f over the K1h launches that still wait for it (asynchronous on the engine's stream).  Before anything reads the counters or F1, touches the
// sketch without atomics (the apply's sweep does), or hands the batches back to the caller.
int close_run(ntc_engine* e, hipEvent_t* recorded) // the end of a bracketed run of tiled hash launches (see run_ev0); *recorded = the event it has just recorded (or null)
{
	if (recorded) *recorded = nullptr;
	if (!e->run_ev0) return 0;
	hipEvent_t ev1 = nullptr;
	HIP_TRY(hipEventCreate(&ev1));
	HIP_TRY(hipEventRecord(ev1, e->stream));
	e->pending_runs.resize(e->pending.size(), 0);
	e->pending.emplace_back(e->run_ev0, ev1);
	if (recorded) *recorded = ev1;
	e->pending_runs.push_back(e->run_submits + 1); // (+ 1: a bracket re-opened in the middle of a k list has counted its submit already and adds none)
	e->run_ev0 = nullptr;
	e->run_submits = 0;
	return 0;
}

int flush_deferred(ntc_engine* e); // the hash launches of the batches that wait in e->deferred (below, behind run_tiled_segs)

int join_k1f(ntc_engine* e, hipEvent_t* last = nullptr) // *last = the event recorded behind the last thing this call launched (K1f's end, or the hash bracket's), or null
{
	if (last) *last = nullptr;
	if (int rc = flush_deferred(e)) return rc;
	hipEvent_t closed = nullptr; // (the end of the hash launches' bracket, recorded this very moment, is K1f's start too: an event record is a stream bubble of its own)
	if (int rc = close_run(e, &closed)) return rc;
	if (last) *last = closed;
	if (e->k1f_n == 0) return 0;
	DevInfo di;
	if (int rc = device_info(e->device, di)) return rc;
	hipEvent_t f0 = nullptr, f1 = nullptr;
	bool f0_shared = false;
	if (e->profiling) {
		HIP_TRY(hipEventCreate(&f1));
		if (closed) {
			f0 = closed;
			f0_shared = true;
		} else {
			HIP_TRY(hipEventCreate(&f0));
			HIP_TRY(hipEventRecord(f0, e->stream));
		}
	}
	const uint32_t n = e->k1f_n;
	e->k1f_n = 0;
	HIP_TRY(ntc::launch_k1h_fixup(e->k1f_batch, n, (unsigned)di.cus, e->stream));
	if (e->profiling) {
		HIP_TRY(hipEventRecord(f1, e->stream));
		e->k1f_events.emplace_back(f0, f1);
		if (f0_shared) e->borrowed.insert(f0);
		if (last) *last = f1;
	} else if (last) {
		*last = nullptr;
	}
	return 0;
}

// Apply the pending hit log to the sketch (asynchronous on the engine's stream): partition, count, add, clear.
int apply_log(ntc_engine* e)
{
	hipEvent_t before = nullptr; // (the event behind K1f / the hash bracket, if this call has just recorded one: the apply's start)
	if (int rc = join_k1f(e, &before)) return rc;
	if (!e->d_log || !e->log_pending) return 0;
	const auto& ap = e->ap;
	const uint32_t nb1 = 1u << ap.b1, nb2 = 1u << ap.b2;
	if (ap.b1 && !e->d_s1) {
		const size_t runs1 = (size_t)ap.g1 * nb1;
		if (hipMalloc((void**)&e->d_s1, runs1 * ap.cap1 * ap.key_bytes(1)) != hipSuccess || hipMalloc((void**)&e->d_c1, runs1 * 4) != hipSuccess)
			return fail(NTC_ERR_MEMORY, "cannot allocate %zu B of partition scratch on device", runs1 * ap.cap1 * ap.key_bytes(1));
	}
	if (ap.b2 && !e->d_s2) {
		const size_t runs2 = (size_t)nb1 * ap.parts2 * nb2;
		if (hipMalloc((void**)&e->d_s2, runs2 * ap.cap2 * ap.key_bytes(2)) != hipSuccess || hipMalloc((void**)&e->d_c2, runs2 * 4) != hipSuccess)
			return fail(NTC_ERR_MEMORY, "cannot allocate %zu B of partition scratch on device", runs2 * ap.cap2 * ap.key_bytes(2));
	}
	hipEvent_t ev0 = nullptr, ev1 = nullptr;
	if (e->profiling) {
		HIP_TRY(hipEventCreate(&ev1));
		if (before) {
			ev0 = before;
			e->borrowed.insert(ev0);
		} else {
			HIP_TRY(hipEventCreate(&ev0));
			HIP_TRY(hipEventRecord(ev0, e->stream));
		}
	}
	// little in the log (decided on the device: fewer than 4 M entries): plain atomics, and the passes below find it empty
	// (an engine whose every k is K1h's always logs, so the host's estimate of a large log is good enough to go straight to the partition passes — which are exact
	// for a small log too, only slower —: two tiny kernels and a stream bubble less per apply)
	if (!e->partition_always && !(e->ts_all && e->log_est >= (double)(64u << 20)))
		HIP_TRY(ntc::launch_log_atomics(e->d_log, e->d_logfill, e->log_region_cap, e->all_log_regions(), (uint32_t*)(e->d_logstats + 2), e->d_sketch, e->d_skdirty, e->stream));
	ntc::CountArgs c;
	std::memset(&c, 0, sizeof c);
	c.slice_bits = ap.slice_bits;
	c.n_slices = ap.n_slices;
	c.sketch = e->d_sketch;
	c.first = e->sk_host_dirty ? 0u : 1u; // nothing the host knows of has touched the sketch since the reset: the device word decides
	c.sk_dirty = e->d_skdirty;
	if (ap.b1 == 0) {
		c.in = e->d_log;
		c.in_cnt = e->d_logfill;
		c.in_cap = e->log_region_cap;
		c.n_in = e->all_log_regions();
		c.mode = 0;
	} else {
		ntc::SplitArgs s1;
		std::memset(&s1, 0, sizeof s1);
		s1.in = e->d_log;
		s1.in_cnt = e->d_logfill;
		s1.in_cap = e->log_region_cap;
		s1.n_in = e->all_log_regions();
		s1.mode = 0;
		s1.sk_dirty = e->d_skdirty;
		s1.shift = ap.key_bits - ap.b1;
		s1.bits = ap.b1;
		s1.out = e->d_s1;
		s1.out_cnt = e->d_c1;
		s1.out_cap = ap.cap1;
		s1.sketch = e->d_sketch;
		s1.narrow = ap.b2 == 0;
		// round 6: between two passes the keys travel three to a 64-bit word (what the first pass leaves of a key is <= 21 bits: 2.7 B per key written and
		// read instead of 4; the run's capacity in words is half its capacity in keys — the same bytes)
		const bool packed = ap.b2 != 0 && ap.key_bits - ap.b1 <= 21;
		if (packed) {
			s1.pack_out = 1;
			s1.out_cap = ap.cap1 / 2;
		}
		HIP_TRY(ntc::launch_split(s1, ap.g1, e->stream));
		if (ap.b2 == 0) {
			c.in = e->d_s1;
			c.in_cnt = e->d_c1;
			c.in_cap = ap.cap1;
			c.n_in = ap.g1 * nb1;
			c.mode = 1;
			c.nb1 = nb1;
			c.nwg1 = ap.g1;
			c.in16 = 1;
		} else {
			ntc::SplitArgs s2;
			std::memset(&s2, 0, sizeof s2);
			s2.in = e->d_s1;
			s2.in_cnt = e->d_c1;
			s2.in_cap = ap.cap1;
			s2.n_in = ap.g1 * nb1;
			s2.mode = 1;
			s2.parts = ap.parts2;
			s2.nb_in = nb1;
			s2.shift = ap.slice_bits;
			s2.bits = ap.b2;
			s2.out = e->d_s2;
			s2.out_cnt = e->d_c2;
			s2.out_cap = ap.cap2;
			s2.sketch = e->d_sketch;
			s2.sk_dirty = e->d_skdirty;
			s2.narrow = 1;
			if (packed) {
				s2.pack_in = 1;
				s2.in_cap = ap.cap1 / 2;
				s2.hi_shift = ap.key_bits - ap.b1;
			}
			HIP_TRY(ntc::launch_split(s2, nb1 * ap.parts2, e->stream));
			c.in = e->d_s2;
			c.in_cnt = e->d_c2;
			c.in_cap = ap.cap2;
			c.n_in = nb1 * ap.parts2 * nb2;
			c.mode = 2;
			c.parts = ap.parts2;
			c.nb2 = nb2;
			c.in16 = 1;
		}
	}
	DevInfo di;
	if (int rc = device_info(e->device, di)) return rc;
	if (c.mode != 0) { // (the count pass reads partition runs, not the log: it clears the log's fill words itself)
		c.clear_fill = e->d_logfill;
		c.n_clear = e->all_log_regions();
	}
	HIP_TRY(ntc::launch_count(c, std::min<unsigned>(ap.n_slices, (ap.slice_bits >= 15 ? 2u : 4u) * (unsigned)di.cus), e->stream));
	e->sk_host_dirty = true;
	if (c.mode == 0) HIP_TRY(hipMemsetAsync(e->d_logfill, 0, (size_t)e->all_log_regions() * 4, e->stream));
	if (e->profiling) {
		HIP_TRY(hipEventRecord(ev1, e->stream));
		e->apply_pending.emplace_back(ev0, ev1);
	}
	e->log_pending = false;
	e->log_est = 0.0;
	e->applies += 1;
	return 0;
}

// launch the hash->sample->count kernel for every k of the list over one device-resident batch
int run_batch(ntc_engine* e, const unsigned char* d_slots, const uint32_t* d_meta, uint64_t n_slots, uint32_t read_len, uint32_t stride, bool tiled = false,
              const std::vector<uint8_t>* skip = nullptr);
// tiled: d_slots is a TILED batch (stride = 16 x its chunks: K1 stages the tiles itself); skip: the k of the list that are NOT this call's (K1h has taken them)
int run_batch(ntc_engine* e, const unsigned char* d_slots, const uint32_t* d_meta, uint64_t n_slots,
              uint32_t read_len, uint32_t stride, bool tiled, const std::vector<uint8_t>* skip)
{
	if (n_slots == 0) return 0;
	if (int rc = close_run(e)) return rc;
	unsigned grid = 0;
	size_t smem = 0;
	const int kind = e->kernel_kind;
	if (kind != KIND_HF) // K1 plans its launch per k group (hf_plan)
		if (int rc = hash_grid(e->device, n_slots, stride, grid, smem)) return rc;
	if (e->hll_bits) {
		// nthll: refresh the "can still matter" threshold between sub-batches that double in size, so the
		// expensive resolve stage only sees a vanishing fraction of the k-mers once the registers warm up
		uint64_t done = 0;
		while (done < n_slots) {
			uint64_t n = std::max<uint64_t>(16384, e->hll_reads_seen);
			n = std::min<uint64_t>((n + 63) & ~63ull, n_slots - done); // whole waves: a sub-batch starts on a 16-byte aligned slot
			HIP_TRY(ntc::launch_hll_threshold(e->d_sketch, 1u << e->hll_bits, e->d_hll_thr, e->stream));
			ntc::HfArgs a;
			std::memset(&a, 0, sizeof a);
			a.slots = d_slots + done * stride;
			a.meta = d_meta ? d_meta + done : nullptr;
			a.n_slots = n;
			a.stride = stride;
			a.read_len = read_len;
			a.r_bits = 27;
			a.s_bits = 7;
			a.n_k = 1;
			a.hll_bits = e->hll_bits;
			a.hll_thr = e->d_hll_thr;
			a.ks[0] = e->hfk[0];
			HfPlan hp;
			if (int rc = hf_plan(e->device, n, stride, &e->klist[0], 1, 0, hp)) return rc;
			HIP_TRY(ntc::launch_sketch_hf(a, hp.grid, hp.wpb, hp.smem, e->stream));
			done += n;
			e->hll_reads_seen += n;
		}
		return 0;
	}
	if (kind == KIND_HF) {
		// upper estimate of the sampled k-mers per slot (both samples ~2^-sBits of the windows each, App. B of SURVEY.md)
		auto mine = [&](size_t ki) { return !(skip && (*skip)[ki]); };
		double per_slot = 0.0;
		for (size_t ki = 0; ki < e->klist.size(); ++ki)
			if (mine(ki)) per_slot += (double)std::max<int64_t>(0, (int64_t)(d_meta ? stride : read_len) - (int64_t)e->klist[ki] + 1) * std::ldexp(1.15, 1 - (int)e->s_bits);
		// The first sizeable equal-length batch after a reset is cut in two: a small head goes first, the probe samples what
		// it logged and decides log vs direct atomics on the device, and the bulk of the batch already runs in that mode.
		// The head is sized to log the ~2^20 entries the probe wants (0.6 M slots at sBits = 7, k = 32); a batch that is
		// not several heads long (large sBits, small batches) is not cut: the probe then runs once enough has been logged.
		constexpr double kProbeEntries = 1.25 * (1 << 20);
		if (e->d_log && e->adaptive && !e->probed && d_meta == nullptr && per_slot > 0.0) {
			const uint64_t head = (((uint64_t)(kProbeEntries / per_slot) + 2047) / 2048) * 2048;
			if (e->log_est < (double)(1u << 20) && n_slots >= 4 * head) {
				// (head is a multiple of 2048 reads: in a tiled batch, whose tiles hold 2048 x stride bytes each, the rest starts at the same offset)
				if (int rc = run_batch(e, d_slots, nullptr, head, read_len, stride, tiled, skip)) return rc;
				return run_batch(e, d_slots + head * stride, nullptr, n_slots - head, read_len, stride, tiled, skip);
			}
		}
		hipEvent_t ev0 = nullptr, ev1 = nullptr;
		if (e->d_log) {
			// this batch's sampled k-mers + what every wave may leave unused at the end of a region; apply first if the log could fill up
			const double est = 64.0 * 4096 + (double)n_slots * per_slot;
			if (e->log_pending && e->log_est + est > 0.85 * (double)e->log_cap)
				if (int rc = apply_log(e)) return rc;
			e->log_est += est;
			e->log_pending = true;
		}
		if (e->profiling) { // the hash kernels of this batch only: an apply has its own pair of events
			HIP_TRY(hipEventCreate(&ev0));
			HIP_TRY(hipEventCreate(&ev1));
			HIP_TRY(hipEventRecord(ev0, e->stream));
		}
		// K1: one launch per group of up to kMaxFusedK values of k (the batch is staged and decoded once per group);
		// a group whose closed-form tables would push the CU below 12 waves (and below what its members reach alone) is split in two
		std::function<int(size_t, size_t, const unsigned char*, uint64_t)> launch_group =
		    [&](size_t b, size_t n, const unsigned char* slots, uint64_t ns) -> int {
			HfPlan hp;
			if (int rc = hf_plan(e->device, ns, stride, &e->klist[b], (uint32_t)n, e->gap, hp)) {
				if (n == 1) return rc;
				hp.waves_per_cu = 0;
			}
			unsigned worst_single = 16; // waves per CU of the least favourable member launched on its own
			for (size_t j = 0; n > 1 && j < n; ++j) {
				HfPlan one;
				size_t sh;
				hf_shape(stride, &e->klist[b + j], 1, e->gap, one, sh);
				worst_single = std::min(worst_single, one.waves_per_cu);
			}
			if (n > 1 && hp.waves_per_cu < 12 && hp.waves_per_cu < worst_single) {
				if (int rc = launch_group(b, n / 2, slots, ns)) return rc;
				return launch_group(b + n / 2, n - n / 2, slots, ns);
			}
			ntc::HfArgs a;
			std::memset(&a, 0, sizeof a);
			a.slots = slots;
			a.meta = d_meta;
			a.n_slots = ns;
			a.stride = stride;
			a.read_len = read_len;
			a.tiled = tiled ? 1u : 0u;
			a.r_bits = e->r_bits;
			a.s_bits = e->s_bits;
			a.n_k = (uint32_t)n;
			a.gap = e->gap;
			a.gap_first = (e->klist[b] - e->gap) / 2;
			a.gapt = e->d_gapt;
			if (e->gap) ntc::build_gap_roll_table(e->klist[b], a.gap_first, e->gap, a.tabg);
			for (size_t j = 0; j < n; ++j)
				a.ks[j] = e->hfk[b + j];
			if (e->d_log) {
				a.log = e->d_log;
				a.log_fill = e->d_logfill;
				a.log_regions = e->log_regions;
				a.log_region_cap = e->log_region_cap;
				a.log_mode = e->d_logmode;
			}
			a.sketch0 = e->d_sketch;
			a.sk_dirty = e->d_skdirty;
			HIP_TRY(ntc::launch_sketch_hf(a, hp.grid, hp.wpb, hp.smem, e->stream));
			return 0;
		};
		for (size_t b = 0; b < e->klist.size();) { // runs of this call's k, up to kMaxFusedK per launch
			if (!mine(b)) {
				++b;
				continue;
			}
			size_t n = 1;
			while (n < ntc::kMaxFusedK && b + n < e->klist.size() && mine(b + n))
				++n;
			if (int rc = launch_group(b, n, d_slots, n_slots)) return rc;
			b += n;
		}
		if (e->profiling) {
			HIP_TRY(hipEventRecord(ev1, e->stream));
			e->pending.emplace_back(ev0, ev1);
		}
		if (e->d_log && e->adaptive && !e->probed && e->log_est >= (double)(1u << 20)) { // enough logged since the reset: sample the log, decide log vs atomics
			e->probed = true;
			HIP_TRY(ntc::launch_log_probe(e->d_log, e->d_logfill, e->log_region_cap, std::min<uint32_t>(e->log_regions, 1024), 256, e->d_probe, 1u << 20,
			                              e->d_logstats, e->d_logmode, e->stream));
		}
		return 0;
	}
	for (size_t ki = 0; ki < e->klist.size(); ++ki) {
		ntc::HashArgs a;
		std::memset(&a, 0, sizeof a);
		a.slots = d_slots;
		a.meta = d_meta;
		a.n_slots = n_slots;
		a.stride = stride;
		a.read_len = read_len;
		a.k = e->klist[ki];
		a.r_bits = e->r_bits;
		a.s_bits = e->s_bits;
		a.sketch = e->d_sketch + ki * e->plane_elems();
		a.f1 = e->d_f1 + ki;
		ntc::build_tables(a.k, a.tab);
		ntc::poly_a_state(a.k, a.init);
		hipEvent_t ev0 = nullptr, ev1 = nullptr;
		if (e->profiling) {
			HIP_TRY(hipEventCreate(&ev0));
			HIP_TRY(hipEventCreate(&ev1));
			HIP_TRY(hipEventRecord(ev0, e->stream));
		}
		a.t1 = e->d_t1[ki];
		a.gapt = e->d_gapt;
		a.gap = e->gap;
		a.gap_first = (a.k - e->gap) / 2;
		HIP_TRY(ntc::launch_hash(0, a, grid, smem, e->stream));
		if (e->profiling) {
			HIP_TRY(hipEventRecord(ev1, e->stream));
			e->pending.emplace_back(ev0, ev1);
		}
	}
	return 0;
}

// a tiled batch for K1: re-laid out as row-major slots on the device (exact; not a fast path)
int run_tiled_as_rows(ntc_engine* e, const unsigned char* d_tiles, uint64_t n_reads, uint32_t read_len)
{
	const uint32_t stride = pick_stride(read_len, e->klist, e->gap);
	const size_t need = (size_t)n_reads * stride + 16;
	if (need > e->untile_cap) {
		HIP_TRY(hipStreamSynchronize(e->stream));
		if (e->d_untile) (void)hipFree(e->d_untile);
		e->d_untile = nullptr;
		e->untile_cap = 0;
		if (hipMalloc((void**)&e->d_untile, need) != hipSuccess) return fail(NTC_ERR_MEMORY, "cannot allocate %zu B of row-major scratch on device", need);
		e->untile_cap = need;
	}
	HIP_TRY(ntc::launch_untile(d_tiles, e->d_untile, n_reads, read_len, stride, e->stream));
	return run_batch(e, e->d_untile, nullptr, n_reads, read_len, stride);
}

// K1h + K1f over one device-resident batch in the tiled layout (include/ntcard_hip.h: ntc_submit_tiled_device)
// One device-resident tiled batch: equal-length reads (d_tails == nullptr) or one length bin of a ragged read set (read_len = 16 C)
struct TiledSeg {
	const unsigned char* d_tiles;
	uint64_t n_reads;
	uint32_t read_len;
	const uint32_t* d_tails;
};
int run_tiled_segs(ntc_engine* e, const TiledSeg* segs_in, uint32_t n_in, uint64_t n_submits = 1);

// A list of which a part is K1's: K1 stages the SAME tiles, 64 slots of 16 x ceil(len / 16) bytes per wave next to its closed-form tables.  Does that fit the
// CU's LDS for every such k on its own (run_batch splits a fused group that does not fit; a single k has to)?  Equal-length reads beyond ~2.4 kb do not:
// host batches then take row slots, whose packer cuts long sequences into overlapping chunks, and a device-resident tiled batch is refused BEFORE
// anything of it has been counted.
bool k1_fits_tiles(const ntc_engine* e, uint32_t read_len)
{
	if (e->ts_all) return true;
	const uint32_t stride = 16u * ((read_len + 15u) / 16u);
	for (size_t ki = 0; ki < e->klist.size(); ++ki) {
		if (e->k_tiled[ki]) continue;
		HfPlan p;
		size_t shared = 0;
		hf_shape(stride, &e->klist[ki], 1, e->gap, p, shared);
		if (p.waves_per_cu == 0 || shared + p.wpb * (64u * (size_t)stride) > kMaxDynLds) return false;
	}
	return true;
}

int run_tiled(ntc_engine* e, const unsigned char* d_tiles, uint64_t n_reads, uint32_t read_len, const uint32_t* d_tails = nullptr)
{
	const TiledSeg one{d_tiles, n_reads, read_len, d_tails};
	return run_tiled_segs(e, &one, 1);
}

// K1h + K1f over up to kK1hSegs tiled batches of different geometry in ONE launch per k (launch_sketch_k1h_multi): the length bins of a ragged read set
// share the launch's workgroups in proportion to their blocks instead of queueing as small launches, each of which would pay the waves' start-up again
int run_tiled_segs(ntc_engine* e, const TiledSeg* segs_in, uint32_t n_in, uint64_t n_submits) // n_submits: the caller's submits these batches came in (ntc_kernel_time's launch count)
{
	std::vector<TiledSeg> segs;
	for (uint32_t i = 0; i < n_in; ++i)
		if (segs_in[i].n_reads) segs.push_back(segs_in[i]);
	if (segs.empty()) return 0;
	bool any_tails = false;
	for (const auto& sg : segs)
		any_tails |= sg.d_tails != nullptr;
	if (!e->ts_all && e->ts_required) return fail(NTC_ERR_ARG, "ntc_submit_tiled_device: the tiled kernel is not available for this configuration (NTC_FLAG_REQUIRE_TILED)");
	if (!e->ts_ok && any_tails) return fail(NTC_ERR_ARG, "ntc_submit_tiled_ragged_device: the tiled kernels are not built for any k of this configuration");
	if (!e->ts_ok) { // this configuration is K1's
		for (const auto& sg : segs)
			if (int rc = run_tiled_as_rows(e, sg.d_tiles, sg.n_reads, sg.read_len)) return rc;
		return 0;
	}
	if (e->ts_ok && !e->ts_all)
		for (const auto& sg : segs)
			if (!k1_fits_tiles(e, sg.read_len))
				return fail(NTC_ERR_ARG, "tiled batch of %u-base reads: the k of this list that the general kernel serves cannot stage such tiles in LDS (submit the reads through ntc_submit / ntc_submit_spans, which cut long sequences into chunks); nothing was counted", sg.read_len);
	DevInfo di;
	if (int rc = device_info(e->device, di)) return rc;
	// (a launch gives every batch at least one workgroup, and the suspect lists are sized for cus x 8 regions: no more batches than CUs)
	const uint32_t max_segs = std::min<uint32_t>(std::min<uint32_t>(ntc::kK1hSegs, ntc::kK1fBatch), (uint32_t)std::max(1, di.cus));
	if (segs.size() > max_segs) { // more bins than one launch takes: groups
		for (size_t i = 0; i < segs.size(); i += max_segs)
			if (int rc = run_tiled_segs(e, segs.data() + i, (uint32_t)std::min<size_t>(max_segs, segs.size() - i))) return rc;
		return 0;
	}
	for (const auto& sg : segs) {
		// K1h addresses its bit arrays (one word per tile, chunk / block and lane) with 32-bit byte offsets: a batch of several hundred GB is cut in
		// two at a tile boundary, as often as it takes (any prefix of a tiled buffer is a batch)
		const uint64_t n_tiles = (sg.n_reads + ntc::kTileReads - 1) / ntc::kTileReads;
		const uint64_t rows = (uint64_t)(sg.read_len + 15u) / 16u + 2u; // chunks, and at most chunks + 1 blocks, per tile
		if (n_tiles * rows * 256u >= (1ull << 32)) {
			for (const auto& s2 : segs) { // (such a set of batches goes one by one, halves first)
				if (&s2 != &sg) {
					if (int rc = run_tiled_segs(e, &s2, 1)) return rc;
					continue;
				}
				const uint64_t head_tiles = n_tiles / 2, head_reads = head_tiles * ntc::kTileReads;
				const TiledSeg head{sg.d_tiles, head_reads, sg.read_len, sg.d_tails};
				const TiledSeg rest{sg.d_tiles + ntc_tiled_bytes(head_reads, sg.read_len), sg.n_reads - head_reads, sg.read_len, sg.d_tails ? sg.d_tails + head_tiles * 16 : nullptr};
				if (int rc = run_tiled_segs(e, &head, 1)) return rc;
				if (int rc = run_tiled_segs(e, &rest, 1)) return rc;
			}
			return 0;
		}
	}
	if (!e->ts_all && e->d_log && e->adaptive && !e->probed && segs.size() == 1 && segs[0].d_tails == nullptr && e->log_est < (double)(1u << 20)) {
		// A list of which a part is K1's: K1 appends to the hit log or increments with device atomics, whichever the probe of the FIRST sizeable batch finds
		// cheaper for this data (run_batch).  The probe needs a log whose sampled entries come from few reads — the head of the batch, hashed by both kernels
		// before the rest: cut the batch as run_batch cuts a row-slot batch (at a tile boundary: any prefix of a tiled buffer is a batch).
		const TiledSeg& sg = segs[0];
		double per_read = 0.0;
		for (uint32_t k : e->klist)
			per_read += (double)std::max<int64_t>(0, (int64_t)sg.read_len - (int64_t)k + 1) * std::ldexp(1.15, 1 - (int)e->s_bits);
		if (per_read > 0.0) {
			const uint64_t head = (((uint64_t)(1.25 * (1 << 20) / per_read) + 2047) / 2048) * 2048;
			if (sg.n_reads >= 4 * head) {
				const TiledSeg first{sg.d_tiles, head, sg.read_len, nullptr};
				const TiledSeg rest{sg.d_tiles + ntc_tiled_bytes(head, sg.read_len), sg.n_reads - head, sg.read_len, nullptr};
				if (int rc = run_tiled_segs(e, &first, 1)) return rc;
				return run_tiled_segs(e, &rest, 1);
			}
		}
	}
	if (e->d_log) {
		// candidates of these batches (both samples ~2^-sBits of the windows each, every k) + what every logging wave may leave unused at the
		// end of a region; a log that could overflow is applied first (outside the hash kernels' timing events)
		double est = 0;
		for (size_t kj = 0; kj < e->klist.size(); ++kj) {
			const uint32_t k = e->klist[kj];
			if (!e->k_tiled[kj]) continue; // (run_batch books K1's share)
			bool any = false;
			for (const auto& sg : segs)
				if (sg.read_len >= k) {
					est += (double)sg.n_reads * (double)(sg.read_len - k + 1) * std::ldexp(1.15, 1 - (int)e->s_bits);
					any = true;
				}
			if (any) est += 64.0 * 4096;
		}
		if (e->log_pending && e->log_est + est > 0.85 * (double)e->log_cap)
			if (int rc = apply_log(e)) return rc;
		e->log_est += est;
		e->log_pending = true;
	}
	// ntRead's loop over kList (ntcard.cpp:147-158): one launch per k over the same resident batches
	auto open_run = [&](hipEvent_t borrow = nullptr) -> int { // (again behind a K1f that had to come in the middle of the list; borrow: the event that K1f has just left)
		if (e->profiling && !e->run_ev0) {
			if (borrow) {
				e->run_ev0 = borrow;
				e->borrowed.insert(borrow);
			} else {
				HIP_TRY(hipEventCreate(&e->run_ev0));
				HIP_TRY(hipEventRecord(e->run_ev0, e->stream));
			}
		}
		return 0;
	};
	bool counted = false; // this submit counts as one launch of ntc_kernel_time once its first kernel is queued (not at all when read_len < every k)
	bool launched_any = false;
	const uint32_t wpg = ntc::sketch_k1h_waves();                                              // waves per workgroup (one workgroup per CU)
	const uint32_t max_waves = (uint32_t)di.cus * wpg;
	// what the hand-over arrays must hold for the most demanding k of the list: sized ONCE, before the first launch, so that an allocation failure can only
	// come while nothing of these batches has been counted (a later k used to be able to fail behind an earlier k's launch)
	size_t need_d_all = 0, need_t_all = 0;
	uint32_t sus_cap_all = 0;
	auto sus_cap_of = [&](uint32_t blocks_per_wave) {
		// Suspects per K1h wave: room for EVERY candidate of the wave's share (reads dense with non-base bytes make every candidate a suspect:
		// with a short list the launch fell back to K1f's slow path — 15 ms per 10 M reads at 2 % N).  The share: the
		// blocks of a wave (plan_sketch_k1h: even shares of a workgroup's quota) + 1, all of
		// them full (2048 reads x 16 windows); ntComp's patterns pass 3 / 256 of the windows at sBits = 7, their 8-bit prefixes 2 / 256 at
		// sBits >= 8 (ntcard.cpp:132-145), measured 1.3 x that on reads with 10 % N (ties ride along): x 1.5, + 1024, at least 2048, at most
		// 1 GiB per launch (beyond that a launch may still overflow: slow path, exact).  Round 6: the batches of ONE launch share one list — a wave's region is
		// its number in the launch, and every batch is walked by waves of its own — so a launch over eight batches needs one list, not eight.
		const double lone_blocks = (double)blocks_per_wave + 1.0;
		const double per_block = 2048.0 * 16.0 * (e->s_bits == 7 ? 3.0 : 2.0) / 256.0;
		return (uint32_t)std::min<double>(std::max<double>(2048.0, 1.5 * lone_blocks * per_block + 1024.0), (double)((1ull << 30) / 16u / max_waves));
	};
	for (size_t ki = 0; ki < e->klist.size(); ++ki) {
		const uint32_t k = e->klist[ki];
		if (!e->k_tiled[ki]) continue;
		ntc::K1hArgs hs0[ntc::kK1hSegs], pl0[ntc::kK1hSegs];
		std::memset(hs0, 0, sizeof hs0);
		uint32_t n0 = 0;
		for (const auto& sg : segs)
			if (sg.read_len >= k) {
				hs0[n0].n_tiles = (uint32_t)((sg.n_reads + ntc::kTileReads - 1) / ntc::kTileReads);
				hs0[n0].read_len = sg.read_len;
				++n0;
			}
		if (n0 == 0) continue;
		(void)ntc::plan_sketch_k1h(hs0, n0, k, (unsigned)di.cus, pl0);
		for (uint32_t i = 0; i < n0; ++i) {
			const uint32_t n_chunks = (hs0[i].read_len + 15u) / 16u, nb = ntc::sketch_k1h_blocks(k, hs0[i].read_len);
			need_d_all = std::max(need_d_all, (size_t)hs0[i].n_tiles * n_chunks * 256);
			need_t_all = std::max(need_t_all, (size_t)hs0[i].n_tiles * nb * 256);
			sus_cap_all = std::max(sus_cap_all, sus_cap_of(pl0[i].blocks_per_wave));
		}
	}
	for (size_t ki = 0; ki < e->klist.size(); ++ki) {
		const uint32_t k = e->klist[ki];
		if (!e->k_tiled[ki]) continue; // K1's (below)
		std::vector<const TiledSeg*> act; // no window of this k in a shorter read (ntHashIterator.hpp:61-64)
		for (const auto& sg : segs)
			if (sg.read_len >= k) act.push_back(&sg);
		if (act.empty()) continue;
		const uint32_t na = (uint32_t)act.size();
		hipEvent_t after = nullptr;
		if (e->k1f_n + na > ntc::kK1fBatch) // no sets left for these launches: K1f over the waiting ones first
			if (int rc = join_k1f(e, &after)) return rc;
		if (int rc = open_run(after)) return rc;
		if (e->profiling && !counted) {
			e->run_submits += n_submits;
			counted = true;
		}
		// the launch's geometry: workgroups and blocks per wave of every batch
		ntc::K1hArgs hs[ntc::kK1hSegs], planned[ntc::kK1hSegs];
		std::memset(hs, 0, sizeof hs);
		for (uint32_t i = 0; i < na; ++i) {
			hs[i].n_tiles = (uint32_t)((act[i]->n_reads + ntc::kTileReads - 1) / ntc::kTileReads);
			hs[i].read_len = act[i]->read_len;
		}
		(void)ntc::plan_sketch_k1h(hs, na, k, (unsigned)di.cus, planned);
		// K1h + K1f: the two bit arrays between them and the suspect list are scratch of a launch pair; the sets are sized for the most demanding k (above)
		const size_t need_d = need_d_all, need_t = need_t_all;
		uint32_t sus_cap = sus_cap_all;
		if (const char* ev = std::getenv("NTC_K1H_SUS_CAP")) { // tests: a short list forces the overflow path
			const long v = std::strtol(ev, nullptr, 10);
			if (v >= 1 && v <= (long)sus_cap) sus_cap = (uint32_t)v;
		}
		// the hand-over arrays of one set; an engine that defers K1f sizes ALL its sets at the first launch of a batch geometry (a set that grows
		// later would stall the stream in the middle of a run)
		auto ensure_set = [&](ntc_engine::K1hSet& s2, bool with_sus) -> int { // with_sus: the first set of a launch holds the launch's suspect list
			if (need_d > s2.dirty_cap || need_t > s2.tie_cap || (with_sus && sus_cap > s2.sus_cap)) HIP_TRY(hipStreamSynchronize(e->stream));
			if (need_d > s2.dirty_cap || need_t > s2.tie_cap) {
				if (s2.d_dirty) (void)hipFree(s2.d_dirty);
				if (s2.d_tie) (void)hipFree(s2.d_tie);
				s2.d_dirty = s2.d_tie = nullptr;
				s2.dirty_cap = s2.tie_cap = 0;
				if (hipMalloc((void**)&s2.d_dirty, need_d) != hipSuccess || hipMalloc((void**)&s2.d_tie, need_t) != hipSuccess)
					return fail(NTC_ERR_MEMORY, "cannot allocate %zu B of scratch for the tiled kernel on device", need_d + need_t);
				s2.dirty_cap = need_d;
				s2.tie_cap = need_t;
			}
			if (with_sus && sus_cap > s2.sus_cap) {
				if (s2.d_sus) (void)hipFree(s2.d_sus);
				s2.d_sus = nullptr;
				s2.sus_cap = 0;
				if (hipMalloc((void**)&s2.d_sus, (size_t)max_waves * sus_cap * 16) != hipSuccess)
					return fail(NTC_ERR_MEMORY, "cannot allocate the %zu-byte suspect list of the tiled kernel on device", (size_t)max_waves * sus_cap * 16);
				s2.sus_cap = sus_cap;
			}
			if (!s2.d_sus_count) {
				if (hipMalloc((void**)&s2.d_sus_count, (size_t)max_waves * 4) != hipSuccess || hipMalloc((void**)&s2.d_fix_state, 16) != hipSuccess)
					return fail(NTC_ERR_MEMORY, "cannot allocate the suspect list of the tiled kernel on device");
				HIP_TRY(hipMemsetAsync(s2.d_fix_state, 0, 16, e->stream));
				HIP_TRY(hipMemsetAsync(s2.d_sus_count, 0, (size_t)max_waves * 4, e->stream));
			}
			return 0;
		};
		bool grow = false;
		for (uint32_t i = 0; i < na; ++i) {
			const auto& ks = e->k1h_set[e->k1f_n + i];
			grow |= need_d > ks.dirty_cap || need_t > ks.tie_cap || (i == 0 && sus_cap > ks.sus_cap) || !ks.d_sus_count;
		}
		if (grow) {
			if (e->k1f_n != 0) // the sets ahead are in use by launches whose K1f is still to come
				if (int rc = join_k1f(e)) return rc;
			for (uint32_t si = 0; si < (e->defer_redo ? ntc::kK1fBatch : na); ++si)
				if (int rc = ensure_set(e->k1h_set[si], si == 0 || na == 1)) { // (launches of one batch each — K1's share of a list, single submits — may start at any set)
					// no memory for K1h's hand-over arrays (8 sets with NTC_FLAG_DEFER_REDO: up to ~0.5 GB each per 10 M reads): the
					// batches are K1's, unless the caller insists on the tiled kernels or part of the k list has been launched already
					if (e->ts_required || launched_any || any_tails) return rc;
					if (int rc2 = close_run(e)) return rc2;
					for (const auto& sg : segs)
						if (int rc3 = run_tiled_as_rows(e, sg.d_tiles, sg.n_reads, sg.read_len)) return rc3;
					return 0;
				}
		}
		for (uint32_t i = 0; i < na; ++i) {
			auto& ks0 = e->k1h_set[e->k1f_n + i]; // (k1f_n may be 0 now)
			ntc::K1hArgs& h = hs[i];
			h.sus = e->k1h_set[e->k1f_n].d_sus; // (the launch's list: that of its first set)
			h.sus_count = e->k1h_set[e->k1f_n].d_sus_count;
			h.sus_cap = sus_cap; // (<= the allocation's)
			h.launch_id = ++e->k1h_launch_id;
			if (h.launch_id == 0) h.launch_id = ++e->k1h_launch_id;
			h.fix_state = ks0.d_fix_state;
			h.tiles = act[i]->d_tiles;
			h.log = e->d_log;
			h.log_fill = e->d_logfill;
			h.sketch0 = e->d_sketch;
			h.sk_dirty = e->d_skdirty;
			h.f1 = e->d_f1 + ki;
			h.dirty = ks0.d_dirty;
			h.tie = ks0.d_tie;
			h.n_chunks = (act[i]->read_len + 15u) / 16u;
			h.nv_last = (uint32_t)(act[i]->n_reads - (uint64_t)(h.n_tiles - 1) * ntc::kTileReads);
			h.key_base = (uint32_t)(ki * e->plane_elems());
			h.rmask2 = (uint32_t)((2ull << e->r_bits) - 1ull);
			h.log_regions = e->d_log ? e->log_regions : 0u;
			h.log_region_cap = e->log_region_cap;
			h.table = e->d_k1h_tabs[ki];
			h.s_bits = e->s_bits;
			h.r_bits = e->r_bits;
			h.tails = act[i]->d_tails;
		}
		ntc::K1hArgs launched[ntc::kK1hSegs];
		uint32_t n_waves = 0;
		if (int rc = open_run()) return rc; // (a K1f above may have closed the bracket)
		HIP_TRY(ntc::launch_sketch_k1h_multi(hs, na, k, e->gap, (unsigned)di.cus, e->stream, launched, &n_waves));
		launched_any = true;
		for (uint32_t i = 0; i < na; ++i) {
			auto& it = e->k1f_batch.item[e->k1f_n++];
			it.a = launched[i];
			it.t4 = e->d_t4s[ki];
			it.k = k;
			it.n_waves = launched[i].n_wg * wpg; // (its suspect regions: those of the workgroups that walked it)
			it.klog = e->d_log ? e->d_log + (size_t)e->log_regions * e->log_region_cap : nullptr;
			it.klog_fill = e->d_log ? e->d_logfill + e->log_regions : nullptr;
			it.klog_n = e->d_log ? e->klog_regions : 0u;
			it.klog_cap = e->log_region_cap;
		}
		if (!e->defer_redo) // the caller may change the batches once the stream has passed this call: K1f now
			if (int rc = join_k1f(e)) return rc;
	}
	if (!e->profiling)
		if (int rc = close_run(e)) return rc; // (profiling was switched off inside a run)
	if (!e->ts_all) // the k of the list K1h is not built for: K1 over the same tiles (staged straight from the tiled layout)
		for (const auto& sg : segs) {
			const uint32_t* d_meta = nullptr;
			if (sg.d_tails) {
				// a ragged batch (round 6): K1 takes every read's length from a slot table, built here from the tiles' prefix tables — the batch stays on tiles,
				// K1h + K1f serve their k from it and K1 the rest (before: the whole batch went to row slots and K1 for every k)
				if (sg.n_reads > e->tmeta_cap) {
					HIP_TRY(hipStreamSynchronize(e->stream));
					if (e->d_tmeta) (void)hipFree(e->d_tmeta);
					e->d_tmeta = nullptr;
					e->tmeta_cap = 0;
					const size_t cap = std::max<size_t>((size_t)sg.n_reads, 1u << 20);
					if (hipMalloc((void**)&e->d_tmeta, cap * 4) != hipSuccess) return fail(NTC_ERR_MEMORY, "cannot allocate the slot table of a ragged tiled batch on device");
					e->tmeta_cap = cap;
				}
				HIP_TRY(ntc::launch_tails_to_meta(sg.d_tails, sg.n_reads, (sg.read_len + 15u) / 16u, e->d_tmeta, e->stream));
				d_meta = e->d_tmeta;
			}
			if (int rc = run_batch(e, sg.d_tiles, d_meta, sg.n_reads, sg.read_len, 16u * ((sg.read_len + 15u) / 16u), true, &e->k_tiled)) return rc;
		}
	return 0;
}

int flush_deferred(ntc_engine* e)
{
	if (e->deferred.empty() || e->in_flush) return 0;
	e->in_flush = true; // (run_tiled_segs calls join_k1f itself when it runs out of hand-over sets)
	std::vector<TiledSeg> segs;
	segs.reserve(e->deferred.size());
	for (const auto& d : e->deferred)
		segs.push_back(TiledSeg{d.d_tiles, d.n_reads, d.read_len, d.d_tails});
	e->deferred.clear();
	const int rc = run_tiled_segs(e, segs.data(), (uint32_t)segs.size(), segs.size());
	e->in_flush = false;
	return rc;
}

// a device-resident tiled batch under NTC_FLAG_DEFER_REDO, every k K1h's: it waits for up to seven more (one K1h launch per k over all of them)
int defer_or_run_tiled(ntc_engine* e, const unsigned char* d_tiles, uint64_t n_reads, uint32_t read_len, const uint32_t* d_tails)
{
	if (!(e->defer_redo && e->ts_all)) return run_tiled(e, d_tiles, n_reads, read_len, d_tails);
	e->deferred.push_back(ntc_engine::DeferredSeg{d_tiles, n_reads, read_len, d_tails});
	if (e->deferred.size() >= std::min<size_t>(ntc::kK1hSegs, ntc::kK1fBatch)) return flush_deferred(e);
	return 0;
}

} // namespace

extern "C" {

uint32_t ntc_abi_version(void) { return NTC_ABI_VERSION; }
uint32_t ntc_max_k(void) { return kMaxK; }
const char* ntc_last_error(void) { return g_err.c_str(); }

int ntc_create(const ntc_config* cfg, ntc_engine** out)
{
	if (!cfg || !out) return fail(NTC_ERR_ARG, "ntc_create: null argument");
	*out = nullptr;
	if (cfg->n_k == 0 || cfg->n_k > NTC_MAX_K_LIST || !cfg->k)
		return fail(NTC_ERR_ARG, "ntc_create: need 1..%d k values", NTC_MAX_K_LIST);
	constexpr uint32_t kKnownFlags = NTC_FLAG_SIMPLE_KERNEL | NTC_FLAG_DIRECT_ATOMICS | NTC_FLAG_ALWAYS_LOG | NTC_FLAG_PARTITION_ALWAYS | NTC_FLAG_LANE_KERNEL |
	                                 NTC_FLAG_REQUIRE_TILED | NTC_FLAG_DEFER_REDO;
	if (cfg->flags & ~kKnownFlags) // (ABI 4's NTC_FLAG_BITSLICE_KERNEL = 4 and NTC_FLAG_TILED_TEAMS = 256 selected kernels that no longer exist)
		return fail(NTC_ERR_ARG, "ntc_create: unknown flag bits 0x%x", cfg->flags & ~kKnownFlags);
	for (uint32_t i = 0; i < cfg->n_k; ++i)
		if (cfg->k[i] < 1 || cfg->k[i] > kMaxK)
			return fail(NTC_ERR_ARG, "ntc_create: k=%u outside 1..%u", cfg->k[i], kMaxK);
	if (cfg->gap != 0) {
		if (cfg->n_k != 1) return fail(NTC_ERR_ARG, "ntc_create: gap seed does not support multiple k");
		if (cfg->gap % 2 != cfg->k[0] % 2 || cfg->gap >= cfg->k[0])
			return fail(NTC_ERR_ARG, "ntc_create: gap size and kmer must have the same modulus");
		if (cfg->flags & NTC_FLAG_SIMPLE_KERNEL)
			return fail(NTC_ERR_ARG, "ntc_create: spaced seeds need the production kernel");
	}
	if (cfg->r_bits < 8 || cfg->r_bits > 30) return fail(NTC_ERR_ARG, "ntc_create: r_bits %u outside 8..30", cfg->r_bits);
	if (cfg->s_bits < 2 || cfg->s_bits > 24) return fail(NTC_ERR_ARG, "ntc_create: s_bits %u outside 2..24", cfg->s_bits);
	if (cfg->log_entries > (1ull << 32)) return fail(NTC_ERR_ARG, "ntc_create: log_entries %llu above 2^32", (unsigned long long)cfg->log_entries);
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
		return fail(NTC_ERR_DEVICE, "ntc_create: no HIP device available (this library has no CPU fallback)");
	if (cfg->device < 0 || cfg->device >= ndev) return fail(NTC_ERR_ARG, "ntc_create: device %d of %d", cfg->device, ndev);
	HIP_TRY(hipSetDevice(cfg->device));
	if (int rc = ensure_kernel_attrs(cfg->device)) return rc;
	ntc_engine* e = new (std::nothrow) ntc_engine();
	if (!e) return fail(NTC_ERR_MEMORY, "ntc_create: out of host memory");
	e->device = cfg->device;
	e->stream = (hipStream_t)cfg->stream;
	e->klist.assign(cfg->k, cfg->k + cfg->n_k);
	e->gap = cfg->gap;
	e->r_bits = cfg->r_bits;
	e->s_bits = cfg->s_bits;
	e->kernel_kind = (cfg->flags & NTC_FLAG_SIMPLE_KERNEL) ? KIND_SIMPLE : KIND_HF;
	const size_t sk_bytes = e->klist.size() * e->plane_elems() * sizeof(uint32_t);
	if (cfg->ext_sketch) {
		e->d_sketch = (uint32_t*)cfg->ext_sketch;
	} else {
		if (hipMalloc((void**)&e->d_sketch, sk_bytes) != hipSuccess) {
			delete e;
			return fail(NTC_ERR_MEMORY, "ntc_create: cannot allocate %zu B sketch on device", sk_bytes);
		}
		e->own_sketch = true;
	}
	if (cfg->ext_f1) {
		e->d_f1 = (unsigned long long*)cfg->ext_f1;
	} else {
		if (hipMalloc((void**)&e->d_f1, e->klist.size() * 8) != hipSuccess) {
			ntc_destroy(e);
			return fail(NTC_ERR_MEMORY, "ntc_create: cannot allocate F1 on device");
		}
		e->own_f1 = true;
	}
	if (hipMalloc((void**)&e->d_skdirty, 64 + 2 * 64 * 8) != hipSuccess) { // (the word, then ntc_log_export_device's cursors and offsets)
		ntc_destroy(e);
		return fail(NTC_ERR_MEMORY, "ntc_create: cannot allocate engine state on device");
	}
	if (hipMalloc((void**)&e->d_phist, e->klist.size() * 2 * 65536 * 4) != hipSuccess) {
		ntc_destroy(e);
		return fail(NTC_ERR_MEMORY, "ntc_create: cannot allocate histogram on device");
	}
	for (size_t ki = 0; ki < e->klist.size(); ++ki) {
		std::vector<uint32_t> t1((size_t)ntc::t2_pairs(e->klist[ki]) * 64);
		const uint32_t gap_first = (e->klist[ki] - e->gap) / 2; // "1"x(k-g)/2 "0"xg "1"x(k-g)/2, ntcard.cpp:407-413
		ntc::build_t2(e->klist[ki], t1.data(), gap_first, e->gap);
		void* d = nullptr;
		if (hipMalloc(&d, t1.size() * 4) != hipSuccess || hipMemcpy(d, t1.data(), t1.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
			ntc_destroy(e);
			return fail(NTC_ERR_MEMORY, "ntc_create: cannot allocate the closed-form seed table on device");
		}
		e->d_t1.push_back(d);
	}
	if (e->gap) {
		std::vector<uint32_t> gt((size_t)((e->gap + 1) / 2) * 64);
		ntc::build_gap_table(e->klist[0], (e->klist[0] - e->gap) / 2, e->gap, gt.data());
		if (hipMalloc(&e->d_gapt, gt.size() * 4) != hipSuccess || hipMemcpy(e->d_gapt, gt.data(), gt.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
			ntc_destroy(e);
			return fail(NTC_ERR_MEMORY, "ntc_create: cannot allocate the spaced-seed table on device");
		}
	}
	e->adaptive = !(cfg->flags & NTC_FLAG_ALWAYS_LOG);
	e->partition_always = (cfg->flags & NTC_FLAG_PARTITION_ALWAYS) != 0;
	if (e->kernel_kind == KIND_HF && !(cfg->flags & NTC_FLAG_DIRECT_ATOMICS)) {
		// The log and the two partition work areas of the same total size (at the default of four entries per counter and rBits = 27:
		// 4 + 5.4 + 7 GiB) are allocated HERE, so that memory runs out at create time and not in the middle of a run; when it does, the
		// capacity is halved until it fits, and an engine that cannot even hold 2^18 entries increments with device atomics instead.
		uint64_t want = cfg->log_entries;
		while (plan_log(e, want)) {
			const auto& ap = e->ap;
			const size_t runs1 = ap.b1 ? (size_t)ap.g1 << ap.b1 : 0, runs2 = ap.b2 ? ((size_t)ap.parts2 << ap.b1) << ap.b2 : 0;
			bool ok = hipMalloc((void**)&e->d_log, (size_t)e->all_log_regions() * e->log_region_cap * 4) == hipSuccess &&
			          hipMalloc((void**)&e->d_logfill, (size_t)e->all_log_regions() * 4) == hipSuccess;
			ok = ok && (e->d_logmode || hipMalloc((void**)&e->d_logmode, 4) == hipSuccess) && (e->d_logstats || hipMalloc((void**)&e->d_logstats, 24) == hipSuccess) &&
			     (e->d_probe || hipMalloc((void**)&e->d_probe, 4u << 20) == hipSuccess);
			ok = ok && (!runs1 || (hipMalloc((void**)&e->d_s1, runs1 * ap.cap1 * ap.key_bytes(1)) == hipSuccess && hipMalloc((void**)&e->d_c1, runs1 * 4) == hipSuccess));
			ok = ok && (!runs2 || (hipMalloc((void**)&e->d_s2, runs2 * ap.cap2 * ap.key_bytes(2)) == hipSuccess && hipMalloc((void**)&e->d_c2, runs2 * 4) == hipSuccess));
			if (ok) break;
			(void)hipGetLastError();
			for (void** d : {(void**)&e->d_log, (void**)&e->d_logfill, (void**)&e->d_s1, (void**)&e->d_c1, (void**)&e->d_s2, (void**)&e->d_c2})
				if (*d) {
					(void)hipFree(*d);
					*d = nullptr;
				}
			if (cfg->log_entries != 0 || e->log_cap <= (1ull << 18)) { // an explicit request that does not fit is an error; otherwise: no log
				if (cfg->log_entries != 0) {
					const unsigned long long asked = (unsigned long long)e->log_cap;
					ntc_destroy(e);
					return fail(NTC_ERR_MEMORY, "ntc_create: cannot allocate the %llu-entry hit log and its partition areas on device", asked);
				}
				e->log_regions = e->log_region_cap = e->klog_regions = 0;
				e->log_cap = 0;
				break;
			}
			want = e->log_cap / 2;
		}
	}
	// The tiled kernel pair K1h + K1f: every k of the list must be one K1h is generated for (k = 12 .. 32; ntcard's -g seed at k = 12 / gap 2 and k = 32 / gap 8); a list is
	// served by one launch per k over the same resident tiles.  Its hit-log keys and K1f's atomics are 32-bit counter indices.  Everything else —
	// row slots, other k, other seeds, nthll — is K1's (NTC_FLAG_LANE_KERNEL: tiled batches too, re-laid out as row slots).
	// A list may mix both kinds (round 5: `-k 16,24,32,48`, BASELINE config 4's `32,64,96,128`): K1h takes its k from the tiles, K1 stages the same tiles for the rest.
	const bool ts_pre = e->kernel_kind == KIND_HF && !(cfg->flags & NTC_FLAG_LANE_KERNEL) && e->hll_bits == 0 && e->klist.size() * e->plane_elems() <= (1ull << 32);
	e->k_tiled.assign(e->klist.size(), 0);
	e->ts_ok = false;
	e->ts_all = ts_pre;
	for (size_t ki = 0; ki < e->klist.size(); ++ki) {
		e->k_tiled[ki] = ts_pre && ntc::sketch_k1h_supports(e->klist[ki], e->gap, e->s_bits, e->r_bits) ? 1 : 0;
		e->ts_ok = e->ts_ok || e->k_tiled[ki];
		e->ts_all = e->ts_all && e->k_tiled[ki];
	}
	e->d_k1h_tabs.assign(e->klist.size(), nullptr);
	e->d_t4s.assign(e->klist.size(), nullptr);
	for (size_t ki = 0; ki < e->klist.size(); ++ki) {
		if (!e->k_tiled[ki]) continue;
		const uint32_t k = e->klist[ki];
		std::vector<uint32_t> t4((size_t)ntc::t4_groups(k) * 256 * 4); // K1f: both strands' 64-bit terms, 4 bases per entry
		ntc::build_t4(k, t4.data(), (k - e->gap) / 2, e->gap);
		std::vector<uint32_t> tab((size_t)2 * ((k + 2) / 3) * 64);     // K1h's resolve pass: the low r_bits + sample bits, 3 bases per entry
		ntc::build_k1h_table(k, e->gap, e->r_bits, e->s_bits, tab.data());
		void* d4 = nullptr;
		uint32_t* d3 = nullptr;
		if (hipMalloc(&d4, t4.size() * 4) != hipSuccess || hipMemcpy(d4, t4.data(), t4.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
		    hipMalloc((void**)&d3, tab.size() * 4) != hipSuccess || hipMemcpy(d3, tab.data(), tab.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
			if (d4) (void)hipFree(d4);
			if (d3) (void)hipFree(d3);
			ntc_destroy(e);
			return fail(NTC_ERR_MEMORY, "ntc_create: cannot allocate the closed-form tables of the tiled kernels on device");
		}
		e->d_t4s[ki] = d4;
		e->d_k1h_tabs[ki] = d3;
	}
	e->ts_required = (cfg->flags & NTC_FLAG_REQUIRE_TILED) != 0;
	e->defer_redo = (cfg->flags & NTC_FLAG_DEFER_REDO) != 0;
	e->hfk.resize(e->klist.size());
	for (size_t ki = 0; ki < e->klist.size(); ++ki)
		fill_hfk(e->hfk[ki], e->klist[ki], e->d_sketch + ki * e->plane_elems(), e->d_f1 + ki, e->d_t1[ki], (uint32_t)(ki * e->plane_elems()));
	int rc = ntc_reset(e);
	if (rc) {
		ntc_destroy(e);
		return rc;
	}
	*out = e;
	return 0;
}

void ntc_destroy(ntc_engine* e)
{
	if (!e) return;
	(void)hipSetDevice(e->device);
	(void)hipStreamSynchronize(e->stream);
	for (auto& pr : e->pending) {
		if (!e->borrowed.count(pr.first)) (void)hipEventDestroy(pr.first);
		(void)hipEventDestroy(pr.second);
	}
	if (e->run_ev0 && !e->borrowed.count(e->run_ev0)) (void)hipEventDestroy(e->run_ev0);
	if (e->own_sketch && e->d_sketch) (void)hipFree(e->d_sketch);
	if (e->own_f1 && e->d_f1) (void)hipFree(e->d_f1);
	if (e->d_phist) (void)hipFree(e->d_phist);
	if (e->d_out16) (void)hipFree(e->d_out16);
	if (e->d_untile) (void)hipFree(e->d_untile);
	if (e->d_tmeta) (void)hipFree(e->d_tmeta);
	if (e->d_skdirty) (void)hipFree(e->d_skdirty);
	for (void* d : {(void*)e->d_log, (void*)e->d_logfill, (void*)e->d_s1, (void*)e->d_c1, (void*)e->d_s2, (void*)e->d_c2, (void*)e->d_logmode, (void*)e->d_logstats, (void*)e->d_probe})
		if (d) (void)hipFree(d);
	for (auto& pr : e->apply_pending) {
		if (!e->borrowed.count(pr.first)) (void)hipEventDestroy(pr.first);
		(void)hipEventDestroy(pr.second);
	}
	for (void* d : e->d_t4s)
		if (d) (void)hipFree(d);
	for (uint32_t* d : e->d_k1h_tabs)
		if (d) (void)hipFree(d);
	for (hipStream_t st : e->mc.lanes)
		if (st) {
			(void)hipStreamSynchronize(st);
			(void)hipStreamDestroy(st);
		}
	for (hipEvent_t ev : e->mc.arrived)
		if (ev) (void)hipEventDestroy(ev);
	if (e->mc.narrowed) (void)hipEventDestroy(e->mc.narrowed);
	if (e->mc.summed) (void)hipEventDestroy(e->mc.summed);
	if (e->mc.narrow) (void)hipFree(e->mc.narrow);
	if (e->mc.recv) (void)hipFree(e->mc.recv);
	for (auto& ks : e->k1h_set)
		for (void* d : {(void*)ks.d_dirty, (void*)ks.d_tie, (void*)ks.d_sus, (void*)ks.d_sus_count, (void*)ks.d_fix_state})
			if (d) (void)hipFree(d);
	for (auto& pr : e->k1f_events) {
		if (!e->borrowed.count(pr.first)) (void)hipEventDestroy(pr.first);
		(void)hipEventDestroy(pr.second);
	}
	for (void* d : e->d_t1) (void)hipFree(d);
	if (e->d_gapt) (void)hipFree(e->d_gapt);
	if (e->d_hll_thr) (void)hipFree(e->d_hll_thr);
	for (auto& sl : e->stage) {
		if (sl.d_stage) (void)hipFree(sl.d_stage);
		if (sl.d_meta) (void)hipFree(sl.d_meta);
		if (sl.h_stage) (void)hipHostFree(sl.h_stage);
		if (sl.h_meta) (void)hipHostFree(sl.h_meta);
		if (sl.done) (void)hipEventDestroy(sl.done);
	}
	delete e;
}

int ntc_reset(ntc_engine* e)
{
	if (!e) return fail(NTC_ERR_ARG, "ntc_reset: null engine");
	std::lock_guard<std::mutex> lk(e->mu);
	HIP_TRY(hipSetDevice(e->device));
	if (int rc = join_k1f(e)) return rc;
	HIP_TRY(hipMemsetAsync(e->d_sketch, 0, e->hll_bits ? (sizeof(uint32_t) << e->hll_bits) : e->klist.size() * e->plane_elems() * sizeof(uint32_t), e->stream));
	e->hll_reads_seen = 0;
	if (e->d_logfill) HIP_TRY(hipMemsetAsync(e->d_logfill, 0, (size_t)e->all_log_regions() * 4, e->stream));
	if (e->d_skdirty) HIP_TRY(hipMemsetAsync(e->d_skdirty, 0, 4, e->stream));
	e->sk_host_dirty = e->sk_exposed; // (a caller that has asked for the counters' address may write there at any time; ext_sketch: include/ntcard_hip.h)
	if (e->d_logmode) {
		HIP_TRY(hipMemsetAsync(e->d_logmode, 0, 4, e->stream));
		HIP_TRY(hipMemsetAsync(e->d_logstats, 0, 24, e->stream));
	}
	e->probed = false;
	e->log_pending = false;
	e->log_est = 0.0;
	HIP_TRY(hipMemsetAsync(e->d_f1, 0, e->klist.size() * 8, e->stream));
	HIP_TRY(hipStreamSynchronize(e->stream));
	if (int rc = drain_events(e)) return rc;
	e->ms_total = 0.0;
	e->launches = 0;
	e->apply_ms = 0.0;
	e->applies = 0;
	e->k1f_ms = 0.0;
	return 0;
}

int ntc_submit_device(ntc_engine* e, const void* d_slots, uint64_t n_reads, uint32_t read_len, uint32_t stride)
{
	if (!e) return fail(NTC_ERR_ARG, "ntc_submit_device: null engine");
	if (n_reads == 0) return 0;
	if (!d_slots || (stride & 3u) || stride < read_len || ((uintptr_t)d_slots & 15u))
		return fail(NTC_ERR_ARG, "ntc_submit_device: need 16-byte aligned slots, stride %% 4 == 0, stride >= read_len");
	if (read_len > 0xffffu) return fail(NTC_ERR_ARG, "ntc_submit_device: read_len %u > 65535", read_len);
	std::lock_guard<std::mutex> lk(e->mu);
	HIP_TRY(hipSetDevice(e->device));
	return run_batch(e, (const unsigned char*)d_slots, nullptr, n_reads, read_len, stride);
}


int ntc_submit_tiled_device(ntc_engine* e, const void* d_tiles, uint64_t n_reads, uint32_t read_len)
{
	if (!e) return fail(NTC_ERR_ARG, "ntc_submit_tiled_device: null engine");
	if (n_reads == 0) return 0;
	if (!d_tiles || ((uintptr_t)d_tiles & 15u)) return fail(NTC_ERR_ARG, "ntc_submit_tiled_device: need a 16-byte aligned buffer");
	if (read_len == 0 || read_len > 0xffffu) return fail(NTC_ERR_ARG, "ntc_submit_tiled_device: read_len %u outside 1..65535", read_len);
	std::lock_guard<std::mutex> lk(e->mu);
	HIP_TRY(hipSetDevice(e->device));
	return defer_or_run_tiled(e, (const unsigned char*)d_tiles, n_reads, read_len, nullptr);
}

int ntc_submit_tiled_ragged_device(ntc_engine* e, const void* d_tiles, uint64_t n_reads, uint32_t n_chunks, const uint32_t* d_tails)
{
	if (!e) return fail(NTC_ERR_ARG, "ntc_submit_tiled_ragged_device: null engine");
	if (n_reads == 0) return 0;
	if (!d_tiles || ((uintptr_t)d_tiles & 15u) || !d_tails || ((uintptr_t)d_tails & 3u)) return fail(NTC_ERR_ARG, "ntc_submit_tiled_ragged_device: need a 16-byte aligned tile buffer and a tails array");
	if (n_chunks == 0 || n_chunks > 0xffffu / 16u) return fail(NTC_ERR_ARG, "ntc_submit_tiled_ragged_device: n_chunks %u outside 1..4095", n_chunks);
	std::lock_guard<std::mutex> lk(e->mu);
	HIP_TRY(hipSetDevice(e->device));
	return defer_or_run_tiled(e, (const unsigned char*)d_tiles, n_reads, 16u * n_chunks, d_tails);
}

int ntc_submit_tiled_bins_device(ntc_engine* e, uint32_t n_bins, const void* const* d_tiles, const uint64_t* n_reads, const uint32_t* read_len,
                                 const uint32_t* const* d_tails)
{
	if (!e) return fail(NTC_ERR_ARG, "ntc_submit_tiled_bins_device: null engine");
	if (n_bins == 0) return 0;
	if (!d_tiles || !n_reads || !read_len) return fail(NTC_ERR_ARG, "ntc_submit_tiled_bins_device: null argument"); // (d_tails == NULL: every bin is equal-length)
	std::vector<TiledSeg> segs;
	for (uint32_t i = 0; i < n_bins; ++i) {
		if (n_reads[i] == 0) continue;
		const uint32_t* tails_i = d_tails ? d_tails[i] : nullptr;
		if (!d_tiles[i] || ((uintptr_t)d_tiles[i] & 15u) || ((uintptr_t)tails_i & 3u)) return fail(NTC_ERR_ARG, "ntc_submit_tiled_bins_device: bin %u: need a 16-byte aligned tile buffer", i);
		if (read_len[i] == 0 || read_len[i] > 0xffffu || (tails_i && (read_len[i] & 15u)))
			return fail(NTC_ERR_ARG, "ntc_submit_tiled_bins_device: bin %u: read_len %u (a ragged bin's is 16 x its chunks, at most 65520)", i, read_len[i]);
		segs.push_back(TiledSeg{(const unsigned char*)d_tiles[i], n_reads[i], read_len[i], tails_i});
	}
	if (segs.empty()) return 0;
	std::lock_guard<std::mutex> lk(e->mu);
	HIP_TRY(hipSetDevice(e->device));
	return run_tiled_segs(e, segs.data(), (uint32_t)segs.size());
}

uint64_t ntc_tiled_bytes(uint64_t n_reads, uint32_t read_len)
{
	const uint64_t n_tiles = (n_reads + ntc::kTileReads - 1) / ntc::kTileReads;
	return n_tiles * ((read_len + 15u) / 16u) * (uint64_t)ntc::kTileReads * 16u;
}

int ntc_gen_reads_tiled_device(int32_t device, void* stream, void* d_tiles, uint64_t seed, uint64_t first_read, uint64_t n_reads, uint32_t read_len,
                               uint32_t dist, uint64_t genome_len)
{
	if (!d_tiles || ((uintptr_t)d_tiles & 15u) || read_len == 0) return fail(NTC_ERR_ARG, "ntc_gen_reads_tiled_device: bad layout");
	if (dist > 1) return fail(NTC_ERR_ARG, "ntc_gen_reads_tiled_device: dist must be 0 (uniform) or 1 (genome)");
	if (dist == 1 && genome_len < read_len) return fail(NTC_ERR_ARG, "ntc_gen_reads_tiled_device: genome shorter than a read");
	if (n_reads == 0) return 0;
	HIP_TRY(hipSetDevice(device));
	HIP_TRY(ntc::launch_gen_tiled((unsigned char*)d_tiles, seed, first_read, n_reads, read_len, dist, genome_len, (hipStream_t)stream));
	return 0;
}
} // extern "C"

namespace {
// read i = bytes [ptr_of(i), ptr_of(i) + len_of(i)): ntc_submit (concatenated reads + offsets) and ntc_submit_spans (spans of
// a caller buffer, e.g. the sequence lines inside a block of a FASTQ file) pack into the pinned staging pair the same way
// Equal-length reads of a host batch go to the device in the TILED layout (round 4): the packing loop writes each read's 16-byte pieces
// where ntc_submit_tiled_device expects them, so the reads the reference's parsers hand to ntRead (ntcard.cpp:182,203,230) reach the
// tiled kernels (K1h / K1c) like a device-resident producer's do.
// (len_of / ptr_of are template callables: the packing loops call them once or twice per read, inlined)
// ragged == false: every read is `len` bases long.  ragged == true (round 5): the reads are 16 C - 15 .. 16 C bases long, C = len / 16, and come
// LONGEST FIRST (so every tile is sorted): the tiles are followed, in the same staging buffer, by tails[tile][16] — the reads of the tile with more than d
// bases in their last piece — and the batch goes to ntc_submit_tiled_ragged_device's path.
// A host batch may hold several bins (HostBin: reads idx[0 .. n) of the caller's numbering, or 0 .. n when idx == nullptr): they are packed one behind the other into ONE
// staging buffer, copied once and hashed by ONE launch per k (run_tiled_segs).
struct HostBin {
	const uint64_t* idx;
	uint64_t n;
	uint32_t len;  // every read's length, or 16 C for a ragged bin
	bool ragged;
};
template <class LenFn, class PtrFn> int submit_tiled_host(ntc_engine* e, const HostBin* bins, uint32_t n_bins, const LenFn& len_of, const PtrFn& ptr_of)
{
	HIP_TRY(hipSetDevice(e->device));
	ntc_engine::StageSlot* sl = nullptr;
	{
		std::unique_lock<std::mutex> lk(e->stage_mu);
		e->stage_cv.wait(lk, [&] {
			for (auto& c : e->stage)
				if (!c.busy) return true;
			return false;
		});
		for (auto& c : e->stage)
			if (!c.busy) {
				sl = &c;
				break;
			}
		sl->busy = true;
	}
	struct Release {
		ntc_engine* e;
		ntc_engine::StageSlot* sl;
		~Release()
		{
			{
				std::lock_guard<std::mutex> lk(e->stage_mu);
				sl->busy = false;
			}
			e->stage_cv.notify_one();
		}
	} release{e, sl};
	if (sl->done == nullptr) HIP_TRY(hipEventCreateWithFlags(&sl->done, hipEventDisableTiming));
	if (sl->used) HIP_TRY(hipEventSynchronize(sl->done));
	// sections: [tiles of bin 0][tails of bin 0][tiles of bin 1] ... (tile sections are multiples of 32 KiB, tails of 64 B: everything stays 16-byte aligned)
	std::vector<size_t> off(n_bins), tile_bytes(n_bins);
	size_t need = 0;
	for (uint32_t b = 0; b < n_bins; ++b) {
		off[b] = need;
		tile_bytes[b] = (size_t)ntc_tiled_bytes(bins[b].n, bins[b].len);
		need += tile_bytes[b] + (bins[b].ragged ? (size_t)((bins[b].n + ntc::kTileReads - 1) / ntc::kTileReads) * 64 : 0);
	}
	if (need > sl->h_stage_cap) {
		if (sl->h_stage) (void)hipHostFree(sl->h_stage);
		sl->h_stage = nullptr;
		size_t cap = std::max(need, sl->h_stage_cap * 2);
		if (hipHostMalloc((void**)&sl->h_stage, cap, hipHostMallocDefault) != hipSuccess) {
			sl->h_stage_cap = 0;
			return fail(NTC_ERR_MEMORY, "ntc_submit: cannot pin %zu B", cap);
		}
		sl->h_stage_cap = cap;
	}
	if (need > sl->d_stage_cap) {
		if (sl->d_stage) (void)hipFree(sl->d_stage);
		sl->d_stage = nullptr;
		size_t cap = std::max(need, sl->d_stage_cap * 2);
		if (hipMalloc((void**)&sl->d_stage, cap) != hipSuccess) {
			sl->d_stage_cap = 0;
			return fail(NTC_ERR_MEMORY, "ntc_submit: cannot allocate %zu B on device", cap);
		}
		sl->d_stage_cap = cap;
	}
	// ---- pack: piece c of read i of a bin -> ((tile * C + c) * 2048 + i % 2048) * 16 of the bin's section ----
	for (uint32_t b = 0; b < n_bins; ++b) {
		const HostBin& hb = bins[b];
		unsigned char* hs = sl->h_stage + off[b];
		const uint32_t C = (hb.len + 15u) / 16u;
		uint32_t* tails = reinterpret_cast<uint32_t*>(hs + tile_bytes[b]);
		if (hb.ragged) std::memset(tails, 0, (size_t)((hb.n + ntc::kTileReads - 1) / ntc::kTileReads) * 64);
		for (uint64_t i = 0; i < hb.n; ++i) {
			const uint64_t r = hb.idx ? hb.idx[i] : i;
			const char* src = ptr_of(r);
			const uint32_t tail = (uint32_t)(hb.ragged ? len_of(r) : hb.len) - (C - 1u) * 16u; // 1 .. 16 bases in the last piece
			unsigned char* dst = hs + ((i / ntc::kTileReads) * C * ntc::kTileReads + i % ntc::kTileReads) * 16u;
			for (uint32_t c = 0; c + 1u < C; ++c)
				std::memcpy(dst + (size_t)c * ntc::kTileReads * 16u, src + 16u * c, 16);
			unsigned char* last = dst + (size_t)(C - 1u) * ntc::kTileReads * 16u;
			std::memcpy(last, src + 16u * (C - 1u), tail);
			std::memset(last + tail, 'A', 16u - tail);
			if (hb.ragged)
				for (uint32_t d = 0; d < tail; ++d)
					++tails[(i / ntc::kTileReads) * 16u + d];
		}
	}
	{
		std::lock_guard<std::mutex> lk(e->mu);
		if (hipMemcpyAsync(sl->d_stage, sl->h_stage, need, hipMemcpyHostToDevice, e->stream) != hipSuccess) {
			(void)hipStreamSynchronize(e->stream);
			return fail(NTC_ERR_DEVICE, "ntc_submit: host to device copy failed");
		}
		std::vector<TiledSeg> segs(n_bins);
		for (uint32_t b = 0; b < n_bins; ++b)
			segs[b] = TiledSeg{sl->d_stage + off[b], bins[b].n, bins[b].len, bins[b].ragged ? reinterpret_cast<const uint32_t*>(sl->d_stage + off[b] + tile_bytes[b]) : nullptr};
		const bool keep = e->defer_redo; // the staging pair is recycled: its K1f may not be deferred
		e->defer_redo = false;
		const int rc = run_tiled_segs(e, segs.data(), n_bins);
		e->defer_redo = keep;
		sl->used = true;
		if (hipEventRecord(sl->done, e->stream) != hipSuccess) (void)hipStreamSynchronize(e->stream);
		if (rc) return rc;
	}
	return 0;
}

// reads -> row slots (one slot per read, long sequences in overlapping chunks) -> K1
template <class LenFn, class PtrFn> int submit_rows(ntc_engine* e, uint64_t n_reads, const LenFn& len_of, const PtrFn& ptr_of)
{
	const uint32_t kmax = *std::max_element(e->klist.begin(), e->klist.end());
	const uint32_t kmin = *std::min_element(e->klist.begin(), e->klist.end());
	// ---- plan: one slot per read, or chunks with kmax-1 overlap for long sequences ----
	uint64_t maxlen = 0;
	bool uniform = true;
	const uint64_t len0 = len_of(0);
	for (uint64_t i = 0; i < n_reads; ++i) {
		const uint64_t l = len_of(i);
		maxlen = std::max(maxlen, l);
		uniform &= (l == len0);
	}
	if (maxlen < kmin) return 0; // nothing can produce a k-mer (ntHashIterator.hpp:61-64)
	const uint32_t cap_chunk = std::max<uint32_t>(kSlotCapMin, ((2 * kmax + 64) + 3) & ~3u);
	const bool chunked = maxlen > cap_chunk;
	const uint32_t stride = pick_stride(chunked ? cap_chunk : maxlen, e->klist, e->gap);
	const uint32_t ch = cap_chunk - (kmax - 1); // window starts per chunk
	uint64_t n_slots = 0;
	if (!chunked) {
		n_slots = n_reads;
	} else {
		for (uint64_t i = 0; i < n_reads; ++i) {
			const uint64_t l = len_of(i);
			if (l < kmin) continue;
			n_slots += l <= cap_chunk ? 1 : (l - (kmax - 1) + ch - 1) / ch;
		}
	}
	HIP_TRY(hipSetDevice(e->device));
	// ---- take a staging pair; wait (this thread only) until the GPU is done with its previous contents ----
	ntc_engine::StageSlot* sl = nullptr;
	{
		std::unique_lock<std::mutex> lk(e->stage_mu);
		e->stage_cv.wait(lk, [&] {
			for (auto& c : e->stage)
				if (!c.busy) return true;
			return false;
		});
		for (auto& c : e->stage)
			if (!c.busy) {
				sl = &c;
				break;
			}
		sl->busy = true;
	}
	struct Release {
		ntc_engine* e;
		ntc_engine::StageSlot* sl;
		~Release()
		{
			{
				std::lock_guard<std::mutex> lk(e->stage_mu);
				sl->busy = false;
			}
			e->stage_cv.notify_one();
		}
	} release{e, sl};
	if (sl->done == nullptr) HIP_TRY(hipEventCreateWithFlags(&sl->done, hipEventDisableTiming));
	if (sl->used) HIP_TRY(hipEventSynchronize(sl->done));
	const size_t need = (size_t)n_slots * stride + 16;
	if (need > sl->h_stage_cap) {
		if (sl->h_stage) (void)hipHostFree(sl->h_stage);
		sl->h_stage = nullptr;
		size_t cap = std::max(need, sl->h_stage_cap * 2);
		if (hipHostMalloc((void**)&sl->h_stage, cap, hipHostMallocDefault) != hipSuccess) {
			sl->h_stage_cap = 0;
			return fail(NTC_ERR_MEMORY, "ntc_submit: cannot pin %zu B", cap);
		}
		sl->h_stage_cap = cap;
	}
	if (need > sl->d_stage_cap) {
		if (sl->d_stage) (void)hipFree(sl->d_stage);
		sl->d_stage = nullptr;
		size_t cap = std::max(need, sl->d_stage_cap * 2);
		if (hipMalloc((void**)&sl->d_stage, cap) != hipSuccess) {
			sl->d_stage_cap = 0;
			return fail(NTC_ERR_MEMORY, "ntc_submit: cannot allocate %zu B on device", cap);
		}
		sl->d_stage_cap = cap;
	}
	const bool need_meta = chunked || !uniform;
	if (need_meta && n_slots > sl->h_meta_cap) {
		if (sl->h_meta) (void)hipHostFree(sl->h_meta);
		if (sl->d_meta) (void)hipFree(sl->d_meta);
		sl->h_meta = nullptr;
		sl->d_meta = nullptr;
		size_t cap = std::max<size_t>(n_slots, sl->h_meta_cap * 2);
		if (hipHostMalloc((void**)&sl->h_meta, cap * 4, hipHostMallocDefault) != hipSuccess ||
		    hipMalloc((void**)&sl->d_meta, cap * 4) != hipSuccess) {
			sl->h_meta_cap = 0;
			return fail(NTC_ERR_MEMORY, "ntc_submit: cannot allocate slot metadata");
		}
		sl->h_meta_cap = cap;
	}
	// ---- pack (the copy the ABI promises: caller's buffers are free on return) ----
	unsigned char* hs = sl->h_stage;
	uint64_t slot = 0;
	// Reads of different lengths are packed longest first (counting sort: counting is order-independent, ntcard.cpp:142-143).
	// 64 consecutive slots form a wave, and a wave whose reads are equally long takes the kernel's fast path: with 5 % of
	// trimmed reads scattered through a batch almost every wave would be ragged (0.95 vs 0.76 ms per 8 M reads).
	std::vector<uint32_t> order;
	if (!chunked && !uniform && n_reads < 0xffffffffull) {
		std::vector<uint64_t> first(maxlen + 2, 0);
		for (uint64_t i = 0; i < n_reads; ++i)
			++first[maxlen - len_of(i) + 1];
		for (uint64_t l = 1; l <= maxlen + 1; ++l)
			first[l] += first[l - 1];
		order.resize(n_reads);
		for (uint64_t i = 0; i < n_reads; ++i)
			order[first[maxlen - len_of(i)]++] = (uint32_t)i;
	}
	for (uint64_t j = 0; j < n_reads; ++j) {
		const uint64_t i = order.empty() ? j : order[j];
		const uint64_t l = len_of(i);
		const char* src = ptr_of(i);
		if (!chunked) {
			unsigned char* dst = hs + slot * stride;
			std::memcpy(dst, src, l);
			std::memset(dst + l, 'A', stride - l);
			if (need_meta) sl->h_meta[slot] = (uint32_t)l | ((uint32_t)l << 16);
			++slot;
			continue;
		}
		if (l < kmin) continue;
		if (l <= cap_chunk) {
			unsigned char* dst = hs + slot * stride;
			std::memcpy(dst, src, l);
			std::memset(dst + l, 'A', stride - l);
			sl->h_meta[slot] = (uint32_t)l | ((uint32_t)l << 16);
			++slot;
			continue;
		}
		for (uint64_t start = 0; start + (kmax - 1) < l || start == 0; start += ch) {
			const uint64_t nbytes = std::min<uint64_t>(cap_chunk, l - start);
			const bool last = start + cap_chunk >= l;
			unsigned char* dst = hs + slot * stride;
			std::memcpy(dst, src + start, nbytes);
			std::memset(dst + nbytes, 'A', stride - nbytes);
			sl->h_meta[slot] = (uint32_t)nbytes | ((uint32_t)(last ? nbytes : ch) << 16);
			++slot;
			if (last) break;
		}
	}
	if (slot != n_slots) return fail(NTC_ERR_STATE, "ntc_submit: internal slot plan mismatch (%llu != %llu)", (unsigned long long)slot, (unsigned long long)n_slots);
	// ---- enqueue: copy + kernels, in order on the engine's stream (asynchronous; ntc_sync / ntc_finish wait) ----
	{
		std::lock_guard<std::mutex> lk(e->mu);
		if (hipMemcpyAsync(sl->d_stage, hs, (size_t)n_slots * stride, hipMemcpyHostToDevice, e->stream) != hipSuccess ||
		    (need_meta && hipMemcpyAsync(sl->d_meta, sl->h_meta, n_slots * 4, hipMemcpyHostToDevice, e->stream) != hipSuccess)) {
			(void)hipStreamSynchronize(e->stream); // nothing may still read the pinned pair when it is handed out again
			return fail(NTC_ERR_DEVICE, "ntc_submit: host to device copy failed");
		}
		const int rc = run_batch(e, sl->d_stage, need_meta ? sl->d_meta : nullptr, n_slots, (uint32_t)len0, stride);
		// the copies above are in flight whatever run_batch said: the pair may only be reused after them
		sl->used = true;
		if (hipEventRecord(sl->done, e->stream) != hipSuccess) (void)hipStreamSynchronize(e->stream);
		if (rc) return rc;
	}
	return 0;
}

template <class LenFn, class PtrFn> int submit_impl(ntc_engine* e, uint64_t n_reads, const LenFn& len_of, const PtrFn& ptr_of)
{
	const uint32_t kmin = *std::min_element(e->klist.begin(), e->klist.end());
	if (e->ts_ok && n_reads >= 1024) {
		// Reads of ONE length (an untrimmed FASTQ file) are one tiled batch.  Otherwise (round 5) the reads are binned by their number of 16-base pieces,
		// C = ceil(len / 16): every bin of at least 32 Ki reads becomes a RAGGED tiled batch — sorted longest first, K1h masks the windows
		// behind every read's end — all of them hashed by ONE launch per k that shares its workgroups among the bins (run_tiled_segs), and what is left
		// (thin bins, reads shorter than every k, sequences beyond 64 Ki bases) takes row slots and K1.  profiles/r05_ragged_host.txt: 8 M reads of which
		// 5 % are trimmed to 50 .. 149 bp: 0.483 ms with bins from 32 Ki reads, 0.527 from 512 Ki (the thin bins through K1), 0.728 through K1 alone;
		// lengths uniform in 100 .. 150: 0.385 against 0.647.  (With one launch PER BIN, the first form of this, thin bins lost to K1 and the bar was 512 Ki.)
		// NTC_FLAG_REQUIRE_TILED, the validation flag, lowers the bar to 1024 reads.
		uint32_t bin_min = e->ts_required ? 1024u : 32u * 1024u;
		if (const char* ev = std::getenv("NTC_BIN_MIN")) bin_min = (uint32_t)std::max(1024l, std::strtol(ev, nullptr, 10)); // tuning runs (tools/ragged_time.py)
		const uint64_t len0 = len_of(0);
		uint64_t same = 0;
		for (uint64_t i = 0; i < n_reads; ++i)
			same += len_of(i) == len0;
		if (same == n_reads) {
			if (len0 >= kmin && len0 <= 0xffffu && k1_fits_tiles(e, (uint32_t)len0)) { // (a mixed list with reads too long for K1's tiled staging: row slots, chunked)
				const HostBin one{nullptr, n_reads, (uint32_t)len0, false};
				return submit_tiled_host(e, &one, 1, len_of, ptr_of);
			}
		} else { // (round 6: also under a list of which a part is K1's — K1 gets a slot table derived from the tiles' prefix tables)
			constexpr uint32_t kMaxC = 0x10000u / 16u;
			std::vector<uint32_t> per_c(kMaxC + 1, 0u);
			for (uint64_t i = 0; i < n_reads; ++i) {
				const uint64_t l = len_of(i);
				if (l >= kmin && l <= 0xffffu) ++per_c[(l + 15u) / 16u];
			}
			std::vector<uint64_t> rest;
			std::vector<std::vector<uint64_t>> bins; // (only the bins that are taken)
			std::vector<int32_t> bin_of(kMaxC + 1, -1);
			for (uint32_t c = 1; c <= kMaxC; ++c)
				if (per_c[c] >= bin_min && k1_fits_tiles(e, 16u * c)) {
					bin_of[c] = (int32_t)bins.size();
					bins.emplace_back();
					bins.back().reserve(per_c[c]);
				}
			if (!bins.empty()) {
				for (uint64_t i = 0; i < n_reads; ++i) {
					const uint64_t l = len_of(i);
					const int32_t b = (l >= kmin && l <= 0xffffu) ? bin_of[(l + 15u) / 16u] : -1;
					if (b >= 0) bins[(size_t)b].push_back(i);
					else rest.push_back(i);
				}
				std::vector<HostBin> hbins;
				for (uint32_t c = kMaxC; c >= 1; --c) { // (longest bin first)
					if (bin_of[c] < 0) continue;
					std::vector<uint64_t>& idx = bins[(size_t)bin_of[c]];
					// longest first: a counting sort by the 16 possible tails (stable)
					std::vector<uint64_t> sorted(idx.size());
					size_t start[17] = { 0 };
					for (uint64_t i : idx)
						++start[16u - (uint32_t)(len_of(i) - 16u * (c - 1u))]; // tail 16 -> bucket 0
					size_t acc = 0;
					for (int t = 0; t < 17; ++t) {
						const size_t n = start[t];
						start[t] = acc;
						acc += n;
					}
					for (uint64_t i : idx)
						sorted[start[16u - (uint32_t)(len_of(i) - 16u * (c - 1u))]++] = i;
					idx.swap(sorted);
					hbins.push_back(HostBin{idx.data(), idx.size(), 16u * c, true});
				}
				// all the bins in one staging buffer and one launch per k (up to 8 bins: run_tiled_segs groups the rest)
				if (int rc = submit_tiled_host(e, hbins.data(), (uint32_t)hbins.size(), len_of, ptr_of)) return rc;
				if (rest.empty()) return 0;
				return submit_rows(e, rest.size(), [&](uint64_t i) { return len_of(rest[i]); }, [&](uint64_t i) { return ptr_of(rest[i]); });
			}
		}
	}
	return submit_rows(e, n_reads, len_of, ptr_of);
}

} // namespace

extern "C" {

int ntc_submit(ntc_engine* e, const char* bases, const uint64_t* offsets, uint64_t n_reads)
{
	if (!e) return fail(NTC_ERR_ARG, "ntc_submit: null engine");
	if (n_reads == 0) return 0;
	if (!bases || !offsets) return fail(NTC_ERR_ARG, "ntc_submit: null buffer");
	for (uint64_t i = 0; i < n_reads; ++i)
		if (offsets[i + 1] < offsets[i]) return fail(NTC_ERR_ARG, "ntc_submit: offsets not monotone at read %llu", (unsigned long long)i);
	return submit_impl(e, n_reads, [&](uint64_t i) { return offsets[i + 1] - offsets[i]; }, [&](uint64_t i) { return bases + offsets[i]; });
}

int ntc_submit_spans(ntc_engine* e, const char* buf, const uint64_t* starts, const uint32_t* lens, uint64_t n_reads)
{
	if (!e) return fail(NTC_ERR_ARG, "ntc_submit_spans: null engine");
	if (n_reads == 0) return 0;
	if (!buf || !starts || !lens) return fail(NTC_ERR_ARG, "ntc_submit_spans: null buffer");
	return submit_impl(e, n_reads, [&](uint64_t i) { return (uint64_t)lens[i]; }, [&](uint64_t i) { return buf + starts[i]; });
}

int ntc_sync(ntc_engine* e)
{
	if (!e) return fail(NTC_ERR_ARG, "ntc_sync: null engine");
	std::lock_guard<std::mutex> lk(e->mu);
	HIP_TRY(hipSetDevice(e->device));
	if (int rc = join_k1f(e)) return rc;   // (K1f reads the batches too)
	HIP_TRY(hipStreamSynchronize(e->stream));
	return drain_events(e);
}

int ntc_finish(ntc_engine* e, uint16_t* t_counter_out, uint32_t* p_hist_out, uint64_t* f1_out)
{
	if (!e) return fail(NTC_ERR_ARG, "ntc_finish: null engine");
	if (e->hll_bits) return fail(NTC_ERR_STATE, "ntc_finish: this is an nthll engine, use ntc_hll_finish");
	std::lock_guard<std::mutex> lk(e->mu);
	HIP_TRY(hipSetDevice(e->device));
	const size_t nk = e->klist.size();
	const uint64_t per_sample = 1ull << e->r_bits;
	if (int rc = apply_log(e)) return rc; // pending increments first: compEst reads the counters (ntcard.cpp:240-247)
	if (t_counter_out && !e->d_out16) {
		if (hipMalloc((void**)&e->d_out16, 2 * per_sample * sizeof(uint16_t)) != hipSuccess)
			return fail(NTC_ERR_MEMORY, "ntc_finish: cannot allocate uint16 staging");
	}
	if (p_hist_out || t_counter_out) {
		HIP_TRY(hipMemsetAsync(e->d_phist, 0, nk * 2 * 65536 * 4, e->stream));
		for (size_t ki = 0; ki < nk; ++ki) {
			HIP_TRY(ntc::launch_finalize(e->d_sketch + ki * e->plane_elems(), per_sample,
			                             e->d_phist + ki * 2 * 65536, t_counter_out ? e->d_out16 : nullptr, e->stream));
			if (t_counter_out)
				HIP_TRY(hipMemcpyAsync(t_counter_out + ki * 2 * per_sample, e->d_out16, 2 * per_sample * sizeof(uint16_t),
				                       hipMemcpyDeviceToHost, e->stream));
		}
		if (p_hist_out)
			HIP_TRY(hipMemcpyAsync(p_hist_out, e->d_phist, nk * 2 * 65536 * 4, hipMemcpyDeviceToHost, e->stream));
	}
	if (f1_out) HIP_TRY(hipMemcpyAsync(f1_out, e->d_f1, nk * 8, hipMemcpyDeviceToHost, e->stream));
	HIP_TRY(hipStreamSynchronize(e->stream));
	return drain_events(e);
}

int ntc_value_hist_device(int32_t device, void* stream, const void* d_counters_u32, uint64_t n, void* d_hist_u32)
{
	if (!d_counters_u32 || !d_hist_u32) return fail(NTC_ERR_ARG, "ntc_value_hist_device: null buffer");
	if ((n & 3u) || ((uintptr_t)d_counters_u32 & 15u)) return fail(NTC_ERR_ARG, "ntc_value_hist_device: need n %% 4 == 0 and 16-byte aligned counters");
	if (n == 0) return 0;
	HIP_TRY(hipSetDevice(device));
	HIP_TRY(ntc::launch_value_hist((const uint32_t*)d_counters_u32, n, (uint32_t*)d_hist_u32, (hipStream_t)stream));
	return 0;
}

// The three device steps of the multi-GPU merge for a caller that moves the slices itself (one process per GPU: RCCL all-to-all under
// torch.distributed, ntcard_amd/parallel.py) — the kernels ntc_merge_devices runs between its peer copies
int ntc_narrow_u16_device(int32_t device, void* stream, const void* d_counters_u32, uint64_t n, void* d_out_u16)
{
	if (!d_counters_u32 || !d_out_u16) return fail(NTC_ERR_ARG, "ntc_narrow_u16_device: null buffer");
	if (((uintptr_t)d_counters_u32 & 15u) || ((uintptr_t)d_out_u16 & 15u)) return fail(NTC_ERR_ARG, "ntc_narrow_u16_device: need 16-byte aligned buffers");
	if (n == 0) return 0;
	HIP_TRY(hipSetDevice(device));
	HIP_TRY(ntc::launch_narrow_u16((const uint32_t*)d_counters_u32, (uint16_t*)d_out_u16, n, (hipStream_t)stream));
	return 0;
}

int ntc_sum_slices_u16_device(int32_t device, void* stream, void* d_slices_u16, uint64_t stride, uint32_t n_slices, uint64_t len)
{
	if (!d_slices_u16) return fail(NTC_ERR_ARG, "ntc_sum_slices_u16_device: null buffer");
	if (((uintptr_t)d_slices_u16 & 15u) || (stride & 7u) || len > stride) return fail(NTC_ERR_ARG, "ntc_sum_slices_u16_device: need 16-byte aligned slices, stride %% 8 == 0, len <= stride");
	if (n_slices <= 1 || len == 0) return 0;
	HIP_TRY(hipSetDevice(device));
	HIP_TRY(ntc::launch_sum_slices_u16((uint16_t*)d_slices_u16, stride, n_slices, len, (hipStream_t)stream));
	return 0;
}

int ntc_value_hist_u16_device(int32_t device, void* stream, const void* d_counters_u16, uint64_t n, void* d_hist_u32)
{
	if (!d_counters_u16 || !d_hist_u32) return fail(NTC_ERR_ARG, "ntc_value_hist_u16_device: null buffer");
	if ((uintptr_t)d_counters_u16 & 15u) return fail(NTC_ERR_ARG, "ntc_value_hist_u16_device: need 16-byte aligned counters");
	if (n == 0) return 0;
	HIP_TRY(hipSetDevice(device));
	HIP_TRY(ntc::launch_value_hist_u16((const uint16_t*)d_counters_u16, n, (uint32_t*)d_hist_u32, (hipStream_t)stream));
	return 0;
}

int ntc_merge_counters(ntc_engine* e, const uint16_t* t_counter, const uint64_t* f1)
{
	if (!e || !t_counter) return fail(NTC_ERR_ARG, "ntc_merge_counters: null argument");
	if (e->hll_bits) return fail(NTC_ERR_STATE, "ntc_merge_counters: not for an nthll engine");
	std::lock_guard<std::mutex> lk(e->mu);
	HIP_TRY(hipSetDevice(e->device));
	const size_t nk = e->klist.size();
	const uint64_t per_k = e->plane_elems(); // counters per k (both samples)
	if (!e->d_out16 && hipMalloc((void**)&e->d_out16, per_k * sizeof(uint16_t)) != hipSuccess)
		return fail(NTC_ERR_MEMORY, "ntc_merge_counters: cannot allocate uint16 staging");
	for (size_t ki = 0; ki < nk; ++ki) {
		HIP_TRY(hipMemcpyAsync(e->d_out16, t_counter + ki * per_k, per_k * sizeof(uint16_t), hipMemcpyHostToDevice, e->stream));
		HIP_TRY(ntc::launch_add_counters(e->d_sketch + ki * per_k, e->d_out16, per_k, e->stream));
		e->sk_host_dirty = true;
	}
	if (f1) {
		std::vector<unsigned long long> cur(nk);
		HIP_TRY(hipMemcpyAsync(cur.data(), e->d_f1, nk * 8, hipMemcpyDeviceToHost, e->stream));
		HIP_TRY(hipStreamSynchronize(e->stream));
		for (size_t ki = 0; ki < nk; ++ki)
			cur[ki] += f1[ki];
		HIP_TRY(hipMemcpyAsync(e->d_f1, cur.data(), nk * 8, hipMemcpyHostToDevice, e->stream));
	}
	HIP_TRY(hipStreamSynchronize(e->stream));
	return 0;
}

int ntc_flush(ntc_engine* e)
{
	if (!e) return fail(NTC_ERR_ARG, "ntc_flush: null engine");
	std::lock_guard<std::mutex> lk(e->mu);
	HIP_TRY(hipSetDevice(e->device));
	return apply_log(e);
}

// ---- multi-GPU merge in ONE host process -------------------------------------------------------
// The reference's threads all increment one shared t_Counter (ntcard.cpp:142-143,445) and add their k-mer counts into
// one totalKmers (ntcard.cpp:464-466); with one private sketch per engine the same state is the element-wise SUM of the
// sketches (MAX for nthll's registers, nthll.cpp:238-243).  t_Counter wraps at 16 bits, so only the low halves of the
// per-engine counters matter: (sum_e c_e) mod 2^16 == (sum_e (c_e mod 2^16)) mod 2^16.  The merge is the same exchange
// bench.py runs between processes with RCCL's all-to-all (ntcard_amd/parallel.py), written with peer copies because here
// all devices belong to one process: every engine narrows its counters to 16 bits, slice j of every engine goes to engine
// j's device — all N x (N-1) copies are in flight together, each on its own point-to-point xGMI link, 2 B x counters / N
// per link —, engine j adds its N slices with wrapping 16-bit adds, the summed slices are gathered on engine 0's device
// (again one slice per link) and widened into engine 0's sketch.  No communicator, no library beyond HIP; devices without
// peer access are served by hipMemcpyPeerAsync's staged path.  nthll's register file (2^nBits dwords) and F1 are tiny:
// copied to the root device and folded there (max / sum, full width).
namespace {
struct MergePeer {
	ntc_engine* e = nullptr;
	uint16_t* narrow = nullptr; // [counters]      this engine's counters mod 2^16
	uint16_t* recv = nullptr;   // [n][slice]      slice `me` of every engine; the sum ends up in recv[0 .. slice)
	std::vector<hipStream_t> lanes; // one copy stream per peer: the copies into this device run side by side
	hipEvent_t narrowed = nullptr, summed = nullptr;
	std::vector<hipEvent_t> arrived;
};
hipError_t copy_between(void* dst, int dst_dev, const void* src, int src_dev, size_t bytes, hipStream_t st)
{
	if (bytes == 0) return hipSuccess;
	return dst_dev == src_dev ? hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st) : hipMemcpyPeerAsync(dst, dst_dev, src, src_dev, bytes, st);
}
void enable_peer_access(const std::vector<MergePeer>& peers)
{
	for (const MergePeer& a : peers)
		for (const MergePeer& b : peers) {
			int can = 0;
			if (a.e->device == b.e->device || hipDeviceCanAccessPeer(&can, a.e->device, b.e->device) != hipSuccess || !can) continue;
			(void)hipSetDevice(a.e->device);
			(void)hipDeviceEnablePeerAccess(b.e->device, 0); // hipErrorPeerAccessAlreadyEnabled is fine
			(void)hipGetLastError();
		}
}
void release_peers(std::vector<MergePeer>& peers) // (buffers, streams and events belong to the engines' merge caches: only wait)
{
	for (MergePeer& p : peers) {
		if (!p.e) continue;
		(void)hipSetDevice(p.e->device);
		for (hipStream_t s : p.lanes)
			if (s) (void)hipStreamSynchronize(s);
		(void)hipStreamSynchronize(p.e->stream);
	}
}
// full-width fold of a small array of every engine into the root's (nthll registers: max; F1: sum)
int fold_small(ntc_engine* const* engines, int32_t n, bool regs)
{
	ntc_engine* root = engines[0];
	const size_t bytes = regs ? (size_t)4 << root->hll_bits : root->klist.size() * 8;
	void* tmp = nullptr;
	HIP_TRY(hipSetDevice(root->device));
	HIP_TRY(hipMalloc(&tmp, bytes));
	int rc = 0;
	for (int32_t i = 1; i < n && !rc; ++i) {
		const void* src = regs ? (const void*)engines[i]->d_sketch : (const void*)engines[i]->d_f1;
		hipError_t h = copy_between(tmp, root->device, src, engines[i]->device, bytes, root->stream);
		if (h == hipSuccess)
			h = regs ? ntc::launch_fold_u32(root->d_sketch, (const uint32_t*)tmp, bytes / 4, true, root->stream)
			         : ntc::launch_fold_u64((unsigned long long*)root->d_f1, (const unsigned long long*)tmp, bytes / 8, root->stream);
		if (h != hipSuccess) rc = fail(NTC_ERR_DEVICE, "ntc_merge_devices: folding engine %d failed: %s", i, hipGetErrorString(h));
	}
	(void)hipStreamSynchronize(root->stream);
	(void)hipFree(tmp);
	return rc;
}
} // namespace

int ntc_merge_devices(ntc_engine* const* engines, int32_t n_engines)
{
	if (!engines || n_engines < 1) return fail(NTC_ERR_ARG, "ntc_merge_devices: need at least one engine");
	ntc_engine* root = engines[0];
	if (!root) return fail(NTC_ERR_ARG, "ntc_merge_devices: null engine");
	for (int32_t i = 0; i < n_engines; ++i) {
		ntc_engine* e = engines[i];
		if (!e || e->klist != root->klist || e->gap != root->gap || e->r_bits != root->r_bits || e->s_bits != root->s_bits || e->hll_bits != root->hll_bits)
			return fail(NTC_ERR_ARG, "ntc_merge_devices: engine %d is not configured like engine 0", i);
		for (int32_t j = 0; j < i; ++j)
			if (engines[j] == e) return fail(NTC_ERR_ARG, "ntc_merge_devices: engine %d listed twice", i);
	}
	// every engine stays locked (in address order) from the applies to the last copy: a submit on one of them from another thread would
	// race with the narrow / widen kernels on its sketch
	std::vector<ntc_engine*> order(engines, engines + n_engines);
	std::sort(order.begin(), order.end());
	std::vector<std::unique_lock<std::mutex>> locks;
	for (ntc_engine* e : order)
		locks.emplace_back(e->mu);
	// 1. pending increments first, everything quiescent
	for (int32_t i = 0; i < n_engines; ++i) {
		HIP_TRY(hipSetDevice(engines[i]->device));
		if (int rc = apply_log(engines[i])) return rc;
		HIP_TRY(hipStreamSynchronize(engines[i]->stream));
	}
	if (n_engines == 1) return 0;
	const uint32_t n = (uint32_t)n_engines;
	// 2. F1 (and nthll's registers) at full width
	if (int rc = fold_small(engines, n_engines, false)) return rc;
	if (root->hll_bits) {
		if (int rc = fold_small(engines, n_engines, true)) return rc;
	} else {
		// 3. the counters: 16-bit slices, all-to-all, wrapping sums, gather, widen
		const uint64_t counters = root->klist.size() * root->plane_elems();
		const uint64_t slice = ((counters + n - 1) / n + 7) & ~7ull; // elements per slice (16-byte multiples); the last one may be short or empty
		auto len_of = [&](uint32_t j) { return (uint64_t)j * slice >= counters ? 0ull : std::min<uint64_t>(slice, counters - (uint64_t)j * slice); };
		std::vector<MergePeer> peers(n);
		auto run = [&]() -> int {
			for (uint32_t i = 0; i < n; ++i) {
				MergePeer& p = peers[i];
				p.e = engines[i];
				HIP_TRY(hipSetDevice(p.e->device));
				auto& mc = p.e->mc;
				if (mc.narrow_cap < counters) {
					if (mc.narrow) (void)hipFree(mc.narrow);
					mc.narrow = nullptr;
					mc.narrow_cap = 0;
					if (hipMalloc((void**)&mc.narrow, counters * 2) != hipSuccess)
						return fail(NTC_ERR_MEMORY, "ntc_merge_devices: cannot allocate the %llu-byte exchange buffer on device %d", (unsigned long long)(counters * 2), p.e->device);
					mc.narrow_cap = counters;
					++p.e->merge_allocs;
				}
				if (mc.recv_cap < (size_t)n * slice) {
					if (mc.recv) (void)hipFree(mc.recv);
					mc.recv = nullptr;
					mc.recv_cap = 0;
					if (hipMalloc((void**)&mc.recv, (size_t)n * slice * 2) != hipSuccess)
						return fail(NTC_ERR_MEMORY, "ntc_merge_devices: cannot allocate the %llu-byte exchange buffer on device %d", (unsigned long long)((size_t)n * slice * 2), p.e->device);
					mc.recv_cap = (size_t)n * slice;
					++p.e->merge_allocs;
				}
				while (mc.lanes.size() < n) {
					hipStream_t st = nullptr;
					hipEvent_t ev = nullptr;
					HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
					if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { // lanes and arrived stay the same length
						(void)hipStreamDestroy(st);
						return fail(NTC_ERR_DEVICE, "ntc_merge_devices: cannot create an event on device %d", p.e->device);
					}
					mc.lanes.push_back(st);
					mc.arrived.push_back(ev);
					p.e->merge_allocs += 2;
				}
				if (!mc.narrowed) {
					HIP_TRY(hipEventCreateWithFlags(&mc.narrowed, hipEventDisableTiming));
					++p.e->merge_allocs;
				}
				if (!mc.summed) { // (tested on its own: a failure here must be retried by the next merge)
					HIP_TRY(hipEventCreateWithFlags(&mc.summed, hipEventDisableTiming));
					++p.e->merge_allocs;
				}
				p.narrow = mc.narrow;
				p.recv = mc.recv;
				p.lanes.assign(mc.lanes.begin(), mc.lanes.begin() + n);
				p.arrived.assign(mc.arrived.begin(), mc.arrived.begin() + n);
				p.narrowed = mc.narrowed;
				p.summed = mc.summed;
			}
			enable_peer_access(peers);
			for (MergePeer& p : peers) { // narrow
				HIP_TRY(hipSetDevice(p.e->device));
				HIP_TRY(ntc::launch_narrow_u16(p.e->d_sketch, p.narrow, counters, p.e->stream));
				HIP_TRY(hipEventRecord(p.narrowed, p.e->stream));
			}
			for (uint32_t j = 0; j < n; ++j) { // all-to-all: slice j of engine i -> engine j, on j's lane i
				MergePeer& dst = peers[j];
				HIP_TRY(hipSetDevice(dst.e->device));
				for (uint32_t t = 0; t < n; ++t) {
					const uint32_t i = (j + t) % n; // staggered start: at every moment the devices talk to distinct partners
					const MergePeer& src = peers[i];
					HIP_TRY(hipStreamWaitEvent(dst.lanes[i], src.narrowed, 0));
					HIP_TRY(copy_between(dst.recv + (uint64_t)i * slice, dst.e->device, src.narrow + (uint64_t)j * slice, src.e->device, len_of(j) * 2, dst.lanes[i]));
					HIP_TRY(hipEventRecord(dst.arrived[i], dst.lanes[i]));
					HIP_TRY(hipStreamWaitEvent(dst.e->stream, dst.arrived[i], 0));
				}
				HIP_TRY(ntc::launch_sum_slices_u16(dst.recv, slice, n, len_of(j), dst.e->stream));
				HIP_TRY(hipEventRecord(dst.summed, dst.e->stream));
			}
			// gather on the root: its own `narrow` is free once every peer has taken its slice of it — simpler: the root's lanes wait for
			// those copies (arrived events of slice 0 on every peer) before overwriting
			MergePeer& r0 = peers[0];
			HIP_TRY(hipSetDevice(r0.e->device));
			for (uint32_t j = 0; j < n; ++j) {
				for (uint32_t q = 0; q < n; ++q)
					HIP_TRY(hipStreamWaitEvent(r0.lanes[j], peers[q].arrived[0], 0)); // engine 0's slices have left `narrow`
				HIP_TRY(hipStreamWaitEvent(r0.lanes[j], peers[j].summed, 0));
				HIP_TRY(copy_between(r0.narrow + (uint64_t)j * slice, r0.e->device, peers[j].recv, peers[j].e->device, len_of(j) * 2, r0.lanes[j]));
				HIP_TRY(hipEventRecord(r0.arrived[j], r0.lanes[j])); // (re-used: slice j of the sum has arrived)
				HIP_TRY(hipStreamWaitEvent(r0.e->stream, r0.arrived[j], 0));
			}
			HIP_TRY(ntc::launch_widen_u16(r0.narrow, r0.e->d_sketch, counters, r0.e->stream));
			r0.e->sk_host_dirty = true;
			for (MergePeer& p : peers) {
				HIP_TRY(hipSetDevice(p.e->device));
				for (hipStream_t s : p.lanes)
					HIP_TRY(hipStreamSynchronize(s));
				HIP_TRY(hipStreamSynchronize(p.e->stream));
			}
			return 0;
		};
		const int rc = run();
		release_peers(peers);
		if (rc) return rc;
	}
	// 4. everything now lives in engine 0 (counters as their value mod 2^16, which is all t_Counter ever held): the others start
	//    from zero again, the sum stays what it was
	locks.clear(); // (ntc_reset takes the engine's lock itself)
	for (int32_t i = 1; i < n_engines; ++i)
		if (int rc = ntc_reset(engines[i])) return rc;
	HIP_TRY(hipSetDevice(root->device));
	return 0;
}

int ntc_log_export_device(ntc_engine* e, uint32_t n_parts, void* d_keys_u32, const uint64_t* part_offset, uint64_t* counts_out)
{
	if (!e || !counts_out) return fail(NTC_ERR_ARG, "ntc_log_export_device: null argument");
	if (n_parts < 1 || n_parts > 64) return fail(NTC_ERR_ARG, "ntc_log_export_device: n_parts %u outside 1..64", n_parts);
	if (d_keys_u32 && !part_offset) return fail(NTC_ERR_ARG, "ntc_log_export_device: keys without part offsets");
	std::lock_guard<std::mutex> lk(e->mu);
	HIP_TRY(hipSetDevice(e->device));
	if (!e->d_log || e->hll_bits) return fail(NTC_ERR_STATE, "ntc_log_export_device: this engine has no hit log");
	const uint64_t counters = e->klist.size() * e->plane_elems();
	if (counters % n_parts) return fail(NTC_ERR_ARG, "ntc_log_export_device: %u parts do not divide %llu counters", n_parts, (unsigned long long)counters);
	if (int rc = join_k1f(e)) return rc; // (K1f's suspects are log entries too)
	// the sketch must still hold the zeros of the last reset: everything counted so far is in the log
	uint32_t dirty = 0;
	HIP_TRY(hipMemcpyAsync(&dirty, e->d_skdirty, 4, hipMemcpyDeviceToHost, e->stream));
	HIP_TRY(hipStreamSynchronize(e->stream));
	if (e->sk_host_dirty || dirty != 0u)
		return fail(NTC_ERR_STATE, "ntc_log_export_device: the sketch already holds counts (a sketch update ran, or a kernel incremented it directly): merge counters instead");
	unsigned long long* const d_cursor = reinterpret_cast<unsigned long long*>(e->d_skdirty + 16);
	unsigned long long* const d_off = d_cursor + 64;
	unsigned long long h_off[64] = {0};
	if (part_offset)
		for (uint32_t p = 0; p < n_parts; ++p)
			h_off[p] = part_offset[p];
	hipError_t rc = hipMemsetAsync(d_cursor, 0, 64 * 8, e->stream);
	if (rc == hipSuccess) rc = hipMemcpyAsync(d_off, h_off, 64 * 8, hipMemcpyHostToDevice, e->stream);
	if (rc == hipSuccess)
		rc = ntc::launch_log_export(e->d_log, e->d_logfill, e->log_region_cap, e->all_log_regions(), n_parts, (uint32_t)(counters / n_parts), (uint32_t*)d_keys_u32, d_off,
		                            d_cursor, e->stream);
	unsigned long long h_cnt[64] = {0};
	if (rc == hipSuccess) rc = hipMemcpyAsync(h_cnt, d_cursor, 64 * 8, hipMemcpyDeviceToHost, e->stream);
	if (rc == hipSuccess) rc = hipStreamSynchronize(e->stream);
	if (rc != hipSuccess) return fail(NTC_ERR_DEVICE, "ntc_log_export_device: %s", hipGetErrorString(rc));
	for (uint32_t p = 0; p < n_parts; ++p)
		counts_out[p] = h_cnt[p];
	return 0;
}

int ntc_log_replace_device(ntc_engine* e, const void* d_keys_u32, uint64_t n_keys)
{
	if (!e || (!d_keys_u32 && n_keys)) return fail(NTC_ERR_ARG, "ntc_log_replace_device: null argument");
	std::lock_guard<std::mutex> lk(e->mu);
	HIP_TRY(hipSetDevice(e->device));
	if (!e->d_log || e->hll_bits) return fail(NTC_ERR_STATE, "ntc_log_replace_device: this engine has no hit log");
	const uint64_t room = (uint64_t)e->all_log_regions() * e->log_region_cap;
	if (n_keys > room) return fail(NTC_ERR_ARG, "ntc_log_replace_device: %llu keys do not fit the %llu-entry log", (unsigned long long)n_keys, (unsigned long long)room);
	if (int rc = join_k1f(e)) return rc; // (nothing may append behind this point)
	if (n_keys) HIP_TRY(hipMemcpyAsync(e->d_log, d_keys_u32, n_keys * 4, hipMemcpyDeviceToDevice, e->stream));
	HIP_TRY(ntc::launch_log_set_fill(e->d_logfill, e->all_log_regions(), e->log_region_cap, n_keys, e->stream));
	e->log_pending = n_keys != 0;
	e->log_est = (double)n_keys;
	return 0;
}

int ntc_device_state(ntc_engine* e, void** d_sketch_u32, uint64_t* n_counters, void** d_f1_u64)
{
	if (!e) return fail(NTC_ERR_ARG, "ntc_device_state: null engine");
	{
		std::lock_guard<std::mutex> lk(e->mu);
		HIP_TRY(hipSetDevice(e->device));
		if (int rc = apply_log(e)) return rc; // the caller is about to read or reduce the counters
	}
	if (d_sketch_u32) {
		*d_sketch_u32 = e->d_sketch;
		e->sk_host_dirty = e->sk_exposed = true; // (the caller may add to the counters: nothing is known about them any more)
	}
	if (n_counters) *n_counters = e->klist.size() * e->plane_elems();
	if (d_f1_u64) *d_f1_u64 = e->d_f1;
	return 0;
}

int ntc_hash_dump_device(int32_t device, void* stream, const void* d_slots, uint64_t n_reads, uint32_t read_len,
                         uint32_t stride, uint32_t k, uint32_t gap, uint32_t max_win, void* d_hash_out,
                         void* d_count_out)
{
	if (!d_slots || !d_hash_out || !d_count_out) return fail(NTC_ERR_ARG, "ntc_hash_dump_device: null buffer");
	if (k < 1 || k > kMaxK) return fail(NTC_ERR_ARG, "ntc_hash_dump_device: k=%u outside 1..%u", k, kMaxK);
	if (gap != 0) // the simple kernel has no spaced-seed form: the production kernel's validation build does it
		return ntc_hash_dump_k1_device(device, stream, d_slots, n_reads, read_len, stride, k, gap, max_win, d_hash_out, d_count_out);
	if ((stride & 3u) || stride < read_len || ((uintptr_t)d_slots & 15u))
		return fail(NTC_ERR_ARG, "ntc_hash_dump_device: need 16-byte aligned slots, stride %% 4 == 0, stride >= read_len");
	if (n_reads == 0) return 0;
	HIP_TRY(hipSetDevice(device));
	unsigned grid = 0;
	size_t smem = 0;
	if (int rc = ensure_kernel_attrs(device)) return rc;
	if (int rc = hash_grid(device, n_reads, stride, grid, smem)) return rc;
	ntc::HashArgs a;
	std::memset(&a, 0, sizeof a);
	a.slots = (const unsigned char*)d_slots;
	a.n_slots = n_reads;
	a.stride = stride;
	a.read_len = read_len;
	a.k = k;
	a.r_bits = 27;
	a.s_bits = 7;
	a.max_win = max_win;
	a.dump = (uint64_t*)d_hash_out;
	a.dump_count = (uint32_t*)d_count_out;
	ntc::build_tables(k, a.tab);
	HIP_TRY(ntc::launch_hash(1, a, grid, smem, (hipStream_t)stream));
	return 0;
}

int ntc_hash_dump_k1_device(int32_t device, void* stream, const void* d_slots, uint64_t n_reads, uint32_t read_len,
                            uint32_t stride, uint32_t k, uint32_t gap, uint32_t max_win, void* d_hash_out, void* d_count_out)
{
	if (!d_slots || !d_hash_out || !d_count_out) return fail(NTC_ERR_ARG, "ntc_hash_dump_k1_device: null buffer");
	if (k < 1 || k > kMaxK) return fail(NTC_ERR_ARG, "ntc_hash_dump_k1_device: k=%u outside 1..%u", k, kMaxK);
	if (gap != 0 && (gap % 2 != k % 2 || gap >= k)) return fail(NTC_ERR_ARG, "ntc_hash_dump_k1_device: gap size and kmer must have the same modulus");
	if ((stride & 3u) || stride < read_len || ((uintptr_t)d_slots & 15u))
		return fail(NTC_ERR_ARG, "ntc_hash_dump_k1_device: need 16-byte aligned slots, stride %% 4 == 0, stride >= read_len");
	if (n_reads == 0) return 0;
	HIP_TRY(hipSetDevice(device));
	if (int rc = ensure_kernel_attrs(device)) return rc;
	hipStream_t st = (hipStream_t)stream;
	const uint32_t n_win = read_len >= k ? read_len - k + 1 : 1;
	const uint32_t gap_first = (k - gap) / 2;
	std::vector<uint32_t> t1((size_t)ntc::t2_pairs(k) * 64), gt((size_t)((gap + 1) / 2) * 64);
	ntc::build_t2(k, t1.data(), gap_first, gap);
	if (gap) ntc::build_gap_table(k, gap_first, gap, gt.data());
	void *d_t1 = nullptr, *d_gt = nullptr;
	uint64_t* d_full = nullptr;
	uint32_t* d_valid = nullptr;
	unsigned long long* d_f1 = nullptr;
	const size_t vbytes = (size_t)n_reads * ((n_win + 31) / 32) * 4;
	auto cleanup = [&]() {
		for (void* p : {d_t1, d_gt, (void*)d_full, (void*)d_valid, (void*)d_f1})
			if (p) (void)hipFree(p);
	};
	if (hipMalloc(&d_t1, t1.size() * 4) != hipSuccess || (gap && hipMalloc(&d_gt, gt.size() * 4) != hipSuccess) ||
	    hipMalloc((void**)&d_full, (size_t)n_reads * n_win * 8) != hipSuccess || hipMalloc((void**)&d_valid, vbytes) != hipSuccess ||
	    hipMalloc((void**)&d_f1, 8) != hipSuccess) {
		cleanup();
		return fail(NTC_ERR_MEMORY, "ntc_hash_dump_k1_device: device allocation failed");
	}
	int rc = 0;
	auto run = [&]() -> int {
		HIP_TRY(hipMemcpyAsync(d_t1, t1.data(), t1.size() * 4, hipMemcpyHostToDevice, st));
		if (gap) HIP_TRY(hipMemcpyAsync(d_gt, gt.data(), gt.size() * 4, hipMemcpyHostToDevice, st));
		HIP_TRY(hipMemsetAsync(d_valid, 0, vbytes, st));
		HIP_TRY(hipMemsetAsync(d_f1, 0, 8, st));
		ntc::HfArgs a;
		std::memset(&a, 0, sizeof a);
		a.slots = (const unsigned char*)d_slots;
		a.n_slots = n_reads;
		a.stride = stride;
		a.read_len = read_len;
		a.r_bits = 27;
		a.s_bits = 7;
		a.n_k = 1;
		a.gap = gap;
		a.gap_first = gap_first;
		a.gapt = d_gt;
		if (gap) ntc::build_gap_roll_table(k, gap_first, gap, a.tabg);
		fill_hfk(a.ks[0], k, nullptr, d_f1, d_t1);
		a.dump = d_full;
		a.dump_valid = d_valid;
		a.dump_win = n_win;
		HfPlan hp;
		if (int r = hf_plan(device, n_reads, stride, &k, 1, gap, hp)) return r;
		HIP_TRY(ntc::launch_sketch_hf(a, hp.grid, hp.wpb, hp.smem, st));
		HIP_TRY(ntc::launch_compact_dump(d_full, d_valid, n_reads, n_win, max_win, (uint64_t*)d_hash_out, (uint32_t*)d_count_out, st));
		HIP_TRY(hipStreamSynchronize(st));
		return 0;
	};
	rc = run();
	cleanup();
	return rc;
}

int ntc_gen_reads_device(int32_t device, void* stream, void* d_slots, uint64_t seed, uint64_t first_read,
                         uint64_t n_reads, uint32_t read_len, uint32_t stride, uint32_t dist, uint64_t genome_len)
{
	if (!d_slots || (stride & 3u) || stride < read_len) return fail(NTC_ERR_ARG, "ntc_gen_reads_device: bad layout");
	if (dist > 1) return fail(NTC_ERR_ARG, "ntc_gen_reads_device: dist must be 0 (uniform) or 1 (genome)");
	if (dist == 1 && genome_len < read_len) return fail(NTC_ERR_ARG, "ntc_gen_reads_device: genome shorter than a read");
	if (n_reads == 0) return 0;
	HIP_TRY(hipSetDevice(device));
	HIP_TRY(ntc::launch_gen((unsigned char*)d_slots, seed, first_read, n_reads, read_len, stride, dist, genome_len,
	                        (hipStream_t)stream));
	return 0;
}

int ntc_hll_create(uint32_t k, uint32_t n_bits, int32_t device, void* stream, ntc_engine** out)
{
	if (!out) return fail(NTC_ERR_ARG, "ntc_hll_create: null argument");
	*out = nullptr;
	if (k < 1 || k > kMaxK) return fail(NTC_ERR_ARG, "ntc_hll_create: k=%u outside 1..%u", k, kMaxK);
	if (n_bits < 4 || n_bits > 24) return fail(NTC_ERR_ARG, "ntc_hll_create: n_bits %u outside 4..24", n_bits);
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
		return fail(NTC_ERR_DEVICE, "ntc_hll_create: no HIP device available (this library has no CPU fallback)");
	if (device < 0 || device >= ndev) return fail(NTC_ERR_ARG, "ntc_hll_create: device %d of %d", device, ndev);
	HIP_TRY(hipSetDevice(device));
	if (int rc = ensure_kernel_attrs(device)) return rc;
	ntc_engine* e = new (std::nothrow) ntc_engine();
	if (!e) return fail(NTC_ERR_MEMORY, "ntc_hll_create: out of host memory");
	e->device = device;
	e->stream = (hipStream_t)stream;
	e->klist.assign(1, k);
	e->r_bits = 27;
	e->s_bits = 7;
	e->hll_bits = n_bits;
	e->kernel_kind = KIND_HF;
	std::vector<uint32_t> t1((size_t)ntc::t2_pairs(k) * 64);
	ntc::build_t2(k, t1.data());
	void* d = nullptr;
	if (hipMalloc((void**)&e->d_sketch, sizeof(uint32_t) << n_bits) != hipSuccess ||
	    hipMalloc((void**)&e->d_f1, 8) != hipSuccess || hipMalloc((void**)&e->d_hll_thr, 4) != hipSuccess ||
	    hipMalloc(&d, t1.size() * 4) != hipSuccess ||
	    hipMemcpy(d, t1.data(), t1.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
		ntc_destroy(e);
		return fail(NTC_ERR_MEMORY, "ntc_hll_create: device allocation failed");
	}
	e->own_sketch = e->own_f1 = true;
	e->d_t1.push_back(d);
	e->hfk.resize(e->klist.size());
	for (size_t ki = 0; ki < e->klist.size(); ++ki)
		fill_hfk(e->hfk[ki], e->klist[ki], e->d_sketch + ki * e->plane_elems(), e->d_f1 + ki, e->d_t1[ki]);
	int rc = ntc_reset(e);
	if (rc) {
		ntc_destroy(e);
		return rc;
	}
	*out = e;
	return 0;
}

int ntc_hll_finish(ntc_engine* e, uint8_t* regs_out, uint64_t* f1_out)
{
	if (!e || !e->hll_bits) return fail(NTC_ERR_STATE, "ntc_hll_finish: not an nthll engine");
	std::lock_guard<std::mutex> lk(e->mu);
	HIP_TRY(hipSetDevice(e->device));
	std::vector<uint32_t> regs((size_t)1 << e->hll_bits);
	HIP_TRY(hipMemcpyAsync(regs.data(), e->d_sketch, regs.size() * 4, hipMemcpyDeviceToHost, e->stream));
	if (f1_out) HIP_TRY(hipMemcpyAsync(f1_out, e->d_f1, 8, hipMemcpyDeviceToHost, e->stream));
	HIP_TRY(hipStreamSynchronize(e->stream));
	if (regs_out)
		for (size_t i = 0; i < regs.size(); ++i)
			regs_out[i] = (uint8_t)regs[i];
	return drain_events(e);
}

int ntc_kernel_time(ntc_engine* e, double* ms_total, uint64_t* launches)
{
	if (!e) return fail(NTC_ERR_ARG, "ntc_kernel_time: null engine");
	std::lock_guard<std::mutex> lk(e->mu);
	HIP_TRY(hipSetDevice(e->device));
	if (int rc = drain_events(e)) return rc;
	if (ms_total) *ms_total = e->ms_total;
	if (launches) *launches = e->launches;
	return 0;
}

int ntc_apply_time(ntc_engine* e, double* ms_total, uint64_t* applies)
{
	if (!e) return fail(NTC_ERR_ARG, "ntc_apply_time: null engine");
	std::lock_guard<std::mutex> lk(e->mu);
	HIP_TRY(hipSetDevice(e->device));
	if (int rc = drain_events(e)) return rc;
	if (ms_total) *ms_total = e->apply_ms;
	if (applies) *applies = e->applies;
	return 0;
}

int ntc_merge_allocations(ntc_engine* e, uint64_t* n)
{
	if (!e || !n) return fail(NTC_ERR_ARG, "ntc_merge_allocations: null argument");
	std::lock_guard<std::mutex> lk(e->mu);
	*n = e->merge_allocs;
	return 0;
}

int ntc_fixup_time(ntc_engine* e, double* ms_total)
{
	if (!e) return fail(NTC_ERR_ARG, "ntc_fixup_time: null engine");
	std::lock_guard<std::mutex> lk(e->mu);
	HIP_TRY(hipSetDevice(e->device));
	if (int rc = drain_events(e)) return rc;
	if (ms_total) *ms_total = e->k1f_ms;
	return 0;
}

int ntc_update_mode(ntc_engine* e, uint32_t* mode_out)
{
	if (!e || !mode_out) return fail(NTC_ERR_ARG, "ntc_update_mode: null argument");
	std::lock_guard<std::mutex> lk(e->mu);
	HIP_TRY(hipSetDevice(e->device));
	*mode_out = 1; // engines without a log increment directly
	if (e->d_logmode) {
		HIP_TRY(hipMemcpyAsync(mode_out, e->d_logmode, 4, hipMemcpyDeviceToHost, e->stream));
		HIP_TRY(hipStreamSynchronize(e->stream));
	}
	return 0;
}

int ntc_set_profiling(ntc_engine* e, int enable)
{
	if (!e) return fail(NTC_ERR_ARG, "ntc_set_profiling: null engine");
	std::lock_guard<std::mutex> lk(e->mu);
	e->profiling = enable != 0;
	return 0;
}

} // extern "C"
