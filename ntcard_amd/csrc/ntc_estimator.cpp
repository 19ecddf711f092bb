// ntc_estimator.cpp — host side of the seam: compEst's closed-form recurrence and the .hist writer.
//
// Restates ntcard.cpp:249-274 (estimator from the value histogram p[2][65536], produced on the
// device by K2) and ntcard.cpp:283,291-294 (output format).  Plain IEEE double, same operation
// order as the reference; this file is compiled with -ffp-contract=off (no FMA contraction).
// Only i <= cov_max is evaluated: f_i depends on p[0..i] and f_1..f_{i-1} only, so the values are
// the ones the reference's full 65 536-step loop produces, without its ~2.3 s fixed cost.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <sys/types.h>

#include "../../include/ntcard_hip.h"

int ntc_internal_fail(int code, const char* fmt, ...); // ntc_engine.hip: sets ntc_last_error()

extern "C" {

int ntc_estimate(const uint32_t* p_hist, uint32_t r_bits, uint32_t s_bits, uint32_t cov_max, double* F0_out,
                 double* f_out)
{
	if (!p_hist || !F0_out || !f_out) return ntc_internal_fail(NTC_ERR_ARG, "ntc_estimate: null argument");
	if (cov_max > 65535) cov_max = 65535; // ntcard.cpp:340-344
	const unsigned n_samp = 2;
	std::vector<double> p_mean(cov_max + 1, 0.0);
	for (size_t i = 0; i <= cov_max; ++i) {
		double acc = 0.0;
		for (size_t j = 0; j < n_samp; ++j)
			acc += p_hist[j * 65536 + i];
		acc /= 1.0 * n_samp;
		p_mean[i] = acc;
	}
	const double ln_p0 = std::log(p_mean[0]);
	const double F0 =
	    (ssize_t)((r_bits * std::log(2) - ln_p0) * 1.0 * ((uint64_t)1 << (s_bits + r_bits)));
	*F0_out = F0;
	for (size_t i = 0; i <= cov_max; ++i)
		f_out[i] = 0;
	const double denom = p_mean[0] * (ln_p0 - r_bits * std::log(2));
	if (denom == 0) return 0;
	if (cov_max >= 1) f_out[1] = -1.0 * p_mean[1] / denom;
	for (size_t i = 2; i <= cov_max; ++i) {
		double sum = 0.0;
		for (size_t j = 1; j < i; ++j)
			sum += j * p_mean[i - j] * f_out[j];
		f_out[i] = -1.0 * p_mean[i] / denom - sum / (i * p_mean[0]);
	}
	for (size_t i = 1; i <= cov_max; ++i)
		f_out[i] = (double)std::labs((long)(ssize_t)(f_out[i] * F0));
	return 0;
}

// nthll.cpp:247-254: alpha * m^2 / sum_j 2^-M[j], alpha halved because the hashes are canonical
int ntc_hll_estimate(const uint8_t* regs, uint32_t n_bits, double* est_out)
{
	if (!regs || !est_out || n_bits > 31) return ntc_internal_fail(NTC_ERR_ARG, "ntc_hll_estimate: null argument or n_bits %u > 31", n_bits);
	const unsigned n_buck = 1u << n_bits;
	double alpha = 1.4426 / (1 + 1.079 / n_buck);
	alpha /= 2;
	double p_est = 0.0;
	for (unsigned j = 0; j < n_buck; ++j)
		p_est += 1.0 / ((uint64_t)1 << regs[j]);
	const double z_est = 1.0 / p_est;
	*est_out = alpha * n_buck * n_buck * z_est;
	return 0;
}

int ntc_write_hist(const char* path, uint64_t f1, double F0, const double* f, uint32_t cov_max)
{
	if (!path || !f) return ntc_internal_fail(NTC_ERR_ARG, "ntc_write_hist: null argument");
	FILE* out = std::fopen(path, "w");
	if (!out) return ntc_internal_fail(NTC_ERR_ARG, "ntc_write_hist: cannot open %s for writing", path);
	std::fprintf(out, "F1\t%llu\n", (unsigned long long)f1);
	std::fprintf(out, "F0\t%llu\n", (unsigned long long)(uint64_t)F0);
	for (uint32_t i = 1; i <= cov_max; ++i)
		std::fprintf(out, "%u\t%llu\n", i, (unsigned long long)(uint64_t)f[i]);
	return std::fclose(out) == 0 ? 0 : ntc_internal_fail(NTC_ERR_ARG, "ntc_write_hist: write to %s failed", path);
}

} // extern "C"
