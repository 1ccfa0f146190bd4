// Host helper of the SDRF draw: cdf = numpy.cumsum(e / S) for the probabilities handed to np.random.choice
// (sdrf_no_cuda.py:49-50, utils/softmax.py:9-10), bit for bit.  numpy's cumsum is the SEQUENTIAL float64 recurrence
// acc_{i+1} = fl(acc_i + p_i): 180k dependent adds for a hub edge, three cycles of latency each, 0.15 ms per SDRF
// iteration when done the obvious way.
//
// The recurrence can be evaluated without that latency chain and still exactly.  While acc stays inside one binade
// [2^k, 2^(k+1)) its ulp is u = 2^(k-52) and acc = M u with an integer M in [2^52, 2^53).  For p >= 0 with p / u not
// exactly half-way between two integers, round-to-nearest-even gives fl(M u + p) = (M + rn(p / u)) u: the rounding of
// the sum only rounds the addend to a multiple of u.  Sums of such integers below 2^53 are exact in float64 whatever the
// order, so a block of eight addends is scaled by 1/u (a power of two: exact), rounded to integers, prefix-summed inside
// a vector register, offset by M and scaled back: eight partial sums per ~7 cycles of dependent latency instead of per
// 24.  A block that contains a tie (fraction exactly 1/2: there the result depends on the parity of M), that leaves the
// binade, or that holds anything unusual (NaN, infinity, negative values) is simply done by the plain loop, as are the
// first elements (acc still tiny) and the tail.  tests/test_host_cpu.py compares the result with numpy.cumsum itself on
// adversarial and random vectors.
#include <cmath>
#include <cstdint>
#include <immintrin.h>

#include "dcr.h"

namespace {

inline void scalar_steps(const double *e, int64_t i0, int64_t i1, double S, double &acc, double *cdf) {
    for (int64_t i = i0; i < i1; ++i) {
        const double p = e[i] / S;  // numpy true_divide
        acc = acc + p;              // numpy cumsum: sequential float64 adds
        cdf[i] = acc;
    }
}

__attribute__((target("avx512f,avx512dq"))) void cdf_avx512(const double *e, int64_t n, double S, double *cdf) {
    double acc = 0.0;
    int64_t i = 0;
    const __m512d vS = _mm512_set1_pd(S);
    const __m512d half = _mm512_set1_pd(0.5), lo = _mm512_set1_pd(0x1p52), hi = _mm512_set1_pd(0x1p53);
    const __m512i zero = _mm512_setzero_si512();
    while (i < n) {
        if (!(acc >= 0x1p-900 && acc < 0x1p900) || i + 8 > n) {  // tiny, huge, NaN, or the tail: plain loop
            const int64_t stop = i + 8 < n ? i + 8 : n;
            scalar_steps(e, i, stop, S, acc, cdf);
            i = stop;
            continue;
        }
        int ex;
        (void)std::frexp(acc, &ex);                 // acc = f 2^ex, f in [0.5, 1): binade [2^(ex-1), 2^ex)
        const double u = std::ldexp(1.0, ex - 53);  // ulp of the binade
        const __m512d vu = _mm512_set1_pd(u), vinv = _mm512_set1_pd(std::ldexp(1.0, 53 - ex));
        __m512d carry = _mm512_set1_pd(acc * std::ldexp(1.0, 53 - ex));  // M, an integer in [2^52, 2^53)
        bool left = false;
        while (i + 8 <= n) {
            const __m512d p = _mm512_div_pd(_mm512_loadu_pd(e + i), vS);
            const __m512d q = _mm512_mul_pd(p, vinv);
            const __m512d r = _mm512_roundscale_pd(q, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC);
            const __mmask8 tie = _mm512_cmp_pd_mask(_mm512_abs_pd(_mm512_sub_pd(q, r)), half, _CMP_EQ_OQ);
            // inclusive prefix sum of the eight integers
            __m512d s = r;
            s = _mm512_add_pd(s, _mm512_castsi512_pd(_mm512_alignr_epi64(_mm512_castpd_si512(s), zero, 7)));
            s = _mm512_add_pd(s, _mm512_castsi512_pd(_mm512_alignr_epi64(_mm512_castpd_si512(s), zero, 6)));
            s = _mm512_add_pd(s, _mm512_castsi512_pd(_mm512_alignr_epi64(_mm512_castpd_si512(s), zero, 4)));
            const __m512d total = _mm512_add_pd(carry, s);
            // every partial sum still inside the binade, every addend non-negative and finite, no tie
            const __mmask8 inside = _mm512_cmp_pd_mask(total, hi, _CMP_LT_OQ) & _mm512_cmp_pd_mask(total, lo, _CMP_GE_OQ) &
                                    _mm512_cmp_pd_mask(r, _mm512_setzero_pd(), _CMP_GE_OQ);
            if (tie != 0 || inside != 0xFF) {
                left = true;
                break;
            }
            _mm512_storeu_pd(cdf + i, _mm512_mul_pd(total, vu));
            carry = _mm512_permutexvar_pd(_mm512_set1_epi64(7), total);
            i += 8;
        }
        acc = _mm512_cvtsd_f64(carry) * u;  // exact: integer below 2^53 times a power of two
        if (left) {                          // the block that stopped the vector loop
            const int64_t stop = i + 8 < n ? i + 8 : n;
            scalar_steps(e, i, stop, S, acc, cdf);
            i = stop;
        }
    }
}

}  // namespace

extern "C" int dcr_host_cdf_from_exp(const double *e, int64_t n, double S, double *cdf, double *out_total) {
    if (!e || !cdf || !out_total || n <= 0) return DCR_EINVAL;
    static const bool simd = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq");
    if (simd && n >= 64) {
        cdf_avx512(e, n, S, cdf);
    } else {
        double acc = 0.0;
        scalar_steps(e, 0, n, S, acc, cdf);
    }
    *out_total = cdf[n - 1];
    return DCR_OK;
}

// the plain loop alone, for tests and for timing the two against each other
extern "C" int dcr_host_cdf_from_exp_plain(const double *e, int64_t n, double S, double *cdf, double *out_total) {
    if (!e || !cdf || !out_total || n <= 0) return DCR_EINVAL;
    double acc = 0.0;
    scalar_steps(e, 0, n, S, acc, cdf);
    *out_total = cdf[n - 1];
    return DCR_OK;
}
