// Stage 2 of the encoder: from autocorrelations and fixed-predictor error sums to the list
// of residual-coding candidates of every signal.
//
// Replaces, for one signal:
//   fixed.c:245-269          guess of the fixed order + bits-per-sample estimates (log)
//   lpc.c:112-154            Levinson-Durbin recursion in double, all orders
//   lpc.c:1325-1375          expected bits per residual sample, best-order guess
//   lpc.c:156-263            coefficient quantisation (shift, error feedback rounding)
//   stream_encoder.c:3241-3354  which (order, precision) pairs are tried and the Rice
//                            parameter suggested to each ("rice_parameter")
// i.e. every floating-point decision of process_subframe_.  The integer work (residuals,
// partition sums, Rice search) is stage 3.
//
// One lane per signal: each of these recursions is a short strictly sequential chain in
// double whose rounding must match the reference (no reassociation, no contraction), so
// there is nothing to spread over a wavefront; the batch supplies the parallelism
// (frames x signals lanes).  A signal's candidate records lie one after the other (header,
// coefficients, their sum: a whole number of 16-byte pieces each); every lane writes its own
// signal's, a record at a time in 16-byte stores.
#include "engine_types.h"

#define FA_LN2 0.69314718055994530942

struct CandWriter {
	uint32_t *hdr; int32_t *q; uint32_t stride, n, max_cand, qrows;   // hdr / q already offset to this signal's records; qrows: row of the coefficient sum
	__device__ __forceinline__ void emit(uint32_t h) { if(n < max_cand) hdr[(size_t)n * stride] = h; n++; }
	__device__ __forceinline__ void coef(uint32_t j, int32_t v) { if(n < max_cand) q[(size_t)n * stride + j] = v; }
};

// lpc.c:1336-1351
__device__ __forceinline__ double expected_bits_scaled(double err, double scale)
{
	if(err > 0.0) { const double b = 0.5 * log(scale * err) / FA_LN2; return b >= 0.0 ? b : 0.0; }
	return err < 0.0 ? 1e32 : 0.0;
}

// fixed.c:256-268: bits per residual sample expected of a fixed predictor whose |residual| sum is e over len samples
__device__ __forceinline__ float fixed_expected_bits(unsigned long long e, double len)
{
	return (float)((e > 0) ? log(FA_LN2 * (double)e / len) / FA_LN2 : 0.0);
}

// Quantise coefficients c[0..order) at `precision` bits and emit one lpc candidate
// (lpc.c:156-263; precision clamp and datapath choice stream_encoder.c:3587-3603).
template<int MAXORD>
__device__ __forceinline__ void quantize_and_emit(CandWriter &cw, const float (&cf)[MAXORD], uint32_t order, uint32_t precision,
	uint32_t bps, uint32_t rice_k)
{
	uint32_t eff = precision;
	const uint32_t lg = fa_ilog2(order);
	if(bps <= 16) { const uint32_t lim = 32u - bps - lg; eff = eff < lim ? eff : lim; }
	const uint32_t pm1 = eff - 1u;
	const int32_t qmax = (int32_t)((1u << pm1) - 1u), qmin = -(int32_t)(1u << pm1);
	double cmax = 0.0;
#pragma unroll
	for(int j = 0; j < MAXORD; j++) if((uint32_t)j < order) { const double d = fabs((double)cf[j]); if(d > cmax) cmax = d; }
	if(!(cmax > 0.0)) return;                       // lpc.c:179-182 (return 2)
	int log2cmax;
	(void)frexp(cmax, &log2cmax);
	log2cmax--;
	int shift = (int)pm1 - log2cmax - 1;
	if(shift > 15) shift = 15; else if(shift < -16) return; // lpc.c:192-195 (return 1)
	// the record is put together in registers and goes out in 16-byte stores: every lane writes its own signal's records,
	// so a store instruction touches 64 cache lines whatever its width
	uint32_t rec[((MAXORD + 4) / 4) * 4];
#pragma unroll
	for(int j = 0; j < ((MAXORD + 4) / 4) * 4; j++) rec[j] = 0;
	double error = 0.0;
	uint32_t sum_abs = 0;   // of the quantised coefficients (each below 2^15, at most 32 of them)
	int32_t sum_q = 0;
	if(shift >= 0) {
		const float scale = (float)(1 << shift);
#pragma unroll
		for(int j = 0; j < MAXORD; j++) if((uint32_t)j < order) {
			error += (double)__fmul_rn(cf[j], scale);   // float * int product, then widened (lpc.c:202)
			int32_t v = error >= 0.0 ? (int32_t)(error + 0.5) : (int32_t)(error - 0.5);
			v = v > qmax ? qmax : (v < qmin ? qmin : v);
			error -= (double)v;
			rec[1 + j] = (uint32_t)v;
			sum_abs += (uint32_t)(v < 0 ? -v : v); sum_q += v;
		}
	}
	else {
		const float scale = (float)(1 << (-shift));
#pragma unroll
		for(int j = 0; j < MAXORD; j++) if((uint32_t)j < order) {
			error += (double)__fdiv_rn(cf[j], scale);   // lpc.c:237
			int32_t v = error >= 0.0 ? (int32_t)(error + 0.5) : (int32_t)(error - 0.5);
			v = v > qmax ? qmax : (v < qmin ? qmin : v);
			error -= (double)v;
			rec[1 + j] = (uint32_t)v;
			sum_abs += (uint32_t)(v < 0 ? -v : v); sum_q += v;
		}
		shift = 0;
	}
	const uint32_t wide = (bps + eff + lg <= 32u) ? 0u : 1u;
	// |prediction| <= (sum_abs * 2^(bps - 1)) >> shift for samples of bps bits
	const uint32_t bounded = ((((unsigned long long)sum_abs << (bps - 1u)) >> shift) < (1ull << 27)) ? 1u : 0u;
	rec[0] = FA_CAND_PACK(1u, order, eff, (uint32_t)shift, rice_k, wide) | (bounded << 23);
	if(cw.n < cw.max_cand) {
		uint4 *dst = (uint4*)(cw.hdr + (size_t)cw.n * cw.stride);   // records start on 16-byte borders and are whole numbers of them long
#pragma unroll
		for(int t = 0; t < (MAXORD + 4) / 4; t++) if((uint32_t)(4 * t) <= order) dst[t] = make_uint4(rec[4 * t], rec[4 * t + 1], rec[4 * t + 2], rec[4 * t + 3]);
		cw.q[(size_t)cw.n * cw.stride + cw.qrows] = sum_q;   // (behind the coefficients of the longest predictor: after the stores above, which may cover its place)
	}
	cw.n++;
}

// One Levinson step i (order i+1): updates lpc[], err; lpc.c:124-146.
template<int MAXORD, int I>
__device__ __forceinline__ void levinson_step(const float (&autoc)[MAXORD + 1], double (&lpc)[MAXORD], double &err)
{
	double r = -(double)autoc[I + 1];
#pragma unroll
	for(int j = 0; j < I; j++) r -= lpc[j] * (double)autoc[I - j];
	r /= err;
	lpc[I] = r;
#pragma unroll
	for(int j = 0; j < (I >> 1); j++) { const double t = lpc[j]; lpc[j] += r * lpc[I-1-j]; lpc[I-1-j] += r * t; }
	if(I & 1) lpc[I >> 1] += lpc[I >> 1] * r;
	err *= (1.0 - r * r);
}

template<int MAXORD, int I>
struct LevinsonRun {
	// Runs steps I..MAXORD-1 while I < max_order. MODE 0: record errors only.
	// MODE 1: emit candidates for every order in [lo, max_order] (exhaustive) or for order == lo only.
	template<int MODE>
	__device__ static __forceinline__ void run(const float (&autoc)[MAXORD + 1], double (&lpc)[MAXORD], double &err, double (&errs)[MAXORD],
		uint32_t &max_order, CandWriter &cw, const FaCfg &c, uint32_t n, uint32_t bps, uint32_t lo)
	{
		if((uint32_t)I >= max_order) return;
		levinson_step<MAXORD, I>(autoc, lpc, err);
		errs[I] = err;
		if(MODE == 1) {
			const uint32_t order = I + 1u;
			if(order >= lo && order < n) {  // orders are also capped at n-1 (stream_encoder.c:3327-3328)
				// stream_encoder.c:3330-3340
				const double eb = expected_bits_scaled(err, 0.5 * FA_LN2 * FA_LN2 / (double)(n - order));
				if(!(eb >= (double)bps)) {
					uint32_t k = (eb > 0.0) ? (uint32_t)(eb + 0.5) : 0u;
					k++;
					if(k >= c.k_limit) k = c.k_limit - 1u;
					float cf[MAXORD];
#pragma unroll
					for(int j = 0; j < MAXORD; j++) cf[j] = (j <= I) ? (float)(-lpc[j]) : 0.0f; // lpc.c:145
					uint32_t plo, phi;
					if(c.prec_search) { // stream_encoder.c:3341-3350
						plo = FA_MIN_QLP_PRECISION;
						// unsigned arithmetic on purpose: 32 - bps - order wraps for large orders and the min() then yields 15, as in the reference
						if(bps <= 17) { phi = 32u - bps - order; if(phi > FA_MAX_QLP_PRECISION) phi = FA_MAX_QLP_PRECISION; if(phi < plo) phi = plo; }
						else phi = FA_MAX_QLP_PRECISION;
					}
					else plo = phi = c.qlp_precision;
					for(uint32_t prec = plo; prec <= phi; prec++) quantize_and_emit<MAXORD>(cw, cf, order, prec, bps, k);
				}
			}
		}
		if(err == 0.0) { max_order = I + 1u; return; } // lpc.c:149-152
		LevinsonRun<MAXORD, I + 1>::template run<MODE>(autoc, lpc, err, errs, max_order, cw, c, n, bps, lo);
	}
};
template<int MAXORD>
struct LevinsonRun<MAXORD, MAXORD> {
	template<int MODE>
	__device__ static __forceinline__ void run(const float (&)[MAXORD + 1], double (&)[MAXORD], double &, double (&)[MAXORD],
		uint32_t &, CandWriter &, const FaCfg &, uint32_t, uint32_t, uint32_t) {}
};

template<int MAXORD>
__global__ void __launch_bounds__(64) fa_model_kernel(FaCfg c, FaWork w)
{
	const uint32_t g = blockIdx.x * 64u + threadIdx.x;
	if(g >= c.G) return;
	const uint32_t f = g / c.nsig, s = g - f * c.nsig;
	const uint32_t n = fa_frame_len(c, w, f);
	const uint32_t wasted = w.wasted[g];
	const uint32_t bps = c.bps - wasted + ((c.midside && s == 3u) ? 1u : 0u);

	CandWriter cw; cw.hdr = w.cand_hdr + fa_cand_idx(c, g, 0); cw.q = w.cand_q + fa_cand_idx(c, g, 0); cw.stride = c.cand_stride; cw.n = 0; cw.max_cand = c.max_cand; cw.qrows = c.max_lpc_order ? c.max_lpc_order : 1u;

	const bool constant = (!c.disable_constant) && (w.is_const[g] & 1u);
	if(n >= FA_MAX_FIXED_ORDER && !constant) {
		// ---- fixed predictors (fixed.c:245-269, stream_encoder.c:3241-3293)
		float fb[5]; unsigned long long e[5];
		const double len = (double)(n - FA_MAX_FIXED_ORDER);
#pragma unroll
		for(int k = 0; k < 5; k++) {
			e[k] = w.fixed_err[(size_t)k * c.G + g];
			fb[k] = fixed_expected_bits(e[k], len);
		}
		uint32_t guess;
		{
			const unsigned long long m34 = e[3] < e[4] ? e[3] : e[4], m234 = e[2] < m34 ? e[2] : m34, m1234 = e[1] < m234 ? e[1] : m234;
			if(e[0] < m1234) guess = 0; else if(e[1] < m234) guess = 1; else if(e[2] < m34) guess = 2; else if(e[3] < e[4]) guess = 3; else guess = 4;
		}
		// the verbatim baseline is "infinite" when verbatim is disabled (stream_encoder.c:3206)
		const bool best_is_max = c.disable_verbatim != 0;
		if(!c.disable_fixed || (c.max_lpc_order == 0 && best_is_max)) {
			uint32_t lo = guess, hi = guess;
			if(c.exhaustive) { lo = 0; hi = FA_MAX_FIXED_ORDER; }
			if(hi >= n) hi = n - 1u;
#pragma unroll
			for(uint32_t o = 0; o <= FA_MAX_FIXED_ORDER; o++) {
				if(o < lo || o > hi) continue;
				if(fb[o] >= (float)bps) continue;
				uint32_t k = (fb[o] > 0.0f) ? (uint32_t)((double)fb[o] + 0.5) : 0u;
				k++;
				if(k >= c.k_limit) k = c.k_limit - 1u;
				cw.emit(FA_CAND_PACK(0u, o, 0u, 0u, k, 0u));
			}
		}
		// ---- linear prediction (stream_encoder.c:3297-3387)
		if(c.max_lpc_order > 0) {
			uint32_t max_order = c.max_lpc_order >= n ? n - 1u : c.max_lpc_order;
			if(max_order > 0) {
				for(uint32_t a = 0; a < c.num_apod; a++) {
					float autoc[MAXORD + 1];
#pragma unroll
					for(int l = 0; l <= MAXORD; l++) {
						autoc[l] = 0.0f;
						if((uint32_t)l < c.lag) {
							// the autocorrelation stage ran on the unshifted signal: exactly 4^wasted too large
							float *slot = &w.autoc[((size_t)a * c.lag + l) * c.G + g];
							autoc[l] = ldexpf(*slot, -2 * (int)wasted);
							if(wasted) *slot = autoc[l];
						}
					}
					if(autoc[0] == 0.0f) continue;
					double lpc[MAXORD], errs[MAXORD], err = (double)autoc[0];
#pragma unroll
					for(int j = 0; j < MAXORD; j++) { lpc[j] = 0.0; errs[j] = 0.0; }
					if(c.exhaustive) {
						LevinsonRun<MAXORD, 0>::template run<1>(autoc, lpc, err, errs, max_order, cw, c, n, bps, 1u);
						if(max_order >= n) max_order = n - 1u;
					}
					else {
						// pass 1: prediction errors of every order, then the order guess (lpc.c:1353-1375)
						LevinsonRun<MAXORD, 0>::template run<0>(autoc, lpc, err, errs, max_order, cw, c, n, bps, 0u);
						const uint32_t overhead = bps + (c.prec_search ? FA_MIN_QLP_PRECISION : c.qlp_precision);
						const double scale = 0.5 * FA_LN2 * FA_LN2 / (double)n;
						double best = 4294967295.0; uint32_t best_i = 0;
#pragma unroll
						for(int i = 0; i < MAXORD; i++) if((uint32_t)i < max_order) {
							const double bits = expected_bits_scaled(errs[i], scale) * (double)(n - (uint32_t)(i + 1)) + (double)((uint32_t)(i + 1) * overhead);
							if(bits < best) { best = bits; best_i = (uint32_t)i; }
						}
						max_order = best_i + 1u;
						if(max_order >= n) max_order = n - 1u;
						// pass 2: recompute the recursion up to the guessed order and emit it
						err = (double)autoc[0];
						uint32_t mo2 = max_order;
						LevinsonRun<MAXORD, 0>::template run<1>(autoc, lpc, err, errs, mo2, cw, c, n, bps, max_order);
					}
				}
			}
		}
	}
	w.ncand[g] = cw.n < c.max_cand ? cw.n : c.max_cand;
}

extern "C" void fa_launch_model(const FaCfg *c, const FaWork *w, hipStream_t st)
{
	const dim3 grid((c->G + 63u) / 64u), block(64);
	if(c->max_lpc_order <= 12) hipLaunchKernelGGL(fa_model_kernel<12>, grid, block, 0, st, *c, *w);
	else hipLaunchKernelGGL(fa_model_kernel<32>, grid, block, 0, st, *c, *w);
}

// ---- the two logarithms above, on inputs of the caller's choosing.  Every float decision downstream of them (the
// Rice parameter suggested to a candidate, the "not worth trying" skip, the order guess) goes through these two
// expressions, and log() is the one operation in them that IEEE 754 does not pin: the parity tests sweep the
// neighbourhood of every decision boundary through this entry point and compare with the host's libm.
__global__ void __launch_bounds__(256) fa_model_logs_kernel(uint32_t n, const unsigned long long *__restrict__ e, const double *__restrict__ len,
	const double *__restrict__ err, const double *__restrict__ scale, float *__restrict__ fb, double *__restrict__ eb)
{
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if(i >= n) return;
	fb[i] = fixed_expected_bits(e[i], len[i]);
	eb[i] = expected_bits_scaled(err[i], scale[i]);
}

extern "C" int flacamd_debug_model_logs(int device, uint32_t n, const uint64_t *h_e, const double *h_len, const double *h_err, const double *h_scale, float *h_fb, double *h_eb)
{
	if(!n || !h_e || !h_len || !h_err || !h_scale || !h_fb || !h_eb) return FLACAMD_E_INVALID_ARGUMENT;
	int ndev = 0;
	if(hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return FLACAMD_E_NO_DEVICE;
	if(hipSetDevice(device) != hipSuccess) return FLACAMD_E_DEVICE;
	uint8_t *d = 0;
	const size_t N = n;
	if(hipMalloc((void**)&d, N * (8 + 8 + 8 + 8 + 4 + 8)) != hipSuccess) return FLACAMD_E_OUT_OF_MEMORY;
	unsigned long long *de = (unsigned long long*)d; double *dl = (double*)(d + N * 8), *dr = (double*)(d + N * 16), *ds = (double*)(d + N * 24), *deb = (double*)(d + N * 32); float *dfb = (float*)(d + N * 40);
	int rc = FLACAMD_OK;
	if(hipMemcpy(de, h_e, N * 8, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(dl, h_len, N * 8, hipMemcpyHostToDevice) != hipSuccess
		|| hipMemcpy(dr, h_err, N * 8, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(ds, h_scale, N * 8, hipMemcpyHostToDevice) != hipSuccess) rc = FLACAMD_E_DEVICE;
	if(rc == FLACAMD_OK) {
		hipLaunchKernelGGL(fa_model_logs_kernel, dim3((n + 255u) / 256u), dim3(256), 0, 0, n, de, dl, dr, ds, dfb, deb);
		if(hipMemcpy(h_fb, dfb, N * 4, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(h_eb, deb, N * 8, hipMemcpyDeviceToHost) != hipSuccess) rc = FLACAMD_E_DEVICE;
	}
	(void)hipFree(d);
	return rc;
}
