// Register-window residual evaluation shared by the search and pack kernels.
//
// A thread holds a chunk of consecutive samples of a signal (CH in the search kernel, any
// W - HIST elsewhere) plus HIST samples of history in registers (xw[HIST + s] is sample s of
// the chunk) and evaluates a predictor on all CH
// samples with fully unrolled code: coefficients are wave-uniform, every tap is one VALU
// instruction.  Orders above HIST take the callers' generic per-sample path.
//
// Restates FLAC__fixed_compute_residual (fixed.c:352-393),
// FLAC__lpc_compute_residual_from_qlp_coefficients (lpc.c:265-529, 32-bit wrapping sum) and
// ..._wide (lpc.c:531-791, 64-bit sum); which of the two applies is the reference's choice
// (stream_encoder.c:3597-3603), made by the model kernel and carried in the candidate.
#pragma once
#include <stdint.h>

#define CH 16
#define HIST 12

enum { MAC_I24 = 0, MAC_I32 = 1, MAC_WIDE = 2 };

// LDS index of signal sample i (i >= -32): one pad word per 16 samples so that lanes
// reading windows at a 16-sample stride hit distinct banks.
__device__ __forceinline__ uint32_t xs_idx(int32_t i) { const uint32_t u = (uint32_t)(i + 32); return u + (u >> 4); }
// words needed for a signal of n samples in that layout
__host__ __device__ inline uint32_t xs_words_for(uint32_t n) { return ((n + 32u) * 17u) / 16u + 17u; }

__device__ __forceinline__ uint32_t abs_u32(int32_t r) { return (uint32_t)(r < 0 ? 0u - (uint32_t)r : (uint32_t)r); }

// d = a * b + c on the 24-bit multiplier; the low 32 bits are exact for |a|, |b| < 2^23.
// Written as asm so that a predictor stays one instruction per tap (the compiler otherwise
// splits the chain into two multiplies and a three-operand add per pair of taps).
__device__ __forceinline__ int32_t mad24(int32_t a, int32_t b, int32_t c)
{
	int32_t d;
	asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
	return d;
}

// d = a.lo16 * b.lo16 + a.hi16 * b.hi16 + c (signed halves, 32-bit wrapping sum): two predictor
// taps per instruction when samples and coefficients fit 16 bits
__device__ __forceinline__ int32_t dot2_i16(uint32_t a, uint32_t b, int32_t c)
{
	int32_t d;
	asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
	return d;
}
// d = |a - b| + c on unsigned operands
__device__ __forceinline__ uint32_t sad_u32(uint32_t a, uint32_t b, uint32_t c)
{
	uint32_t d;
	asm("v_sad_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
	return d;
}
// (lo16 of lo) | (lo16 of hi) << 16
__device__ __forceinline__ uint32_t pack_lo16(int32_t lo, int32_t hi) { return __builtin_amdgcn_perm((uint32_t)hi, (uint32_t)lo, 0x05040100u); }

// f(s, x, pred, P): sample s of the chunk and its order-P prediction
template<int P, int MODE, int W, class F>
__device__ __forceinline__ void lpc_pred_each(const int32_t (&xw)[W], const int32_t (&q)[HIST], uint32_t shift, F &&f)
{
#pragma unroll
	for(int s = 0; s < W - HIST; s++) {
		int32_t pred;
		if(MODE == MAC_WIDE) {
			long long acc = 0;
#pragma unroll
			for(int j = 0; j < P; j++) acc += (long long)q[j] * (long long)xw[HIST + s - 1 - j];
			pred = (int32_t)(acc >> shift);
		}
		else if(MODE == MAC_I24) {
			int32_t acc = 0;
#pragma unroll
			for(int j = 0; j < P; j++) acc = mad24(q[j], xw[HIST + s - 1 - j], acc);
			pred = acc >> shift;
		}
		else {
			// the wrapping 32-bit sum is the low word of the 64-bit one: one v_mad_i64_i32 per tap instead of a multiply and an add
			long long acc = 0;
#pragma unroll
			for(int j = 0; j < P; j++) acc += (long long)q[j] * (long long)xw[HIST + s - 1 - j];
			pred = (int32_t)(uint32_t)acc >> shift;
		}
		f(s, xw[HIST + s], pred, P);
	}
}

// the same on packed pairs: xp[t] = pack_lo16(xw[t], xw[t-1]), qp[m] = pack_lo16(q[2m], q[2m+1]);
// exact (mod 2^32, like the 32-bit path of lpc.c:296) when samples and coefficients fit 16 bits
template<int P, int W, class F>
__device__ __forceinline__ void lpc_pred_each_dot2(const int32_t (&xw)[W], const uint32_t (&xp)[W], const uint32_t (&qp)[HIST / 2], uint32_t shift, F &&f)
{
#pragma unroll
	for(int s = 0; s < W - HIST; s++) {
		int32_t acc = 0;
#pragma unroll
		for(int m = 0; m < (P + 1) / 2; m++) acc = dot2_i16(qp[m], xp[HIST + s - 1 - 2 * m], acc);
		f(s, xw[HIST + s], acc >> shift, P);
	}
}

// the same sums handed over unshifted, starting from `init`: f(s, sum, P)
template<int P, int W, class F>
__device__ __forceinline__ void lpc_sum_each_dot2(const uint32_t (&xp)[W], const uint32_t (&qp)[HIST / 2], int32_t init, F &&f)
{
#pragma unroll
	for(int s = 0; s < W - HIST; s++) {
		int32_t acc = init;
#pragma unroll
		for(int m = 0; m < (P + 1) / 2; m++) acc = dot2_i16(qp[m], xp[HIST + s - 1 - 2 * m], acc);
		f(s, acc, P);
	}
}

#define RESIDUAL_ORDER_SWITCH(order, CALL) \
	switch(order) { \
	case 1: CALL(1); break; case 2: CALL(2); break; case 3: CALL(3); break; case 4: CALL(4); break; \
	case 5: CALL(5); break; case 6: CALL(6); break; case 7: CALL(7); break; case 8: CALL(8); break; \
	case 9: CALL(9); break; case 10: CALL(10); break; case 11: CALL(11); break; default: CALL(12); break; }

template<int MODE, int W, class F>
__device__ __forceinline__ void lpc_pred_each_dispatch(uint32_t order, const int32_t (&xw)[W], const int32_t (&q)[HIST], uint32_t shift, F &&f)
{
#define RESIDUAL_CALL(P) lpc_pred_each<P, MODE>(xw, q, shift, f)
	RESIDUAL_ORDER_SWITCH(order, RESIDUAL_CALL)
#undef RESIDUAL_CALL
}
template<int W, class F>
__device__ __forceinline__ void lpc_pred_each_dot2_dispatch(uint32_t order, const int32_t (&xw)[W], const uint32_t (&xp)[W], const uint32_t (&qp)[HIST / 2], uint32_t shift, F &&f)
{
#define RESIDUAL_CALL(P) lpc_pred_each_dot2<P>(xw, xp, qp, shift, f)
	RESIDUAL_ORDER_SWITCH(order, RESIDUAL_CALL)
#undef RESIDUAL_CALL
}

template<int W, class F>
__device__ __forceinline__ void lpc_sum_each_dot2_dispatch(uint32_t order, const uint32_t (&xp)[W], const uint32_t (&qp)[HIST / 2], int32_t init, F &&f)
{
#define RESIDUAL_CALL(P) lpc_sum_each_dot2<P>(xp, qp, init, f)
	RESIDUAL_ORDER_SWITCH(order, RESIDUAL_CALL)
#undef RESIDUAL_CALL
}

// f(s, r, P): residual r of chunk sample s under an order-P predictor
template<int P, int MODE, int W, class F>
__device__ __forceinline__ void lpc_each(const int32_t (&xw)[W], const int32_t (&q)[HIST], uint32_t shift, F &&f)
{
#pragma unroll
	for(int s = 0; s < W - HIST; s++) {
		int32_t pred;
		if(MODE == MAC_WIDE) {
			long long acc = 0;
#pragma unroll
			for(int j = 0; j < P; j++) acc += (long long)q[j] * (long long)xw[HIST + s - 1 - j];
			pred = (int32_t)(acc >> shift);
		}
		else if(MODE == MAC_I24) {
			int32_t acc = 0;
#pragma unroll
			for(int j = 0; j < P; j++) acc = mad24(q[j], xw[HIST + s - 1 - j], acc); // low 32 bits of the sum: the wrap of lpc.c:296
			pred = acc >> shift;
		}
		else {
			long long acc = 0;   // (low word = the wrapping 32-bit sum, as above)
#pragma unroll
			for(int j = 0; j < P; j++) acc += (long long)q[j] * (long long)xw[HIST + s - 1 - j];
			pred = (int32_t)(uint32_t)acc >> shift;
		}
		f(s, (int32_t)((uint32_t)xw[HIST + s] - (uint32_t)pred), P);
	}
}

template<int MODE, int W, class F>
__device__ __forceinline__ void lpc_each_dispatch(uint32_t order, const int32_t (&xw)[W], const int32_t (&q)[HIST], uint32_t shift, F &&f)
{
	switch(order) {
	case 1: lpc_each<1, MODE>(xw, q, shift, f); break;
	case 2: lpc_each<2, MODE>(xw, q, shift, f); break;
	case 3: lpc_each<3, MODE>(xw, q, shift, f); break;
	case 4: lpc_each<4, MODE>(xw, q, shift, f); break;
	case 5: lpc_each<5, MODE>(xw, q, shift, f); break;
	case 6: lpc_each<6, MODE>(xw, q, shift, f); break;
	case 7: lpc_each<7, MODE>(xw, q, shift, f); break;
	case 8: lpc_each<8, MODE>(xw, q, shift, f); break;
	case 9: lpc_each<9, MODE>(xw, q, shift, f); break;
	case 10: lpc_each<10, MODE>(xw, q, shift, f); break;
	case 11: lpc_each<11, MODE>(xw, q, shift, f); break;
	default: lpc_each<12, MODE>(xw, q, shift, f); break;
	}
}

// f(s, db, P): db = 2^31 + (prediction - x) mod 2^32 for chunk sample s under an order-P predictor.  |db - 2^31| taken on
// unsigned words (one absolute-difference instruction against a constant) is then exactly abs() of the reference's
// 32-bit residual x - prediction, whatever the values: an unsigned word's distance from 2^31 is the magnitude of the
// two's complement number (word - 2^31), and that number is -(x - prediction) mod 2^32, INT_MIN included.
// The sample itself enters the 64-bit sum as one more tap with the coefficient -2^shift and the sum starts from
// 2^(31 + shift), a multiple of 2^shift: ((2^(31+shift) + S - x 2^shift) >> shift) = 2^31 + (S >> shift) - x, all mod 2^64,
// and one funnel shift of the two halves yields its low word - the truncation to 32 bits of lpc.c:577-586.  Candidates
// the reference sums in 32 bits (lpc.c:296) take this path too: it gives a candidate 32 bits only where samples and
// coefficients of the stated widths cannot carry the sum out of them (stream_encoder.c:3597-3603: bps + precision +
// ilog2(order) <= 32), so the exact sum is the wrapping one.
// The window may hold the samples plus a constant b (see lpc_biased_pred_each): the caller then starts the sum from
// init = 2^(31 + shift) - b (sum of the coefficients - 2^shift), which takes the constant out of every tap.
template<int P, int W, class F>
__device__ __forceinline__ void lpc_offset_diff_each(const int32_t (&xw)[W], const int32_t (&q)[HIST], uint32_t shift, long long init, F &&f)
{
	const int32_t self = -(int32_t)(1u << shift);
#pragma unroll
	for(int s = 0; s < W - HIST; s++) {
		long long acc = init;
		acc += (long long)self * (long long)xw[HIST + s];
#pragma unroll
		for(int j = 0; j < P; j++) acc += (long long)q[j] * (long long)xw[HIST + s - 1 - j];
		f(s, __builtin_amdgcn_alignbit((uint32_t)((unsigned long long)acc >> 32), (uint32_t)acc, shift), P);
	}
}

template<int W, class F>
__device__ __forceinline__ void lpc_offset_diff_each_dispatch(uint32_t order, const int32_t (&xw)[W], const int32_t (&q)[HIST], uint32_t shift, long long init, F &&f)
{
#define RESIDUAL_CALL(P) lpc_offset_diff_each<P>(xw, q, shift, init, f)
	RESIDUAL_ORDER_SWITCH(order, RESIDUAL_CALL)
#undef RESIDUAL_CALL
}

// f(s, pb, P): pb = prediction + b for chunk sample s under an order-P predictor, from a window that holds the samples
// plus the constant b (xw[t] = x[t] + b, b a small power of two that makes every sample positive): the sum starts from
// init = (b << shift) - b (sum of the coefficients), so that the taps' share of b is taken out and b itself stays on
// the shifted result, ((S + b 2^shift) >> shift) = (S >> shift) + b.  Where |prediction| < b the words xw[HIST + s] and
// pb are both positive and |xw[HIST + s] - pb| on unsigned words (one absolute-difference instruction) is
// |x - prediction|: no tap for the sample itself.
template<int P, int W, class F>
__device__ __forceinline__ void lpc_biased_pred_each(const int32_t (&xw)[W], const int32_t (&q)[HIST], uint32_t shift, long long init, F &&f)
{
#pragma unroll
	for(int s = 0; s < W - HIST; s++) {
		long long acc = init;
#pragma unroll
		for(int j = 0; j < P; j++) acc += (long long)q[j] * (long long)xw[HIST + s - 1 - j];
		f(s, __builtin_amdgcn_alignbit((uint32_t)((unsigned long long)acc >> 32), (uint32_t)acc, shift), P);
	}
}

template<int W, class F>
__device__ __forceinline__ void lpc_biased_pred_each_dispatch(uint32_t order, const int32_t (&xw)[W], const int32_t (&q)[HIST], uint32_t shift, long long init, F &&f)
{
#define RESIDUAL_CALL(P) lpc_biased_pred_each<P>(xw, q, shift, init, f)
	RESIDUAL_ORDER_SWITCH(order, RESIDUAL_CALL)
#undef RESIDUAL_CALL
}

// k-th difference, 32-bit wrapping
template<int K, int W, class F>
__device__ __forceinline__ void fixed_each(const int32_t (&xw)[W], F &&f)
{
#pragma unroll
	for(int s = 0; s < W - HIST; s++) {
		const uint32_t a = (uint32_t)xw[HIST + s], b = (uint32_t)xw[HIST + s - 1], c = (uint32_t)xw[HIST + s - 2], d = (uint32_t)xw[HIST + s - 3], e = (uint32_t)xw[HIST + s - 4];
		uint32_t r;
		if(K == 0) r = a; else if(K == 1) r = a - b; else if(K == 2) r = a - 2u*b + c; else if(K == 3) r = a - 3u*b + 3u*c - d; else r = a - 4u*b + 6u*c - 4u*d + e;
		f(s, (int32_t)r, K);
	}
}

template<int W, class F>
__device__ __forceinline__ void fixed_each_dispatch(uint32_t order, const int32_t (&xw)[W], F &&f)
{
	switch(order) {
	case 0: fixed_each<0>(xw, f); break;
	case 1: fixed_each<1>(xw, f); break;
	case 2: fixed_each<2>(xw, f); break;
	case 3: fixed_each<3>(xw, f); break;
	default: fixed_each<4>(xw, f); break;
	}
}

// generic residual of sample i (i >= order) read from the LDS copy of the signal; q is
// any pointer to the coefficients
__device__ __forceinline__ int32_t residual_generic(const int32_t *xs, uint32_t i, bool lpc, uint32_t order, uint32_t shift, uint32_t wide, const int32_t *q)
{
	if(!lpc) {
		const uint32_t a = (uint32_t)xs[xs_idx((int32_t)i)], b = (uint32_t)xs[xs_idx((int32_t)i - 1)], c = (uint32_t)xs[xs_idx((int32_t)i - 2)],
		               d = (uint32_t)xs[xs_idx((int32_t)i - 3)], e = (uint32_t)xs[xs_idx((int32_t)i - 4)];
		return (int32_t)(order == 0 ? a : order == 1 ? a - b : order == 2 ? a - 2u*b + c : order == 3 ? a - 3u*b + 3u*c - d : a - 4u*b + 6u*c - 4u*d + e);
	}
	int32_t pred;
	if(wide) {
		long long acc = 0;
		for(uint32_t j = 0; j < order; j++) acc += (long long)q[j] * (long long)xs[xs_idx((int32_t)(i - 1u - j))];
		pred = (int32_t)(acc >> shift);
	}
	else {
		uint32_t acc = 0;
		for(uint32_t j = 0; j < order; j++) acc += (uint32_t)q[j] * (uint32_t)xs[xs_idx((int32_t)(i - 1u - j))];
		pred = (int32_t)acc >> shift;
	}
	return (int32_t)((uint32_t)xs[xs_idx((int32_t)i)] - (uint32_t)pred);
}
