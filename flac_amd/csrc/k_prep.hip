// Stage 0 of the encoder: per (frame, signal) wasted bits, constant detection and the
// fixed-predictor error sums.
//
// Replaces get_wasted_bits_ (stream_encoder.c:4113-4135), the constant check of
// process_subframe_ (:3215-3232) and the accumulation loop of
// FLAC__fixed_compute_best_predictor{,_wide} (fixed.c:229-243, :287-305).
//
// One 256-thread workgroup per signal.  The signal (left, right, mid or side, or channel c)
// is built straight from the interleaved PCM into LDS; the k-th differences are then read
// from LDS with unit stride.  HBM traffic: the frame's PCM once per signal (L2 absorbs the
// repeats), a few dozen bytes out.
#include "engine_types.h"

__device__ __forceinline__ uint32_t wave_or(uint32_t v)
{
#pragma unroll
	for(int o = 32; o > 0; o >>= 1) v |= (uint32_t)__shfl_xor((int)v, o, 64);
	return v;
}
__device__ __forceinline__ unsigned long long wave_sum64(unsigned long long v)
{
#pragma unroll
	for(int o = 32; o > 0; o >>= 1) v += (unsigned long long)__shfl_xor((long long)v, o, 64);
	return v;
}

__global__ void __launch_bounds__(256) fa_prep_kernel(const void *__restrict__ pcm, FaCfg c, FaWork w)
{
	extern __shared__ int32_t xs[];
	__shared__ uint32_t s_or, s_ne;
	__shared__ unsigned long long s_err[5];

	const uint32_t g = blockIdx.x, f = g / c.nsig, s = g - f * c.nsig;
	const uint32_t n = fa_frame_len(c, f);
	const uint64_t base = (uint64_t)f * c.N;
	const uint32_t tid = threadIdx.x;

	if(tid == 0) { s_or = 0; s_ne = 0; }
	if(tid < 5) s_err[tid] = 0;
	__syncthreads();

	const int32_t x0 = fa_signal(pcm, c, s, base);
	uint32_t acc = 0, ne = 0;
	for(uint32_t i = tid; i < n; i += 256) {
		const int32_t v = fa_signal(pcm, c, s, base + i);
		xs[i] = v;
		acc |= (uint32_t)v;
		ne |= (v != x0) ? 1u : 0u;
	}
	acc = wave_or(acc); ne = wave_or(ne);
	if((tid & 63) == 0) { atomicOr(&s_or, acc); atomicOr(&s_ne, ne); }
	__syncthreads();

	const uint32_t all = s_or;
	const uint32_t wasted = all ? (uint32_t)__ffs((int)all) - 1u : 0u;

	// sums of |k-th difference| over samples [4, n), differences wrapping in 32 bits
	unsigned long long e0 = 0, e1 = 0, e2 = 0, e3 = 0, e4 = 0;
	for(uint32_t i = tid + FA_MAX_FIXED_ORDER; i < n; i += 256) {
		const uint32_t a = (uint32_t)(xs[i] >> wasted), b = (uint32_t)(xs[i-1] >> wasted), cc = (uint32_t)(xs[i-2] >> wasted),
		               d = (uint32_t)(xs[i-3] >> wasted), ee = (uint32_t)(xs[i-4] >> wasted);
		const int32_t d0 = (int32_t)a, d1 = (int32_t)(a - b), d2 = (int32_t)(a - 2u*b + cc),
		              d3 = (int32_t)(a - 3u*b + 3u*cc - d), d4 = (int32_t)(a - 4u*b + 6u*cc - 4u*d + ee);
		e0 += (uint32_t)(d0 < 0 ? 0u - (uint32_t)d0 : (uint32_t)d0);
		e1 += (uint32_t)(d1 < 0 ? 0u - (uint32_t)d1 : (uint32_t)d1);
		e2 += (uint32_t)(d2 < 0 ? 0u - (uint32_t)d2 : (uint32_t)d2);
		e3 += (uint32_t)(d3 < 0 ? 0u - (uint32_t)d3 : (uint32_t)d3);
		e4 += (uint32_t)(d4 < 0 ? 0u - (uint32_t)d4 : (uint32_t)d4);
	}
	e0 = wave_sum64(e0); e1 = wave_sum64(e1); e2 = wave_sum64(e2); e3 = wave_sum64(e3); e4 = wave_sum64(e4);
	if((tid & 63) == 0) {
		atomicAdd(&s_err[0], e0); atomicAdd(&s_err[1], e1); atomicAdd(&s_err[2], e2); atomicAdd(&s_err[3], e3); atomicAdd(&s_err[4], e4);
	}
	__syncthreads();
	if(tid == 0) {
		w.wasted[g] = (uint8_t)wasted;
		w.is_const[g] = (uint8_t)(s_ne ? 0 : 1);
	}
	if(tid < 5) {
		unsigned long long v = s_err[tid];
		if(!c.wide_fixed) v &= 0xffffffffull;  // fixed.c:234: 32-bit accumulators wrap
		w.fixed_err[(size_t)tid * c.G + g] = v;
	}
}

extern "C" void fa_launch_prep(const void *pcm, const FaCfg *c, const FaWork *w, hipStream_t st)
{
	const size_t lds = sizeof(int32_t) * (size_t)c->N;
	hipLaunchKernelGGL(fa_prep_kernel, dim3(c->G), dim3(256), lds, st, pcm, *c, *w);
}
