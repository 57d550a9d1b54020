// Stage 1 of the encoder: apodization window multiply + autocorrelation.
//
// Replaces FLAC__lpc_window_data (lpc.c:56-61) and FLAC__lpc_compute_autocorrelation
// (lpc.c:63-110).  The reference accumulates autoc[l] += d[i] * d[i+l] in float32,
// strictly in sample order, with a separately rounded product.  Every quantity computed
// downstream (Levinson recursion, quantised coefficients, the "don't even try" test,
// the Rice parameter suggested for partition order 0) is a discontinuous function of
// those floats, so the sums are kept bit-identical here: ONE LANE PER SIGNAL owns all
// `lag` running sums of that signal and walks its samples in order with unfused
// multiply and add.  A wave therefore covers 64 signals (16 stereo frames); the chip
// is filled by the number of frames in the batch, not by splitting a sum.
//
// Each lane reads its own signal directly from the interleaved PCM (16 samples per step,
// straight into registers; the samples of one frame are shared by its 4 signals through
// L1/L2), applies the wasted-bit shift and the window (uniform index: scalar loads), and
// keeps the last `lag` windowed samples in a register shift chain.  No LDS.
#include "engine_types.h"

template<int LAG>
__global__ void __launch_bounds__(64) fa_autoc_kernel(const void *__restrict__ pcm, FaCfg c, FaWork w)
{
	const uint32_t lane = threadIdx.x;
	const uint32_t g = blockIdx.x * 64u + lane;
	const bool active = g < c.G;
	const uint32_t gg = active ? g : c.G - 1u;
	const uint32_t f = gg / c.nsig, s = gg - f * c.nsig;
	const uint32_t n = fa_frame_len(c, f);
	const uint64_t base = (uint64_t)f * c.N;
	const uint32_t wasted = w.wasted[gg];
	// lanes are in frame order, so if the first lane sits in the last frame all lanes do
	const uint32_t f0 = (blockIdx.x * 64u) / c.nsig;
	const uint32_t nloop = (f0 + 1 == c.F) ? c.last_n : c.N;

	for(uint32_t a = 0; a < c.num_apod; a++) {
		const float *__restrict__ win = w.window + (size_t)a * c.N;
		float acc[LAG], hist[LAG];
#pragma unroll
		for(int l = 0; l < LAG; l++) { acc[l] = 0.0f; hist[l] = 0.0f; }

		for(uint32_t i0 = 0; i0 < nloop; i0 += 16) {
			float d[16];
#pragma unroll
			for(int j = 0; j < 16; j++) {
				const uint32_t i = i0 + (uint32_t)j;
				float v = 0.0f;
				if(i < n) v = __fmul_rn((float)(fa_signal(pcm, c, s, base + i) >> wasted), win[i]); // lpc.c:60
				d[j] = v;
			}
#pragma unroll
			for(int j = 0; j < 16; j++) {
#pragma unroll
				for(int l = LAG - 1; l > 0; l--) hist[l] = hist[l-1];
				hist[0] = d[j];
				// autoc[l] += d[m-l] * d[m]  (same products, same order as lpc.c:100-109;
				// the zero-filled history adds +-0 for m < l, which leaves the sum unchanged)
#pragma unroll
				for(int l = 0; l < LAG; l++) acc[l] = __fadd_rn(acc[l], __fmul_rn(hist[l], d[j]));
			}
		}
		if(active) {
#pragma unroll
			for(int l = 0; l < LAG; l++)
				if((uint32_t)l < c.lag) w.autoc[((size_t)a * c.lag + l) * c.G + g] = acc[l];
		}
	}
}

extern "C" void fa_launch_autoc(const void *pcm, const FaCfg *c, const FaWork *w, hipStream_t st)
{
	const dim3 grid((c->G + 63u) / 64u), block(64);
	if(c->lag <= 9) hipLaunchKernelGGL(fa_autoc_kernel<9>, grid, block, 0, st, pcm, *c, *w);
	else if(c->lag <= 13) hipLaunchKernelGGL(fa_autoc_kernel<13>, grid, block, 0, st, pcm, *c, *w);
	else hipLaunchKernelGGL(fa_autoc_kernel<33>, grid, block, 0, st, pcm, *c, *w);
}
