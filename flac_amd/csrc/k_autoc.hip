// Stage 1 of the encoder: apodization window multiply + autocorrelation.
//
// Replaces FLAC__lpc_window_data (lpc.c:56-61) and FLAC__lpc_compute_autocorrelation
// (lpc.c:63-110).  The reference accumulates autoc[l] += d[i] * d[i+l] in float32,
// strictly in sample order, with a separately rounded product.  Every quantity computed
// downstream (Levinson recursion, quantised coefficients, the "don't even try" test,
// the Rice parameter suggested for partition order 0) is a discontinuous function of
// those floats, so the sums are kept bit-identical here: ONE LANE PER SIGNAL owns all
// `lag` running sums of that signal and walks its samples in order with unfused
// multiply and add (an MFMA or tree reduction would change the rounding).  A wave covers
// 64 signals (16 stereo frames); the chip is filled by the number of frames in the batch.
//
// Each lane streams its own signal straight from the interleaved PCM into registers,
// 16 samples per step with the next step's loads already in flight (the 4 signals of a
// frame share the same bytes through L1/L2), applies the window
// (wave-uniform index: scalar loads) and keeps the last `lag` windowed samples in a
// register delay line.  No LDS.
//
// The reference removes a signal's wasted (all-zero low) bits before this stage.  Here the
// unshifted signal is used, so that this kernel does not depend on the wasted-bit scan and can
// run beside it: every sample is the shifted one times 2^w, int-to-float conversion, products
// and sums all scale by exact powers of two (nothing overflows below 2^62), so each sum comes
// out as 4^w times the reference's, bit for bit, and the model stage divides it back out.  Cost: 2*lag + ~8 VALU instructions per sample per lane.
#include "engine_types.h"
#include "residual.h"

#define AC_STEP 16

// 16 raw wide samples of a stereo stream: PK = 1: one dword (left | right << 16) each (16-bit samples as they lie in a
// 16-bit stream), PK = 2: two dwords (left, right) each (the 32-bit layout process_interleaved hands over)
// ... and of a one-channel stream (MONO): PK = 1: eight dwords of two 16-bit samples, PK = 2: sixteen dwords
template<int NV> struct RawStereo { uint4 v[NV]; };

template<int NV>
__device__ __forceinline__ uint32_t raw_dword(const RawStereo<NV> &r, int d)
{
	const uint4 &q = r.v[d >> 2];
	return (d & 3) == 0 ? q.x : (d & 3) == 1 ? q.y : (d & 3) == 2 ? q.z : q.w;
}

// Fast path: stereo with mid/side analysis (signals L, R, M, S), or one channel (MONO: a lane per frame), blocksize a
// multiple of 16; PK as above.  A batch of one-channel frames has a quarter of the lanes (an hour: 606 waves for 1024
// SIMDs), so there the loads run two steps ahead: a wave with a SIMD to itself gets through a step in less time
// than a load takes.
template<int LAG, int PK, bool MONO = false>
__global__ void __launch_bounds__(64) fa_autoc_ms16_kernel(const uint32_t *__restrict__ pcm, FaCfg c, FaWork w)
{
	constexpr int NV = MONO ? 2 * PK : 4 * PK;          // 16-byte loads per step
	constexpr int DPS = MONO ? (PK == 1 ? 1 : 2) : 2 * PK;   // half dwords per sample: the step's AC_STEP samples take AC_STEP * DPS / 2 dwords
	// These waves are few (one lane per signal) and each is a long serial chain: they go first wherever they share a SIMD
	// with the preparation kernel's many short ones, which fill the gaps
	__builtin_amdgcn_s_setprio(3);
	const uint32_t lane = threadIdx.x;
	const uint32_t g = blockIdx.x * 64u + lane;
	const bool active = g < c.G;
	const uint32_t gg = active ? g : c.G - 1u;
	const uint32_t f = MONO ? gg : gg >> 2, s = MONO ? 0u : gg & 3u;
	const uint32_t n = fa_frame_len(c, w, f);
	const uint32_t *__restrict__ p = pcm + (size_t)f * c.N * (uint32_t)DPS / 2u;   // dwords (the blocksize is a multiple of 16)
	const uint32_t f0 = MONO ? blockIdx.x * 64u : (blockIdx.x * 64u) >> 2;
	const uint32_t nloop = (!c.multi && f0 + 1 == c.F) ? c.last_n : c.N;   // short frames in between: their lanes add zeros past their end
	// signal = (ca * left + cb * right) >> cs: L, R, (L+R)>>1, L-R (stream_encoder.c:2051-2058), wasted bits left in
	const int32_t ca = (s == 1u) ? 0 : 1, cb = (s == 0u) ? 0 : (s == 3u ? -1 : 1);
	const uint32_t sh = s == 2u ? 1u : 0u;
	const uint32_t cab = ((uint32_t)ca & 0xffffu) | ((uint32_t)cb << 16);   // the two factors as a pair of 16-bit halves

	auto load16 = [&](uint32_t i0) {
		RawStereo<NV> r;
		if(i0 + AC_STEP <= n) {
			const uint4 *q = (const uint4*)(p + (size_t)i0 * DPS / 2u);
#pragma unroll
			for(int j = 0; j < NV; j++) r.v[j] = q[j];
		}
		else { // tail of a short last block: never read past the end of the stream
			uint32_t t[4 * NV];
			if constexpr(MONO && PK == 1) {
				const uint16_t *h = (const uint16_t*)p + i0;
#pragma unroll
				for(int j = 0; j < 4 * NV; j++) t[j] = ((i0 + 2u * j < n) ? (uint32_t)h[2 * j] : 0u) | ((i0 + 2u * j + 1u < n) ? (uint32_t)h[2 * j + 1] << 16 : 0u);
			}
			else {
				constexpr int DW = DPS / 2;   // dwords per sample
#pragma unroll
				for(int j = 0; j < 4 * NV; j++) t[j] = (i0 + (uint32_t)(j / DW) < n) ? p[(size_t)i0 * DW + j] : 0u;
			}
#pragma unroll
			for(int j = 0; j < NV; j++) r.v[j] = make_uint4(t[4*j], t[4*j+1], t[4*j+2], t[4*j+3]);
		}
		return r;
	};

	// Lags are handled two at a time with the packed multiply and add instructions (two independent
	// IEEE operations each, unfused: same results as the scalar ones).  The pair (d[t], d[t-1]) that the
	// lags (k, k+1) of a sample need must sit in one aligned register pair, whatever the parity of t: the
	// windowed samples are therefore kept twice, paired (even, odd) and (odd, even), which costs one copy
	// per sample instead of re-aligning the whole delay line.
	typedef float f2 __attribute__((ext_vector_type(2)));
	constexpr int H = (LAG - 1 + 1) & ~1;      // history kept between steps, even
	constexpr int NP = LAG / 2;                // lag pairs (0,1), (2,3), ...; an odd LAG leaves lag LAG-1 on its own
	constexpr int T = H + AC_STEP;             // time slots of a step: t = H + j for sample j
	for(uint32_t a = 0; a < c.num_apod; a++) {
		const float *__restrict__ win = w.window + (size_t)a * c.N;
		f2 A[NP > 0 ? NP : 1];
		float alast = 0.0f;
		f2 E[T / 2], O[T / 2];                  // E[i] = (d[2i], d[2i+1]), O[i] = (d[2i+1], d[2i+2])
#pragma unroll
		for(int i = 0; i < NP; i++) A[i] = (f2){0.0f, 0.0f};
#pragma unroll
		for(int i = 0; i < T / 2; i++) { E[i] = (f2){0.0f, 0.0f}; O[i] = (f2){0.0f, 0.0f}; }
		RawStereo<NV> nxt = load16(0), nxt2 = nxt;
		if(MONO && AC_STEP < nloop) nxt2 = load16(AC_STEP);
		for(uint32_t i0 = 0; i0 < nloop; i0 += AC_STEP) {
			const RawStereo<NV> cur = nxt;
			if constexpr(MONO) { nxt = nxt2; if(i0 + 2 * AC_STEP < nloop) nxt2 = load16(i0 + 2 * AC_STEP); }
			else if(i0 + AC_STEP < nloop) nxt = load16(i0 + AC_STEP);
#pragma unroll
			for(int j = 0; j < AC_STEP; j++) {
				int32_t x;
				if constexpr(MONO) x = PK == 1 ? (int32_t)(raw_dword(cur, j >> 1) << (16 * (1 - (j & 1)))) >> 16 : (int32_t)raw_dword(cur, j);
				else if(PK == 1) x = dot2_i16(raw_dword(cur, j), cab, 0) >> sh;   // ca * left + cb * right straight from the packed word: one instruction
				else {
					const int32_t l = (int32_t)raw_dword(cur, 2 * j), r = (int32_t)raw_dword(cur, 2 * j + 1);
					x = (int32_t)((uint32_t)ca * (uint32_t)l + (uint32_t)cb * (uint32_t)r) >> sh;   // (32-bit wrapping, like fa_signal)
				}
				const uint32_t i = i0 + (uint32_t)j;
				// lpc.c:60; samples past the end of a short block are zero (raw is zero there and the
				// window index is clamped to stay inside the table)
				const float d = __fmul_rn((float)x, win[i]);
				constexpr int dummy = 0; (void)dummy;
				const int t = H + j;
				if(t & 1) E[t / 2].y = d; else E[t / 2].x = d;
				if((t - 1) & 1) O[(t - 1) / 2].y = d; else O[(t - 1) / 2].x = d;
				const f2 dd = (f2){d, d};
				// autoc[l] += d[m-l] * d[m]  (same products, same order as lpc.c:100-109; the
				// zero-filled history adds +-0 for m < l, which leaves the sum unchanged)
#pragma unroll
				for(int q = 0; q < NP; q++) {
					const int m = t - 2 * q;                          // d[m], d[m-1] for lags 2q, 2q+1
					const f2 pr = (m & 1) ? E[(m - 1) / 2] : O[(m - 2) / 2];   // (d[m-1], d[m])
					A[q] = A[q] + __builtin_shufflevector(pr, pr, 1, 0) * dd;
				}
				if(LAG & 1) {
					const int m = t - (LAG - 1);
					const float hv = (m & 1) ? E[m / 2].y : E[m / 2].x;
					alast = __fadd_rn(alast, __fmul_rn(hv, d));
				}
			}
			// the last H slots become the history of the next step
#pragma unroll
			for(int i = 0; i < H / 2; i++) { E[i] = E[i + AC_STEP / 2]; O[i] = O[i + AC_STEP / 2]; }
		}
		if(active) {
#pragma unroll
			for(int k = 0; k < LAG; k++) {
				const float v = (k == LAG - 1 && (LAG & 1)) ? alast : ((k & 1) ? A[k / 2].y : A[k / 2].x);
				if((uint32_t)k < c.lag) w.autoc[((size_t)a * c.lag + k) * c.G + g] = v;
			}
		}
	}
}

// Generic path: any channel count / sample format / blocksize.
template<int LAG>
__global__ void __launch_bounds__(64) fa_autoc_kernel(const void *__restrict__ pcm, FaCfg c, FaWork w)
{
	const uint32_t lane = threadIdx.x;
	const uint32_t g = blockIdx.x * 64u + lane;
	const bool active = g < c.G;
	const uint32_t gg = active ? g : c.G - 1u;
	const uint32_t f = gg / c.nsig, s = gg - f * c.nsig;
	const uint32_t n = fa_frame_len(c, w, f);
	const uint64_t base = (uint64_t)f * c.N;
	// lanes are in frame order, so if the first lane sits in the last frame all lanes do
	const uint32_t f0 = (blockIdx.x * 64u) / c.nsig;
	const uint32_t nloop = (!c.multi && f0 + 1 == c.F) ? c.last_n : c.N;   // short frames in between: their lanes add zeros past their end
	const uint32_t ca = (s == 1u) ? 0u : 1u, cb = (s == 0u) ? 0u : (s == 3u ? 0xffffffffu : 1u), sh = (s == 2u) ? 1u : 0u;

	for(uint32_t a = 0; a < c.num_apod; a++) {
		const float *__restrict__ win = w.window + (size_t)a * c.N;
		float acc[LAG], hist[LAG];
#pragma unroll
		for(int l = 0; l < LAG; l++) { acc[l] = 0.0f; hist[l] = 0.0f; }
		// GS samples per step, the next step's loads in flight while this one's sums are taken: a lane has nothing
		// else to cover the latency of its loads with (every lane reads its own signal, few waves share a SIMD)
		constexpr int GS = 16;
		// no branch around a load: past the end the last sample is fetched again and replaced by zero
		auto ld = [&](uint32_t i) {
			const uint64_t wide = base + (i < n ? i : n - 1u);
			int32_t v;
			if(c.midside) {   // L, R, (L+R)>>1, L-R (stream_encoder.c:2051-2058) as (ca*l + cb*r) >> sh, 32-bit wrapping like fa_signal
				int32_t l, r;
				if(c.pcm_format == FLACAMD_PCM_S16) { const uint32_t lr = ((const uint32_t*)pcm)[wide]; l = (int16_t)(lr & 0xffffu); r = (int32_t)lr >> 16; }
				else { const int2 lr = ((const int2*)pcm)[wide]; l = lr.x; r = lr.y; }
				v = (int32_t)(ca * (uint32_t)l + cb * (uint32_t)r) >> sh;
			}
			else v = fa_pcm(pcm, c.pcm_format, c.channels, wide, s);
			return i < n ? v : 0;
		};
		int32_t nxt[GS];
#pragma unroll
		for(int j = 0; j < GS; j++) nxt[j] = ld((uint32_t)j);
		for(uint32_t i0 = 0; i0 < nloop; i0 += GS) {
			int32_t cur[GS];
#pragma unroll
			for(int j = 0; j < GS; j++) cur[j] = nxt[j];
#pragma unroll
			for(int j = 0; j < GS; j++) nxt[j] = ld(i0 + (uint32_t)GS + (uint32_t)j);
#pragma unroll
			for(int j = 0; j < GS; j++) {
				const uint32_t i = i0 + (uint32_t)j;
				// lpc.c:60; past the end of the block the sample is zero and so is every product it enters
				const float d = __fmul_rn((float)cur[j], win[i < c.N ? i : c.N - 1u]);
#pragma unroll
				for(int k = LAG - 1; k > 0; k--) hist[k] = hist[k-1];
				hist[0] = d;
#pragma unroll
				for(int k = 0; k < LAG; k++) acc[k] = __fadd_rn(acc[k], __fmul_rn(hist[k], d));
			}
		}
		if(active) {
#pragma unroll
			for(int k = 0; k < LAG; k++)
				if((uint32_t)k < c.lag) w.autoc[((size_t)a * c.lag + k) * c.G + g] = acc[k];
		}
	}
}

extern "C" void fa_launch_autoc(const void *pcm, const FaCfg *c, const FaWork *w, hipStream_t st)
{
	const dim3 grid((c->G + 63u) / 64u), block(64);
	const bool ms = c->midside && c->channels == 2u && (c->N % AC_STEP) == 0 && ((uintptr_t)pcm % 16u) == 0;
	if(ms && c->pcm_format == FLACAMD_PCM_S16) {
		if(c->lag <= 9) hipLaunchKernelGGL((fa_autoc_ms16_kernel<9, 1>), grid, block, 0, st, (const uint32_t*)pcm, *c, *w);
		else if(c->lag <= 13) hipLaunchKernelGGL((fa_autoc_ms16_kernel<13, 1>), grid, block, 0, st, (const uint32_t*)pcm, *c, *w);
		else hipLaunchKernelGGL((fa_autoc_ms16_kernel<33, 1>), grid, block, 0, st, (const uint32_t*)pcm, *c, *w);
	}
	else if(ms && c->pcm_format == FLACAMD_PCM_S32 && c->lag <= 13) {   // (lag 33 with 32 raw dwords in flight twice over would not fit the register file)
		if(c->lag <= 9) hipLaunchKernelGGL((fa_autoc_ms16_kernel<9, 2>), grid, block, 0, st, (const uint32_t*)pcm, *c, *w);
		else hipLaunchKernelGGL((fa_autoc_ms16_kernel<13, 2>), grid, block, 0, st, (const uint32_t*)pcm, *c, *w);
	}
	else if(c->channels == 1u && c->nsig == 1u && (c->N % AC_STEP) == 0 && ((uintptr_t)pcm % 16u) == 0 && c->lag <= 13u
		&& (c->pcm_format == FLACAMD_PCM_S16 || c->pcm_format == FLACAMD_PCM_S32) && !(c->debug & 0x10000000u)) {
		if(c->pcm_format == FLACAMD_PCM_S16) {
			if(c->lag <= 9) hipLaunchKernelGGL((fa_autoc_ms16_kernel<9, 1, true>), grid, block, 0, st, (const uint32_t*)pcm, *c, *w);
			else hipLaunchKernelGGL((fa_autoc_ms16_kernel<13, 1, true>), grid, block, 0, st, (const uint32_t*)pcm, *c, *w);
		}
		else {
			if(c->lag <= 9) hipLaunchKernelGGL((fa_autoc_ms16_kernel<9, 2, true>), grid, block, 0, st, (const uint32_t*)pcm, *c, *w);
			else hipLaunchKernelGGL((fa_autoc_ms16_kernel<13, 2, true>), grid, block, 0, st, (const uint32_t*)pcm, *c, *w);
		}
	}
	else {
		if(c->lag <= 9) hipLaunchKernelGGL(fa_autoc_kernel<9>, grid, block, 0, st, pcm, *c, *w);
		else if(c->lag <= 13) hipLaunchKernelGGL(fa_autoc_kernel<13>, grid, block, 0, st, pcm, *c, *w);
		else hipLaunchKernelGGL(fa_autoc_kernel<33>, grid, block, 0, st, pcm, *c, *w);
	}
}
