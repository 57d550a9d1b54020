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
#include "wave_ops.h"
#include "residual.h"

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

// SPILL: blocks too long for an LDS copy of the signal; the differences re-read the PCM instead (cache hits)
// which: 0 every frame of the batch; 1 the batch's last frame alone; 2 the short frames listed in FaWork::short_list
// (the full-length frames then have the kernel further down that reads a frame once for all its channels)
template<bool SPILL>
__global__ void __launch_bounds__(256) fa_prep_kernel(const void *__restrict__ pcm, FaCfg c, FaWork w, uint32_t which)
{
	extern __shared__ __align__(16) int32_t xs[];
	__shared__ uint32_t s_or, s_ne;
	__shared__ unsigned long long s_err[5];

	uint32_t g;
	if(which == 0u) g = fa_xcd_signal(blockIdx.x, gridDim.x, c.nsig);   // the signals of a frame share an XCD
	else if(which == 1u) g = (c.F - 1u) * c.nsig + blockIdx.x;
	else g = w.short_list[blockIdx.x / c.nsig] * c.nsig + blockIdx.x % c.nsig;
	const uint32_t f = g / c.nsig, s = g - f * c.nsig;
	const uint32_t n = fa_frame_len(c, w, f);
	if(which == 2u && n == c.N) return;
	const uint64_t base = (uint64_t)f * c.N;
	const uint32_t tid = threadIdx.x;

	if(tid == 0) { s_or = 0; s_ne = 0; }
	if(tid < 5) s_err[tid] = 0;
	__syncthreads();

	const int32_t x0 = fa_signal(pcm, c, s, base);
	uint32_t acc = 0, ne = 0;
	// one channel: the signal lies as it is, so a thread takes 16 bytes at a time (a block is two rounds of loads then,
	// not sixteen that each wait for memory)
	const bool lies_flat = !SPILL && c.channels == 1u && !c.midside && (c.N % 8u) == 0 && ((uintptr_t)pcm % 16u) == 0
		&& (c.pcm_format == FLACAMD_PCM_S16 || c.pcm_format == FLACAMD_PCM_S32);
	if(lies_flat) {
		if(c.pcm_format == FLACAMD_PCM_S16) {
			const int16_t *src = (const int16_t*)pcm + base;
			for(uint32_t i = tid * 8u; i < n; i += 2048u) {
				int32_t v[8];
				if(i + 8u <= n) {
					const uint4 q = *(const uint4*)(src + i);
					const uint32_t d[4] = { q.x, q.y, q.z, q.w };
#pragma unroll
					for(int j = 0; j < 4; j++) { v[2 * j] = (int32_t)(int16_t)(d[j] & 0xffffu); v[2 * j + 1] = (int32_t)d[j] >> 16; }
					*(int4*)(xs + i) = make_int4(v[0], v[1], v[2], v[3]); *(int4*)(xs + i + 4u) = make_int4(v[4], v[5], v[6], v[7]);
				}
				else {
#pragma unroll
					for(int j = 0; j < 8; j++) { v[j] = (i + (uint32_t)j < n) ? (int32_t)src[i + (uint32_t)j] : x0; if(i + (uint32_t)j < n) xs[i + (uint32_t)j] = v[j]; }
				}
#pragma unroll
				for(int j = 0; j < 8; j++) { acc |= (i + (uint32_t)j < n) ? (uint32_t)v[j] : 0u; ne |= (v[j] != x0) ? 1u : 0u; }
			}
		}
		else {
			const int32_t *src = (const int32_t*)pcm + base;
			for(uint32_t i = tid * 4u; i < n; i += 1024u) {
				int32_t v[4];
				if(i + 4u <= n) { const int4 q = *(const int4*)(src + i); v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; *(int4*)(xs + i) = q; }
				else {
#pragma unroll
					for(int j = 0; j < 4; j++) { v[j] = (i + (uint32_t)j < n) ? src[i + (uint32_t)j] : x0; if(i + (uint32_t)j < n) xs[i + (uint32_t)j] = v[j]; }
				}
#pragma unroll
				for(int j = 0; j < 4; j++) { acc |= (i + (uint32_t)j < n) ? (uint32_t)v[j] : 0u; ne |= (v[j] != x0) ? 1u : 0u; }
			}
		}
	}
	else
	for(uint32_t i = tid; i < n; i += 256) {
		const int32_t v = fa_signal(pcm, c, s, base + i);
		if constexpr(!SPILL) xs[i] = v;
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
	auto X = [&](uint32_t j) -> int32_t { if constexpr(SPILL) return fa_signal(pcm, c, s, base + j); else return xs[j]; };
	auto one_sample = [&](uint32_t i) {
		const uint32_t a = (uint32_t)(X(i) >> wasted), b = (uint32_t)(X(i-1) >> wasted), cc = (uint32_t)(X(i-2) >> wasted),
		               d = (uint32_t)(X(i-3) >> wasted), ee = (uint32_t)(X(i-4) >> wasted);
		const int32_t d0 = (int32_t)a, d1 = (int32_t)(a - b), d2 = (int32_t)(a - 2u*b + cc),
		              d3 = (int32_t)(a - 3u*b + 3u*cc - d), d4 = (int32_t)(a - 4u*b + 6u*cc - 4u*d + ee);
		e0 += (uint32_t)(d0 < 0 ? 0u - (uint32_t)d0 : (uint32_t)d0);
		e1 += (uint32_t)(d1 < 0 ? 0u - (uint32_t)d1 : (uint32_t)d1);
		e2 += (uint32_t)(d2 < 0 ? 0u - (uint32_t)d2 : (uint32_t)d2);
		e3 += (uint32_t)(d3 < 0 ? 0u - (uint32_t)d3 : (uint32_t)d3);
		e4 += (uint32_t)(d4 < 0 ? 0u - (uint32_t)d4 : (uint32_t)d4);
	};
	// Samples of at most 25 bits (fourth differences below 2^29, nothing wraps): a thread takes runs of eight consecutive
	// samples, each difference order from the one below (d_k(i) = d_(k-1)(i) - d_(k-1)(i-1)) and |d_k| added up by the
	// |a - b| + c instruction on the two order-(k-1) values moved into unsigned order; eight of them fit 32 bits.
	// Run 0 starts at sample 4; the samples after the last whole run go one by one.
	const uint32_t bps_max = c.bps + (c.midside ? 1u : 0u);
	uint32_t done = FA_MAX_FIXED_ORDER;
	if(!SPILL && bps_max <= 25u && n >= 8u) {
		const uint32_t runs = n / 8u, B = 0x80000000u;
		for(uint32_t r = tid; r < runs; r += 256) {
			const uint32_t i0 = r * 8u;
			uint32_t v[12];
			{
				// three 16-byte LDS reads (the run's start less four samples is a multiple of four words); run 0 has no history
				const int4 *p4 = (const int4*)(xs + i0);
				const int4 h = r ? p4[-1] : make_int4(0, 0, 0, 0), lo = p4[0], hi = p4[1];
				v[0] = (uint32_t)(h.x >> wasted); v[1] = (uint32_t)(h.y >> wasted); v[2] = (uint32_t)(h.z >> wasted); v[3] = (uint32_t)(h.w >> wasted);
				v[4] = (uint32_t)(lo.x >> wasted); v[5] = (uint32_t)(lo.y >> wasted); v[6] = (uint32_t)(lo.z >> wasted); v[7] = (uint32_t)(lo.w >> wasted);
				v[8] = (uint32_t)(hi.x >> wasted); v[9] = (uint32_t)(hi.y >> wasted); v[10] = (uint32_t)(hi.z >> wasted); v[11] = (uint32_t)(hi.w >> wasted);
			}
			// state before the run's first sample: x, d1, d2, d3 at i0 - 1
			uint32_t px = v[3], pd1 = v[3] - v[2], pd2 = pd1 - (v[2] - v[1]), pd3 = pd2 - ((v[2] - v[1]) - (v[1] - v[0]));
			uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0;
#pragma unroll
			for(int j = 0; j < 8; j++) {
				const uint32_t a = v[4 + j], d1 = a - px, d2 = d1 - pd1, d3 = d2 - pd2;
				if(r != 0u || j >= 4) {
					a0 = sad_u32(a ^ B, B, a0); a1 = sad_u32(a ^ B, px ^ B, a1); a2 = sad_u32(d1 ^ B, pd1 ^ B, a2);
					a3 = sad_u32(d2 ^ B, pd2 ^ B, a3); a4 = sad_u32(d3 ^ B, pd3 ^ B, a4);
				}
				px = a; pd1 = d1; pd2 = d2; pd3 = d3;
			}
			e0 += a0; e1 += a1; e2 += a2; e3 += a3; e4 += a4;
		}
		done = runs * 8u;
	}
	for(uint32_t i = tid + done; i < n; i += 256) one_sample(i);
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

// ---- independent channels of 32-bit words: one workgroup per FRAME, every channel from one read ------
//
// The kernel above fetches an interleaved frame once per channel and uses one word in C of every cache line it pulls
// through L1; for eight channels that traffic, not the arithmetic, is what it takes its time over.  Here a workgroup
// takes the frame in pieces of PL_CHUNK samples: the piece is read as it lies (16 bytes per thread and load, every line
// once) and put into LDS channel by channel, then wave v goes through channels v, v + 4, ... with one run of eight
// samples per lane, by the same successive differences as above.  A channel belongs to one wave from the first piece
// to the last, so its sums, the OR of its samples and "differs from the first sample" need one reduction over the
// wave at the end and nothing shared.  Wasted bits are taken out of the sums afterwards (all samples of a channel
// share their trailing zeros, so every difference does; samples of at most 25 bits: nothing wraps).
// MS: two channels analysed as left, right, mid and side (stream_encoder.c:2051-2058): the two rows are staged the
// same way and wave v forms signal v from them as it reads.
// Full-length frames only; blocksize a multiple of PL_CHUNK.
#define PL_CHUNK 512
#define PL_ROW (PL_CHUNK + 4)     // four samples of history in front of a channel's piece
template<int C, bool MS = false>
__global__ void __launch_bounds__(256) fa_prep_planar_kernel(const int32_t *__restrict__ pcm, FaCfg c, FaWork w)
{
	static_assert(!MS || C == 2, "mid and side are formed from two channels");
	constexpr int NSIG = MS ? 4 : C;
	constexpr int SLOTS = (NSIG + 3) / 4;   // signals per wave
	__shared__ __align__(16) int32_t xs[C][PL_ROW];
	__shared__ __align__(16) int32_t s_hist[NSIG][4];   // a signal's last four samples of the piece before (written and read by the signal's own wave)
	const uint32_t f = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
	if(fa_frame_len(c, w, f) != c.N) return;     // short frames go to the kernel above
	const int32_t *__restrict__ src = pcm + (size_t)f * c.N * C;
	unsigned long long e[SLOTS][5];
	uint32_t orv[SLOTS], ne[SLOTS]; int32_t first[SLOTS];
#pragma unroll
	for(int q = 0; q < SLOTS; q++) { orv[q] = 0; ne[q] = 0; first[q] = 0; for(int k = 0; k < 5; k++) e[q][k] = 0; }
	const uint32_t B = 0x80000000u;
	for(uint32_t i0 = 0; i0 < c.N; i0 += PL_CHUNK) {
		// the piece as it lies: PL_CHUNK * C words, four to a load
		const uint4 *p4 = (const uint4*)(src + (size_t)i0 * C);
#pragma unroll
		for(int k = 0; k < C / 2 + (C & 1); k++) {
			const uint32_t u = tid + 256u * (uint32_t)k;
			if((C & 1) == 0 || u < (uint32_t)(PL_CHUNK * C / 4)) {
				const uint4 v = p4[u];
				const uint32_t flat = 4u * u, smp = flat / (uint32_t)C, ch = flat - smp * (uint32_t)C;
				const uint32_t vv[4] = { v.x, v.y, v.z, v.w };
				uint32_t sm = smp, cc = ch;
#pragma unroll
				for(int j = 0; j < 4; j++) { xs[cc][4u + sm] = (int32_t)vv[j]; if(++cc == (uint32_t)C) { cc = 0; sm++; } }
			}
		}
		__syncthreads();
#pragma unroll
		for(int q = 0; q < SLOTS; q++) {
			const uint32_t ch = wv + 4u * (uint32_t)q;
			if(ch < (uint32_t)NSIG) {
				// samples 8 lane - 4 .. 8 lane + 7 of the piece; the first lane's four of history are its wave's own keeping
				int4 h, lo, hi;
				if constexpr(MS) {
					const int4 *ra = (const int4*)(&xs[0][0] + 8u * lane), *rb = (const int4*)(&xs[1][0] + 8u * lane);
					const int4 ha = ra[0], la = ra[1], ia = ra[2], hb = rb[0], lb = rb[1], ib = rb[2];
					auto form = [&](int32_t l, int32_t r) -> int32_t {   // 32-bit wrapping, like fa_signal
						return ch == 0u ? l : ch == 1u ? r : ch == 2u ? (int32_t)((uint32_t)l + (uint32_t)r) >> 1 : (int32_t)((uint32_t)l - (uint32_t)r);
					};
					h = make_int4(form(ha.x, hb.x), form(ha.y, hb.y), form(ha.z, hb.z), form(ha.w, hb.w));
					lo = make_int4(form(la.x, lb.x), form(la.y, lb.y), form(la.z, lb.z), form(la.w, lb.w));
					hi = make_int4(form(ia.x, ib.x), form(ia.y, ib.y), form(ia.z, ib.z), form(ia.w, ib.w));
				}
				else {
					const int4 *r4 = (const int4*)(&xs[ch][0] + 8u * lane);
					h = r4[0]; lo = r4[1]; hi = r4[2];
				}
				if(lane == 0u && i0 != 0u) h = *(const int4*)&s_hist[ch][0];
				if(lane == 63u) *(int4*)&s_hist[ch][0] = hi;   // (after the first lane's read: the same wave, in program order)
				const uint32_t v[12] = { (uint32_t)h.x, (uint32_t)h.y, (uint32_t)h.z, (uint32_t)h.w, (uint32_t)lo.x, (uint32_t)lo.y, (uint32_t)lo.z, (uint32_t)lo.w,
				                         (uint32_t)hi.x, (uint32_t)hi.y, (uint32_t)hi.z, (uint32_t)hi.w };
				if(i0 == 0u) first[q] = (int32_t)__builtin_amdgcn_readfirstlane((int)lo.x);   // the signal's first sample
				const bool head = i0 == 0u && lane == 0u;   // the frame's first four samples: no history, and no part in the sums (fixed.c:229-243 starts at sample 4)
				uint32_t px = v[3], pd1 = v[3] - v[2], pd2 = pd1 - (v[2] - v[1]), pd3 = pd2 - ((v[2] - v[1]) - (v[1] - v[0]));
				uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, o = 0, d = 0;
#pragma unroll
				for(int j = 0; j < 8; j++) {
					const uint32_t a = v[4 + j], d1 = a - px, d2 = d1 - pd1, d3 = d2 - pd2;
					o |= a; d |= a ^ (uint32_t)first[q];
					if(!head || j >= 4) {
						a0 = sad_u32(a ^ B, B, a0); a1 = sad_u32(a ^ B, px ^ B, a1); a2 = sad_u32(d1 ^ B, pd1 ^ B, a2);
						a3 = sad_u32(d2 ^ B, pd2 ^ B, a3); a4 = sad_u32(d3 ^ B, pd3 ^ B, a4);
					}
					px = a; pd1 = d1; pd2 = d2; pd3 = d3;
				}
				e[q][0] += a0; e[q][1] += a1; e[q][2] += a2; e[q][3] += a3; e[q][4] += a4;
				orv[q] |= o; ne[q] |= d;
			}
		}
		__syncthreads();
	}
#pragma unroll
	for(int q = 0; q < SLOTS; q++) {
		const uint32_t ch = wv + 4u * (uint32_t)q;
		if(ch < (uint32_t)NSIG) {
			const uint32_t all = wave_or(orv[q]), differs = wave_or(ne[q]);
			const uint32_t wasted = all ? (uint32_t)__ffs((int)all) - 1u : 0u;
			const uint32_t g = f * c.nsig + ch;
#pragma unroll
			for(int k = 0; k < 5; k++) {
				unsigned long long v = wave_sum64(e[q][k]) >> wasted;
				if(!c.wide_fixed) v &= 0xffffffffull;  // fixed.c:234: 32-bit accumulators wrap
				if(lane == 0) w.fixed_err[(size_t)k * c.G + g] = v;
			}
			if(lane == 0) { w.wasted[g] = (uint8_t)wasted; w.is_const[g] = (uint8_t)(differs ? 0 : 1); }
		}
	}
}

// ---- 16-bit stereo: one workgroup per FRAME, every signal of the frame from one read ------
//
// Each thread loads 16 wide samples (+4 of history) as packed dwords, forms left, right, mid
// and side in registers and accumulates, per signal, the OR of the samples, "differs from
// the first sample" and the five |k-th difference| sums.  Wasted bits are removed AFTER
// summing: all samples of a signal share their trailing zeros, so the differences of the
// shifted signal are the differences of the unshifted one shifted by the same amount
// (nothing wraps below 28 bits), and sum >> wasted is exact.  The per-lane partials are
// combined with a halving butterfly (each exchange step halves the number of values a lane
// still carries: 31 shuffles for 20 sums instead of 120).
#define PP_CH 16
#define PP_HIST 4

struct PrepFastShared { unsigned long long err[4][5]; uint32_t orv[4], ne[4]; };

__global__ void __launch_bounds__(256) fa_prep_stereo16_kernel(const uint32_t *__restrict__ pcm, FaCfg c, FaWork w, uint32_t frame0, uint32_t only_short)
{
	__shared__ PrepFastShared sh;
	__shared__ uint32_t s_fps[4][5][64];   // per signal and difference order: sums over the finest partitions (FaCfg::fixed_psum)
	// only_short = 2: the workgroups cover the short frames listed in FaWork::short_list (a batch of several streams)
	const uint32_t f = only_short == 2u ? w.short_list[blockIdx.x] : frame0 + blockIdx.x, tid = threadIdx.x, lane = tid & 63u;
	const uint32_t n = fa_frame_len(c, w, f);
	if(only_short && n == c.N) return;     // full-length frames went to the kernel below
	if(only_short) __builtin_amdgcn_s_setprio(3);   // a few workgroups beside tens of thousands of the other kernel's: do not queue behind them for every instruction
	if(c.fixed_psum) for(uint32_t t = tid; t < 4u * 5u * 64u; t += 256u) (&s_fps[0][0][0])[t] = 0u;
	const uint32_t *__restrict__ p = pcm + (size_t)f * c.N;
	const uint32_t i0 = tid * PP_CH;
	if(tid < 20u) sh.err[tid / 5u][tid % 5u] = 0;
	if(tid < 4u) { sh.orv[tid] = 0; sh.ne[tid] = 0; }

	uint32_t raw[PP_HIST + PP_CH];
#pragma unroll
	for(int j = 0; j < (PP_HIST + PP_CH) / 4; j++) {
		const int32_t rel = (int32_t)i0 - PP_HIST + 4 * j;
		uint4 v = make_uint4(0u, 0u, 0u, 0u);
		if(rel >= 0 && (uint32_t)rel + 4u <= n) v = *(const uint4*)(p + rel);
		else if(rel >= 0 && (uint32_t)rel < n) { // tail of a short last block: never read past the end of the stream
			v.x = p[rel];
			if((uint32_t)rel + 1u < n) v.y = p[rel + 1];
			if((uint32_t)rel + 2u < n) v.z = p[rel + 2];
		}
		raw[4*j] = v.x; raw[4*j+1] = v.y; raw[4*j+2] = v.z; raw[4*j+3] = v.w;
	}
	const uint32_t first = p[0];
	__syncthreads();

	uint32_t sums[4][5], orv[4], ne[4];
	// Full blocks whose threads all hold 16 samples: every difference order comes from the one below
	// (d_k(i) = d_(k-1)(i) - d_(k-1)(i-1)) and |d_k| is added up by the |a - b| + c instruction on the two
	// order-(k-1) values, biased into unsigned order (differences of 17-bit samples are far from wrapping).
	// The first four samples of a frame do not count (fixed.c:229-243): the frame's first thread takes its
	// share back out afterwards.  Anything else (a short last block) goes sample by sample below.
	const bool whole = n == c.N && (n % PP_CH) == 0;
#pragma unroll
	for(int s = 0; s < 4; s++) {
		const int32_t ca = (s == 1) ? 0 : 1, cb = (s == 0) ? 0 : (s == 3 ? -1 : 1);
		const int sh1 = (s == 2) ? 1 : 0;
		int32_t x[PP_HIST + PP_CH];
#pragma unroll
		for(int j = 0; j < PP_HIST + PP_CH; j++) {
			const int32_t l = (int32_t)(int16_t)(raw[j] & 0xffffu), r = (int32_t)raw[j] >> 16;
			x[j] = (ca * l + cb * r) >> sh1;
		}
		const int32_t x0 = (ca * (int32_t)(int16_t)(first & 0xffffu) + cb * ((int32_t)first >> 16)) >> sh1;
		uint32_t o = 0, d = 0, e0 = 0, e1 = 0, e2 = 0, e3 = 0, e4 = 0;
		if(whole) {
			if(i0 < n) {
				const uint32_t B = 0x80000000u;
				uint32_t diff = 0, mag = 0;
				uint32_t pxb = (uint32_t)x[3] ^ B, pd1, pd2, pd3;
				{
					const uint32_t h1 = (uint32_t)x[3], h2 = (uint32_t)x[2], h3 = (uint32_t)x[1], h4 = (uint32_t)x[0];
					const uint32_t a1 = h1 - h2, a2 = h2 - h3, a3 = h3 - h4;   // d1(-1), d1(-2), d1(-3)
					const uint32_t b1 = a1 - a2, b2 = a2 - a3;                  // d2(-1), d2(-2)
					pd1 = a1 ^ B; pd2 = b1 ^ B; pd3 = (b1 - b2) ^ B;
				}
#pragma unroll
				for(int q = 0; q < PP_CH; q++) {
					const uint32_t a = (uint32_t)x[PP_HIST + q], ab = a ^ B;
					const uint32_t d1b = (ab - pxb) ^ B, d2b = (d1b - pd1) ^ B, d3b = (d2b - pd2) ^ B;   // the biases cancel in each difference
					e0 = sad_u32(ab, B, e0); e1 = sad_u32(ab, pxb, e1); e2 = sad_u32(d1b, pd1, e2); e3 = sad_u32(d2b, pd2, e3); e4 = sad_u32(d3b, pd3, e4);
					o |= a; diff |= a ^ (uint32_t)x0;
					if(s == 3) mag |= a ^ (uint32_t)((int32_t)a >> 31);
					pxb = ab; pd1 = d1b; pd2 = d2b; pd3 = d3b;
				}
				// bit 0: differs from the first sample somewhere; bit 1 (side signal of 16-bit channels, nominally 17 bits): a sample outside the 16-bit range
				d = (diff ? 1u : 0u) | ((s == 3 && (mag >> 15)) ? 2u : 0u);
				uint32_t r0 = e0, r1 = e1, r2 = e2, r3 = e3, r4 = e4;   // the same sums over the samples that have a residual: i >= k for order k
				if(tid == 0) {
					// samples 0..3 of the frame: history is zero, so d_k(q) is the k-th difference of (0, 0, 0, 0, x0, x1, x2, x3)
					const uint32_t s0 = (uint32_t)x[4], s1 = (uint32_t)x[5], s2 = (uint32_t)x[6], s3 = (uint32_t)x[7];
					const uint32_t f1[4] = { s0, s1 - s0, s2 - s1, s3 - s2 };
					const uint32_t f2[4] = { f1[0], f1[1] - f1[0], f1[2] - f1[1], f1[3] - f1[2] };
					const uint32_t f3[4] = { f2[0], f2[1] - f2[0], f2[2] - f2[1], f2[3] - f2[2] };
					const uint32_t f4[4] = { f3[0], f3[1] - f3[0], f3[2] - f3[1], f3[3] - f3[2] };
					auto mg = [](uint32_t v) { const int32_t t = (int32_t)v; return (uint32_t)(t < 0 ? -t : t); };
					e0 -= mg(s0) + mg(s1) + mg(s2) + mg(s3);
					e1 -= mg(f1[0]) + mg(f1[1]) + mg(f1[2]) + mg(f1[3]);
					e2 -= mg(f2[0]) + mg(f2[1]) + mg(f2[2]) + mg(f2[3]);
					e3 -= mg(f3[0]) + mg(f3[1]) + mg(f3[2]) + mg(f3[3]);
					e4 -= mg(f4[0]) + mg(f4[1]) + mg(f4[2]) + mg(f4[3]);
					r1 -= mg(f1[0]); r2 -= mg(f2[0]) + mg(f2[1]); r3 -= mg(f3[0]) + mg(f3[1]) + mg(f3[2]); r4 -= mg(f4[0]) + mg(f4[1]) + mg(f4[2]) + mg(f4[3]);
				}
				if(c.fixed_psum) {
					// finest partition of this thread's 16 samples (stream_encoder.c:2934-2941 for a full block; the search kernel
					// only takes these sums when that partition is a whole number of threads)
					const uint32_t tz = (uint32_t)__ffs((int)n) - 1u, maxpo = tz < c.max_po ? tz : c.max_po;
					const uint32_t part = i0 / (n >> maxpo);
					if(part < 64u) {
						atomicAdd(&s_fps[s][0][part], r0); atomicAdd(&s_fps[s][1][part], r1); atomicAdd(&s_fps[s][2][part], r2);
						atomicAdd(&s_fps[s][3][part], r3); atomicAdd(&s_fps[s][4][part], r4);
					}
				}
			}
		}
		else {
#pragma unroll
			for(int q = 0; q < PP_CH; q++) {
				const uint32_t i = i0 + (uint32_t)q;
				if(i < n) {
					const uint32_t a = (uint32_t)x[PP_HIST + q], b = (uint32_t)x[PP_HIST + q - 1], cc = (uint32_t)x[PP_HIST + q - 2], dd = (uint32_t)x[PP_HIST + q - 3], ee = (uint32_t)x[PP_HIST + q - 4];
					o |= a; d |= (a != (uint32_t)x0) ? 1u : 0u;
					// the side signal of 16-bit channels has 17 bits; bit 1 records a sample outside the 16-bit range
					if(s == 3) d |= (((a ^ (uint32_t)((int32_t)a >> 31)) >> 15) != 0u) ? 2u : 0u;
					if(i >= FA_MAX_FIXED_ORDER) { // fixed.c:229-243 sums start at the fifth sample
						const int32_t d0 = (int32_t)a, d1 = (int32_t)(a - b), d2 = (int32_t)(a - 2u*b + cc), d3 = (int32_t)(a - 3u*b + 3u*cc - dd), d4 = (int32_t)(a - 4u*b + 6u*cc - 4u*dd + ee);
						e0 += (uint32_t)(d0 < 0 ? -d0 : d0); e1 += (uint32_t)(d1 < 0 ? -d1 : d1); e2 += (uint32_t)(d2 < 0 ? -d2 : d2);
						e3 += (uint32_t)(d3 < 0 ? -d3 : d3); e4 += (uint32_t)(d4 < 0 ? -d4 : d4);
					}
				}
			}
		}
		sums[s][0] = e0; sums[s][1] = e1; sums[s][2] = e2; sums[s][3] = e3; sums[s][4] = e4; orv[s] = o; ne[s] = d;
	}

	// ---- wave level: 16 + 4 sums and 4 + 4 OR values through halving butterflies
	uint32_t va[16], vb[4], vo[8];
#pragma unroll
	for(int s = 0; s < 4; s++) {
#pragma unroll
		for(int k = 0; k < 4; k++) va[s * 4 + k] = sums[s][k];
		vb[s] = sums[s][4]; vo[s] = orv[s]; vo[4 + s] = ne[s];
	}
	wave_combine<16, false>(va, lane);   // lane l: total of va[l >> 2]
	wave_combine<4, false>(vb, lane);    // lane l: total of vb[l >> 4]
	wave_combine<8, true>(vo, lane);     // lane l: OR of vo[l >> 3]
	if((lane & 3u) == 0) { const uint32_t idx = lane >> 2; atomicAdd(&sh.err[idx >> 2][idx & 3u], (unsigned long long)va[0]); }
	if((lane & 15u) == 0) atomicAdd(&sh.err[lane >> 4][4], (unsigned long long)vb[0]);
	if((lane & 7u) == 0) { const uint32_t idx = lane >> 3; if(idx < 4u) atomicOr(&sh.orv[idx], vo[0]); else atomicOr(&sh.ne[idx - 4u], vo[0]); }
	__syncthreads();

	if(c.fixed_psum && whole) {
		uint32_t *dst = w.fixed_psum + (size_t)f * c.nsig * 320u;
		for(uint32_t t = tid; t < c.nsig * 320u; t += 256u) dst[t] = (&s_fps[0][0][0])[t];   // signal-major in both
	}
	if(tid < c.nsig * 5u) {
		const uint32_t s = tid / 5u, k = tid - s * 5u, g = f * c.nsig + s;
		const uint32_t all = sh.orv[s];
		const uint32_t wasted = all ? (uint32_t)__ffs((int)all) - 1u : 0u;
		unsigned long long v = sh.err[s][k] >> wasted;
		if(!c.wide_fixed) v &= 0xffffffffull;  // fixed.c:234: 32-bit accumulators wrap
		w.fixed_err[(size_t)k * c.G + g] = v;
		if(k == 0) { w.wasted[g] = (uint8_t)wasted; w.is_const[g] = (uint8_t)(((sh.ne[s] & 1u) ? 0u : 1u) | ((sh.ne[s] & 2u) ? 0u : 2u)); }   // bit 0: constant, bit 1: every sample fits 16 bits
	}
}

// The same for full-length frames of batches whose per-partition sums are wanted (FaCfg::fixed_psum, blocks a multiple
// of 16 samples, at most 64 finest partitions): the per-thread sums go to the partition table in LDS and nowhere else -
// a frame's totals are the sums of its table rows less the head terms that the two conventions count differently
// (the table leaves out sample i < k at order k, fixed.c:229-243 leaves out the first four samples at every order) -
// so no thread carries the twenty sums of the four signals: 64 registers, eight waves per SIMD.  This kernel runs beside
// the autocorrelation kernel, whose long-lived waves hold three fifths of the register file: what decides its speed is
// how many of its own waves fit next to them.
__global__ void __launch_bounds__(256, 8) fa_prep_stereo16_lean_kernel(const uint32_t *__restrict__ pcm, FaCfg c, FaWork w)
{
	__shared__ uint32_t s_fps[4][5][64];
	__shared__ uint32_t s_head[4][5], s_orv[4], s_ne[4];
	const uint32_t f = blockIdx.x, tid = threadIdx.x, lane = tid & 63u;
	const uint32_t n = fa_frame_len(c, w, f);
	if(n != c.N) return;                   // a short block: the kernel above takes it
	for(uint32_t t = tid; t < 4u * 5u * 64u; t += 256u) (&s_fps[0][0][0])[t] = 0u;
	if(tid < 4u) { s_orv[tid] = 0; s_ne[tid] = 0; }
	const uint32_t *__restrict__ p = pcm + (size_t)f * c.N;
	const uint32_t i0 = tid * PP_CH;
	const bool active = i0 < n;

	uint32_t raw[PP_HIST + PP_CH];
#pragma unroll
	for(int j = 0; j < (PP_HIST + PP_CH) / 4; j++) {
		const int32_t rel = (int32_t)i0 - PP_HIST + 4 * j;
		uint4 v = make_uint4(0u, 0u, 0u, 0u);
		if(active && rel >= 0) v = *(const uint4*)(p + rel);
		raw[4*j] = v.x; raw[4*j+1] = v.y; raw[4*j+2] = v.z; raw[4*j+3] = v.w;
	}
	const uint32_t tz = (uint32_t)__ffs((int)n) - 1u, maxpo = tz < c.max_po ? tz : c.max_po;
	const uint32_t part = i0 / (n >> maxpo);      // < 64 (host check)
	__syncthreads();

	uint32_t vo[8];
#pragma unroll
	for(int s = 0; s < 4; s++) {
		// left, right, (left + right) >> 1, left - right from the packed word; the sum and the difference by one two-term
		// multiply-add on the 16-bit halves
		auto sig = [&](uint32_t lr) -> uint32_t {
			if(s == 0) return (uint32_t)(int32_t)(int16_t)(lr & 0xffffu);
			if(s == 1) return (uint32_t)((int32_t)lr >> 16);
			if(s == 2) return (uint32_t)(dot2_i16(lr, 0x00010001u, 0) >> 1);
			return (uint32_t)dot2_i16(lr, 0xffff0001u, 0);
		};
		const uint32_t B = 0x80000000u;
		uint32_t o = 0, mag = 0, e0 = 0, e1 = 0, e2 = 0, e3 = 0, e4 = 0;
		// (every signal unpacks the words again: kept unpacked for all four, they would take forty registers)
#pragma unroll
		for(int j = 0; j < PP_HIST + PP_CH; j++) asm volatile("" : "+v"(raw[j]));
		const uint32_t h1 = sig(raw[3]), h2 = sig(raw[2]), h3 = sig(raw[1]), h4 = sig(raw[0]);
		uint32_t pxb = h1 ^ B, pd1, pd2, pd3;
		{
			const uint32_t a1 = h1 - h2, a2 = h2 - h3, a3 = h3 - h4;   // d1(-1), d1(-2), d1(-3)
			const uint32_t b1 = a1 - a2, b2 = a2 - a3;                  // d2(-1), d2(-2)
			pd1 = a1 ^ B; pd2 = b1 ^ B; pd3 = (b1 - b2) ^ B;
		}
#pragma unroll
		for(int q = 0; q < PP_CH; q++) {
			const uint32_t a = sig(raw[PP_HIST + q]), ab = a ^ B;
			const uint32_t d1b = (ab - pxb) ^ B, d2b = (d1b - pd1) ^ B, d3b = (d2b - pd2) ^ B;   // the biases cancel in each difference
			e0 = sad_u32(ab, B, e0); e1 = sad_u32(ab, pxb, e1); e2 = sad_u32(d1b, pd1, e2); e3 = sad_u32(d2b, pd2, e3); e4 = sad_u32(d3b, pd3, e4);
			o |= a;
			if(s == 3) mag |= a ^ (uint32_t)((int32_t)a >> 31);
			pxb = ab; pd1 = d1b; pd2 = d2b; pd3 = d3b;
		}
		if(tid == 0) {
			// samples 0..3 of the frame: history is zero, so d_k(q) is the k-th difference of (0, 0, 0, 0, x0, x1, x2, x3)
			const uint32_t s0 = sig(raw[4]), s1 = sig(raw[5]), s2 = sig(raw[6]), s3 = sig(raw[7]);
			const uint32_t f1[4] = { s0, s1 - s0, s2 - s1, s3 - s2 };
			const uint32_t f2[4] = { f1[0], f1[1] - f1[0], f1[2] - f1[1], f1[3] - f1[2] };
			const uint32_t f3[4] = { f2[0], f2[1] - f2[0], f2[2] - f2[1], f2[3] - f2[2] };
			const uint32_t f4[4] = { f3[0], f3[1] - f3[0], f3[2] - f3[1], f3[3] - f3[2] };
			auto mg = [](uint32_t v) { const int32_t t = (int32_t)v; return (uint32_t)(t < 0 ? -t : t); };
			// the table's convention: sample i counts at order k when i >= k
			e1 -= mg(f1[0]); e2 -= mg(f2[0]) + mg(f2[1]); e3 -= mg(f3[0]) + mg(f3[1]) + mg(f3[2]); e4 -= mg(f4[0]) + mg(f4[1]) + mg(f4[2]) + mg(f4[3]);
			// ... and what the frame totals leave out besides: samples k .. 3
			s_head[s][0] = mg(s0) + mg(s1) + mg(s2) + mg(s3); s_head[s][1] = mg(f1[1]) + mg(f1[2]) + mg(f1[3]);
			s_head[s][2] = mg(f2[2]) + mg(f2[3]); s_head[s][3] = mg(f3[3]); s_head[s][4] = 0u;
		}
		if(active) {
			atomicAdd(&s_fps[s][0][part], e0); atomicAdd(&s_fps[s][1][part], e1); atomicAdd(&s_fps[s][2][part], e2);
			atomicAdd(&s_fps[s][3][part], e3); atomicAdd(&s_fps[s][4][part], e4);
		}
		// (side signal of 16-bit channels, nominally 17 bits: a sample outside the 16-bit range)
		vo[s] = active ? o : 0u;
		vo[4 + s] = (active && s == 3 && (mag >> 15)) ? 2u : 0u;
	}
	wave_combine<8, true>(vo, lane);     // lane l: OR of vo[l >> 3]
	if((lane & 7u) == 0) { const uint32_t idx = lane >> 3; if(idx < 4u) atomicOr(&s_orv[idx], vo[0]); else atomicOr(&s_ne[idx - 4u], vo[0]); }
	__syncthreads();

	{
		uint32_t *dst = w.fixed_psum + (size_t)f * c.nsig * 320u;
		for(uint32_t t = tid; t < c.nsig * 320u; t += 256u) dst[t] = (&s_fps[0][0][0])[t];   // signal-major in both
	}
	// eight threads per (signal, order): the total of its 64 table entries
	const uint32_t pair = tid >> 3, j = tid & 7u;
	if(pair < c.nsig * 5u) {
		const uint32_t s = pair / 5u, k = pair - s * 5u, g = f * c.nsig + s;
		const uint4 a = ((const uint4*)&s_fps[s][k][8u * j])[0], b = ((const uint4*)&s_fps[s][k][8u * j])[1];
		unsigned long long v = (unsigned long long)a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
		v += (unsigned long long)__shfl_xor((long long)v, 1, 64); v += (unsigned long long)__shfl_xor((long long)v, 2, 64); v += (unsigned long long)__shfl_xor((long long)v, 4, 64);
		// a signal is constant when no sample differs from the one before it: the first differences' row adds up to zero
		const uint4 a1 = ((const uint4*)&s_fps[s][1][8u * j])[0], b1 = ((const uint4*)&s_fps[s][1][8u * j])[1];
		uint32_t moves = a1.x | a1.y | a1.z | a1.w | b1.x | b1.y | b1.z | b1.w;
		moves |= (uint32_t)__shfl_xor((int)moves, 1, 64); moves |= (uint32_t)__shfl_xor((int)moves, 2, 64); moves |= (uint32_t)__shfl_xor((int)moves, 4, 64);
		if(j == 0u) {
			const uint32_t all = s_orv[s];
			const uint32_t wasted = all ? (uint32_t)__ffs((int)all) - 1u : 0u;
			v = (v - s_head[s][k]) >> wasted;
			if(!c.wide_fixed) v &= 0xffffffffull;  // fixed.c:234: 32-bit accumulators wrap
			w.fixed_err[(size_t)k * c.G + g] = v;
			if(k == 0) { w.wasted[g] = (uint8_t)wasted; w.is_const[g] = (uint8_t)((moves ? 0u : 1u) | ((s_ne[s] & 2u) ? 0u : 2u)); }   // bit 0: constant, bit 1: every sample fits 16 bits
		}
	}
}

// 16-bit stereo (left/right, plus mid/side when analysed) with blocks of at most 256 x 16 samples
extern "C" int fa_prep_fast_ok(const void *pcm, const FaCfg *c)
{
	return c->channels == 2u && c->pcm_format == FLACAMD_PCM_S16 && c->N <= 256u * PP_CH && (c->N % 4u) == 0 && ((uintptr_t)pcm % 16u) == 0 && !(c->debug & 0x2000u);
}

// ... of which the full-length frames can take the kernel that keeps its sums in the partition table
static int fa_prep_lean_ok(const FaCfg *c)
{
	if(!c->fixed_psum || (c->N % PP_CH) != 0 || (c->debug & 0x100000u)) return 0;
	uint32_t tz = 0;
	while(!((c->N >> tz) & 1u)) tz++;
	const uint32_t maxpo = tz < c->max_po ? tz : c->max_po;
	return maxpo <= 6u;
}

// st_short: where the kernel for the short blocks of a batch goes when the full-length frames have their own (it may
// be st; a different stream must be joined by the caller before the results are used)
extern "C" void fa_launch_prep(const void *pcm, const FaCfg *c, const FaWork *w, hipStream_t st, hipStream_t st_short)
{
	if(fa_prep_fast_ok(pcm, c)) {
		if(fa_prep_lean_ok(c)) {
			hipLaunchKernelGGL(fa_prep_stereo16_lean_kernel, dim3(c->F), dim3(256), 0, st, (const uint32_t*)pcm, *c, *w);
			// short blocks: anywhere in a batch of several streams, else the last frame at most
			if(c->multi) { if(c->nshort) hipLaunchKernelGGL(fa_prep_stereo16_kernel, dim3(c->nshort), dim3(256), 0, st_short, (const uint32_t*)pcm, *c, *w, 0u, 2u); }
			else if(c->last_n != c->N) hipLaunchKernelGGL(fa_prep_stereo16_kernel, dim3(1), dim3(256), 0, st_short, (const uint32_t*)pcm, *c, *w, c->F - 1u, 1u);
			return;
		}
		hipLaunchKernelGGL(fa_prep_stereo16_kernel, dim3(c->F), dim3(256), 0, st, (const uint32_t*)pcm, *c, *w, 0u, 0u);
		return;
	}
	if(c->spill) { hipLaunchKernelGGL(fa_prep_kernel<true>, dim3(c->G), dim3(256), 0, st, pcm, *c, *w, 0u); return; }
	const size_t lds = sizeof(int32_t) * (size_t)c->N;
	(void)hipFuncSetAttribute((const void*)fa_prep_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	// independent channels of 32-bit words in whole pieces: the full-length frames by the kernel that reads a frame once
	const bool whole = c->pcm_format == FLACAMD_PCM_S32 && (c->N % PL_CHUNK) == 0 && ((uintptr_t)pcm % 16u) == 0 && !(c->debug & 0x40000000u);
	const bool planar = whole && !c->midside && c->nsig == c->channels && c->channels >= 2u && c->channels <= 8u && c->bps <= 25u;
	const bool planar_ms = whole && c->midside && c->channels == 2u && c->nsig == 4u && c->bps + 1u <= 25u;
	if(planar || planar_ms) {
		const int32_t *p = (const int32_t*)pcm;
		if(planar_ms) hipLaunchKernelGGL((fa_prep_planar_kernel<2, true>), dim3(c->F), dim3(256), 0, st, p, *c, *w);
		else switch(c->channels) {
		case 2: hipLaunchKernelGGL(fa_prep_planar_kernel<2>, dim3(c->F), dim3(256), 0, st, p, *c, *w); break;
		case 3: hipLaunchKernelGGL(fa_prep_planar_kernel<3>, dim3(c->F), dim3(256), 0, st, p, *c, *w); break;
		case 4: hipLaunchKernelGGL(fa_prep_planar_kernel<4>, dim3(c->F), dim3(256), 0, st, p, *c, *w); break;
		case 5: hipLaunchKernelGGL(fa_prep_planar_kernel<5>, dim3(c->F), dim3(256), 0, st, p, *c, *w); break;
		case 6: hipLaunchKernelGGL(fa_prep_planar_kernel<6>, dim3(c->F), dim3(256), 0, st, p, *c, *w); break;
		case 7: hipLaunchKernelGGL(fa_prep_planar_kernel<7>, dim3(c->F), dim3(256), 0, st, p, *c, *w); break;
		default: hipLaunchKernelGGL(fa_prep_planar_kernel<8>, dim3(c->F), dim3(256), 0, st, p, *c, *w); break;
		}
		if(c->multi) { if(c->nshort) hipLaunchKernelGGL(fa_prep_kernel<false>, dim3(c->nshort * c->nsig), dim3(256), lds, st_short, pcm, *c, *w, 2u); }
		else if(c->last_n != c->N) hipLaunchKernelGGL(fa_prep_kernel<false>, dim3(c->nsig), dim3(256), lds, st_short, pcm, *c, *w, 1u);
		return;
	}
	hipLaunchKernelGGL(fa_prep_kernel<false>, dim3(c->G), dim3(256), lds, st, pcm, *c, *w, 0u);
}
