// Stage 3, register-resident variant of the candidate search (see k_search.hip for what is
// replaced in the reference and for the general kernel).
//
// Applies to full-length blocks of at most 4096 samples whose finest partitions are whole
// multiples of 16 samples, to partition orders up to 6 and predictor orders up to HIST: the
// common streams.  One 256-thread workgroup per (frame, signal), 16 samples + HIST of history
// per thread, read straight from the interleaved PCM into registers (left/right/mid/side are
// formed from the same words): no LDS copy of the signal.  Every global load the workgroup
// needs is issued before the first barrier.
//
//   multiply-add phase  every thread evaluates all candidates on its 16 samples; 16-bit signals
//                       take two taps per instruction (v_dot2_i32_i16 on packed pairs) and
//                       |x - pred| accumulates in one instruction (v_sad_u32); one LDS atomic per
//                       thread and candidate into the 64 finest-partition sums
//   evaluation phase    eight lanes per candidate, no barriers and no atomics: lane j of a group adds the
//                       finest partitions 8j .. 8j + 7 up to the partitions of the next three orders and prices
//                       all fifteen, three exchange steps inside the group give the sums of the three coarsest
//                       orders, the bits per order are summed over the group and the best order is chosen
//   first wave          ordered arg-min over the candidates (estimate, then index: the
//                       reference's strict-'<' keep-first rule) and the winner record
//
// Three barriers per workgroup.
#include "engine_types.h"
#include "residual.h"
#include "wave_ops.h"
#include <type_traits>

#define SF_THREADS 256
#define SF_CH 16
#define SF_W (HIST + SF_CH)
#define SF_WAVES (SF_THREADS / 64)
#define SF_PARTS 64u     // finest partitions one wave prices: partition orders up to 6
#define SF_LEVELS 7
#define SF_ROW 64u       // entries of a candidate's row of partition sums (finest order)

__device__ __forceinline__ uint32_t sf_limit_po(uint32_t limit, uint32_t blocksize, uint32_t order) // format.c:548-561
{
	while(limit > 0 && (blocksize >> limit) <= order) limit--;
	return limit;
}

// Rice parameter of a partition of nsamp residuals with |residual| sum `sum` (stream_encoder.c:4052:
// the smallest k with nsamp << k >= sum, clipped) and the bits it costs (:3929-3937, unsigned 32-bit)
template<typename SUM>
__device__ __forceinline__ void sf_price(SUM sum, uint32_t nsamp, bool use_given_k, uint32_t given_k, uint32_t k_limit, uint32_t &k, uint32_t &est)
{
	if(sizeof(SUM) == 4) {
		const uint32_t s32 = (uint32_t)sum;
		// nsamp << k has the bit length of sum (k = 0 when sum is the shorter; "| 1" gives a zero sum a length without changing any other)
		const int32_t d = __builtin_clz(nsamp) - __builtin_clz(s32 | 1u);
		k = (uint32_t)(d > 0 ? d : 0);
		k += ((nsamp << k) < s32) ? 1u : 0u;
		k = k < k_limit - 1u ? k : k_limit - 1u;
		if(use_given_k) k = given_k;
		// sum >> (k - 1), or sum << 1 for k = 0: the low word of the 33-bit 2 * sum shifted right by k
		est = (uint32_t)__umul24(k, nsamp) + __builtin_amdgcn_alignbit(s32 >> 31, s32 << 1, k) + (4u + nsamp - (nsamp >> 1));
	}
	else {
		const unsigned long long s64 = (unsigned long long)sum;
		k = 0;
		if(s64 > (unsigned long long)nsamp) {
			const uint32_t bl_s = 64u - (uint32_t)__clzll((long long)s64), bl_n = 32u - (uint32_t)__clz((int)nsamp);
			k = bl_s - bl_n;
			if(((unsigned long long)nsamp << k) < s64) k++;
		}
		if(k >= k_limit) k = k_limit - 1u;
		if(use_given_k) k = given_k;
		est = 4u + (1u + k) * nsamp + (k ? (uint32_t)(s64 >> (k - 1u)) : (uint32_t)(s64 << 1)) - (nsamp >> 1);
	}
}

// PK = 1: 16-bit stereo (left/right, plus mid/side when analysed), one packed dword per wide sample;
// PK = 2: the same from 32-bit samples (what process_interleaved hands over);
// PK = 3, 4: one channel, 16-bit and 32-bit samples.
// LIGHT = 1: no candidate of the stream can need the 64-bit prediction sum or more than 24-bit
// samples (decided on the host), so those paths and their registers are compiled out.
// CGT != 0: the candidate group size is that constant (every LDS offset an immediate) and the development switches of
// FaCfg::debug are compiled out; CGT == 0: both are run-time values.
template<typename SUM, int PK, int LIGHT, int CGT>
__device__ __forceinline__ void sf_body(const void *__restrict__ pcm, const FaCfg &c_in, const FaWork &w, uint32_t cg_arg)
{
	extern __shared__ __align__(16) unsigned char lds_raw[];
	const uint32_t cg = CGT ? (uint32_t)CGT : cg_arg;
	const uint32_t dbg = CGT ? 0u : c_in.debug;
	// the stream parameters the kernel works from: run-time values, or (CGT == 17) those of `flac -8` on 16-bit stereo
	// with blocks of 4096, which the launcher has checked field by field
	struct View { uint32_t N, bps, disable_constant, disable_verbatim, fixed_psum, k_limit, max_cand, max_lpc_order, max_po, midside, min_po, multi, nsig, cand_stride; };
	const View c = CGT == 17
		? View{ 4096u, 16u, 0u, 0u, 1u, 15u, 17u, 12u, 6u, 1u, 0u, 0u, 4u, 16u }
		: View{ c_in.N, c_in.bps, c_in.disable_constant, c_in.disable_verbatim, c_in.fixed_psum, c_in.k_limit, c_in.max_cand, c_in.max_lpc_order, c_in.max_po, c_in.midside, c_in.min_po, c_in.multi, c_in.nsig, c_in.cand_stride };
	auto cand_idx = [&](uint32_t sig, uint32_t slot) -> size_t { return ((size_t)sig * c.max_cand + slot) * c.cand_stride; };

	const uint32_t g = fa_xcd_signal(blockIdx.x, gridDim.x, c.nsig), f = g / c.nsig, s = g - f * c.nsig;   // the signals of a frame share an XCD
	const uint32_t n = c.N;
	if(c.multi && w.frame_len[f] != n) return;   // a stream's short final block: the general kernel codes it
	const uint64_t base = (uint64_t)f * c.N;
	const uint32_t tid = threadIdx.x, lane = tid & 63u;
	const uint32_t i0 = tid * SF_CH;
	const uint32_t wasted = (uint32_t)__builtin_amdgcn_readfirstlane((int)w.wasted[g]);
	const uint32_t ncand = (uint32_t)__builtin_amdgcn_readfirstlane((int)w.ncand[g]);
	const uint32_t sigflags = (uint32_t)__builtin_amdgcn_readfirstlane((int)w.is_const[g]);   // bit 0: constant signal, bit 1: every sample fits 16 bits
	const uint32_t is_const = sigflags & 1u;
	const uint32_t bps = c.bps - wasted + ((c.midside && s == 3u) ? 1u : 0u);

	// frame-level partition order range (stream_encoder.c:2934-2941); every block here is full length
	const uint32_t tz = (uint32_t)__ffs((int)n) - 1u;
	const uint32_t maxpo_f = tz < c.max_po ? tz : c.max_po;            // <= 6 (host check)
	const uint32_t minpo_f = c.min_po < maxpo_f ? c.min_po : maxpo_f;
	const uint32_t parts0 = 1u << maxpo_f, dps0 = n >> maxpo_f;
	const uint32_t emax = c.max_po < FLACAMD_MAX_PARTITION_ORDER ? c.max_po : FLACAMD_MAX_PARTITION_ORDER;
	const uint32_t parts_cap = 1u << emax;

	// LDS carve-up
	int32_t *cq = (int32_t*)lds_raw;                                   // [cg][HIST] coefficients
	SUM *psum = (SUM*)(cq + (size_t)cg * HIST);                        // [cg][128] |residual| sums of the partitions of every order
	unsigned long long *ckey = (unsigned long long*)(psum + (size_t)cg * SF_ROW);   // [cg] (estimate << 32) | candidate index
	uint32_t *cqp = (uint32_t*)(ckey + cg);                            // [cg][HIST/2] coefficient pairs for the two-tap instruction
	uint32_t *chdr = cqp + (size_t)cg * (HIST / 2);                    // [cg] candidate headers
	uint32_t *cinfo = chdr + cg;                                       // [cg] order | maxpo_c<<8 | minpo_c<<12 | narrow<<16
	uint32_t *cpo = cinfo + cg;                                        // [cg] best partition order

	const bool active = i0 < n;
	// samples and coefficients fit 16 bits: two taps per instruction.  The side channel of a 16-bit
	// stream has 17 bits nominally; the preparation stage notes when its samples fit 16 bits anyway
	// (left and right within 32767 of each other, the usual case)
	const bool use_dot2 = (bps <= 16u || ((sigflags & 2u) && c.bps <= 16u && !(dbg & 0x20000u))) && !(dbg & 0x8000u);
	const bool sad_on = sizeof(SUM) == 4 && !(dbg & 0x4000u);

	const uint32_t qrows = c.max_lpc_order ? c.max_lpc_order : 1u;
	const uint32_t part = i0 / dps0;
	const bool stamp = (dbg & 0x1000u) && tid == 0 && g < FA_DBG_WORDS / 8u;
#define SF_STAMP(i) do { if(stamp) w.dbg[(size_t)g * 8u + (i)] = (unsigned long long)clock64(); } while(0)
	SF_STAMP(0);

	// best candidate so far: kept by the first wave, uniform across its lanes
	uint32_t have_best = 0, best_est = 0, best_cand = 0xffffffffu, best_po = 0, best_rice2 = 0, best_hdr = 0;
	int32_t best_q = 0;   // lane j of the first wave: coefficient j of the best candidate
	FaWinner *win = &w.winner[g];

	for(uint32_t cb = 0; cb < ncand; cb += cg) {
		const uint32_t gn = (ncand - cb) < cg ? (ncand - cb) : cg;
		for(uint32_t t = tid; t < cg * SF_ROW; t += SF_THREADS) psum[t] = 0;
		// candidate headers and coefficients of this group: one parallel fetch into LDS, for every
		// slot of the group whether used or not, so that the loads do not wait for the candidate count
		for(uint32_t t = tid; t < cg * (HIST / 2 + 1u); t += SF_THREADS) {
			if(t < cg) chdr[t] = (cb + t < c.max_cand) ? w.cand_hdr[cand_idx(g, (cb + t))] : 0u;
			else {
				const uint32_t tt = t - cg, ci = tt / (HIST / 2), m = tt - ci * (HIST / 2);
				const bool ok = cb + ci < c.max_cand;
				// header and coefficient pair fetched side by side (a slot that does not exist reads the signal's first record
				// instead, a row the coefficient table does not have reads row 0: no load hangs on a condition another load decides);
				// rows at and above the candidate's order are not written by the model stage: taken as zero
				const size_t rec = cand_idx(g, ok ? cb + ci : 0u);
				const uint32_t hdr_ci = w.cand_hdr[rec];
				const int32_t l0 = w.cand_q[rec + (2u * m < qrows ? 2u * m : 0u)], l1 = w.cand_q[rec + (2u * m + 1u < qrows ? 2u * m + 1u : 0u)];
				const uint32_t ord = ok ? FA_CAND_ORDER(hdr_ci) : 0u;
				const int32_t q0 = (2u * m < ord && 2u * m < qrows) ? l0 : 0, q1 = (2u * m + 1u < ord && 2u * m + 1u < qrows) ? l1 : 0;
				cq[ci * HIST + 2u * m] = q0; cq[ci * HIST + 2u * m + 1u] = q1;
				cqp[tt] = pack_lo16(q0, q1);
			}
		}
		// the fixed predictors' partition sums, when the preparation stage left them: entry t and t + 256 of the signal's
		// [5][64] table (asked for here, used after the barrier)
		uint32_t fps_a = 0, fps_b = 0;
		const bool take_fps = c.fixed_psum && cb == 0u;
		if(take_fps) {
			const uint32_t *fp = w.fixed_psum + (size_t)g * 320u;
			fps_a = fp[tid];
			if(tid < 64u) fps_b = fp[256u + tid];
		}
		// ---- this thread's samples i0 - HIST .. i0 + SF_CH - 1, fetched anew for every candidate group (there is
		// one group unless a precision search multiplies the candidates) so that they do not occupy
		// registers during the evaluation phase; issued before anything waits
		int32_t xw[SF_W];
		if(PK == 4) {   // one channel, 32-bit samples: a dword per sample, rows of four
			uint32_t raw[SF_W];
#pragma unroll
			for(int j = 0; j < SF_W / 4; j++) {
				const int32_t rel = (int32_t)i0 - HIST + 4 * j;
				uint4 v = make_uint4(0u, 0u, 0u, 0u);
				if(active && rel >= 0) v = *(const uint4*)((const uint32_t*)pcm + base + (uint32_t)rel);
				raw[4*j] = v.x; raw[4*j+1] = v.y; raw[4*j+2] = v.z; raw[4*j+3] = v.w;
			}
#pragma unroll
			for(int j = 0; j < SF_W; j++) xw[j] = (int32_t)raw[j] >> wasted;
		}
		else if(PK == 3) {   // one channel, 16-bit samples: four to an 8-byte load
			uint32_t raw[SF_W / 2];
#pragma unroll
			for(int j = 0; j < SF_W / 4; j++) {
				const int32_t rel = (int32_t)i0 - HIST + 4 * j;
				uint2 v = make_uint2(0u, 0u);
				if(active && rel >= 0) v = *(const uint2*)((const uint16_t*)pcm + base + (uint32_t)rel);
				raw[2*j] = v.x; raw[2*j+1] = v.y;
			}
#pragma unroll
			for(int j = 0; j < SF_W / 2; j++) { xw[2*j] = (int32_t)(int16_t)(raw[j] & 0xffffu) >> wasted; xw[2*j+1] = (int32_t)raw[j] >> (16u + wasted); }
		}
		else if(PK == 1) {
			uint32_t raw[SF_W];
#pragma unroll
			for(int j = 0; j < SF_W / 4; j++) {
				const int32_t rel = (int32_t)i0 - HIST + 4 * j;
				uint4 v = make_uint4(0u, 0u, 0u, 0u);
				if(active && rel >= 0) v = *(const uint4*)((const uint32_t*)pcm + base + (uint32_t)rel);
				raw[4*j] = v.x; raw[4*j+1] = v.y; raw[4*j+2] = v.z; raw[4*j+3] = v.w;
			}
			// L, R, (L+R)>>1, L-R (stream_encoder.c:2051-2058), less the wasted bits.  Which one is the same for the whole
			// workgroup: one branch around the loop instead of arithmetic that covers all four per sample; and a signal
			// without wasted bits (nearly every one) takes the copy of the code in which the shift count is the literal 0
			auto form = [&](const uint32_t wz) {
				if(s == 0u) {
#pragma unroll
					for(int j = 0; j < SF_W; j++) xw[j] = (int32_t)(int16_t)(raw[j] & 0xffffu) >> wz;
				}
				else if(s == 1u) {
#pragma unroll
					for(int j = 0; j < SF_W; j++) xw[j] = (int32_t)raw[j] >> (16u + wz);
				}
				else if(s == 2u) {
#pragma unroll
					for(int j = 0; j < SF_W; j++) xw[j] = dot2_i16(raw[j], 0x00010001u, 0) >> (1u + wz);   // left + right by one two-term multiply-add on the halves
				}
				else {
#pragma unroll
					for(int j = 0; j < SF_W; j++) xw[j] = dot2_i16(raw[j], 0xffff0001u, 0) >> wz;            // left - right likewise
				}
			};
			if(wasted == 0u) form(0u); else form(wasted);
		}
		else {
			// 32-bit stereo: two wide samples per 16-byte load
			int32_t l[SF_W], r[SF_W];
#pragma unroll
			for(int j = 0; j < SF_W / 2; j++) {
				const int32_t rel = (int32_t)i0 - HIST + 2 * j;
				uint4 v = make_uint4(0u, 0u, 0u, 0u);
				if(active && rel >= 0) v = *(const uint4*)((const int32_t*)pcm + (base + (uint32_t)rel) * 2u);
				l[2*j] = (int32_t)v.x; r[2*j] = (int32_t)v.y; l[2*j+1] = (int32_t)v.z; r[2*j+1] = (int32_t)v.w;
			}
			if(s == 0u) {
#pragma unroll
				for(int j = 0; j < SF_W; j++) xw[j] = l[j] >> wasted;
			}
			else if(s == 1u) {
#pragma unroll
				for(int j = 0; j < SF_W; j++) xw[j] = r[j] >> wasted;
			}
			else if(s == 2u) {   // (mid and side only exist with c.midside)
#pragma unroll
				for(int j = 0; j < SF_W; j++) xw[j] = ((int32_t)((uint32_t)l[j] + (uint32_t)r[j]) >> 1) >> wasted;
			}
			else {
#pragma unroll
				for(int j = 0; j < SF_W; j++) xw[j] = (int32_t)((uint32_t)l[j] - (uint32_t)r[j]) >> wasted;
			}
		}

		__syncthreads();
		SF_STAMP(1);
		if(tid < gn) {
			const uint32_t order = FA_CAND_ORDER(chdr[tid]);
			const uint32_t maxpo_c = sf_limit_po(maxpo_f, n, order);
			const uint32_t minpo_c = minpo_f < maxpo_c ? minpo_f : maxpo_c;
			// the reference's accumulator at the candidate's own finest order wraps at 32 bits when
			// ilog2(partition length) + bps < 32 (stream_encoder.c:3808)
			const uint32_t narrow_c = (fa_ilog2(n >> maxpo_c) + bps < 32u) ? 1u : 0u;
			cinfo[tid] = order | (maxpo_c << 8) | (minpo_c << 12) | (narrow_c << 16);
		}

		if(take_fps) {
			// rows of the fixed candidates (the leading ones) from the table: entry (k, p) belongs to the candidate of order k, if
			// there is one; every thread of the workgroup takes part, whatever the block length
			for(uint32_t ci = 0; ci < gn && FA_CAND_KIND(chdr[ci]) == 0; ci++) {
				const uint32_t order = FA_CAND_ORDER(chdr[ci]);
				if((tid >> 6) == order) psum[(size_t)ci * SF_ROW + (tid & 63u)] = (SUM)(fps_a >> wasted);
				if(order == 4u && tid < 64u) psum[(size_t)ci * SF_ROW + tid] = (SUM)(fps_b >> wasted);
			}
		}

		// ---- |residual| sums per finest partition.  Samples before `order` carry no residual: each
		// sample of the chunk goes (statically) to one of three accumulators, of which thread 0 keeps
		// only the tail, thread 1 the tail and the head, every other thread all three.
		// Every lane of a wave that holds samples goes through the sums (the lanes beyond the block's end on zeros, depositing
		// nothing): the candidate loop below votes across the wave and reads the candidates' headers out of its lanes
		if(__any(active ? 1 : 0) && !(dbg & 1u)) {
			SUM accT, accH, accM;
			auto by_residual = [&](int sidx, int32_t rv, int P) {
				const uint32_t m = abs_u32(rv);
				if(sidx + SF_CH < P) accM += m; else if(sidx < P) accH += m; else accT += m;
			};
			auto by_abs = [&](int sidx, int32_t x, int32_t pred, int P) { by_residual(sidx, (int32_t)((uint32_t)x - (uint32_t)pred), P); };
			// |x - pred| + acc in one instruction on operands biased into unsigned order; equals the
			// reference's abs(x - pred) whenever x - pred does not wrap, which |pred| <= 2^30 (shift >= 1) ensures
			auto by_sad = [&](int sidx, int32_t x, int32_t pred, int P) {
				const uint32_t xb = (uint32_t)x ^ 0x80000000u, pb = (uint32_t)pred ^ 0x80000000u;
				if(sidx + SF_CH < P) accM = (SUM)sad_u32(xb, pb, (uint32_t)accM); else if(sidx < P) accH = (SUM)sad_u32(xb, pb, (uint32_t)accH); else accT = (SUM)sad_u32(xb, pb, (uint32_t)accT);
			};
			auto deposit = [&](uint32_t ci) {
				const SUM v = accT + (tid >= 1u ? accH : (SUM)0) + (tid >= 2u ? accM : (SUM)0);
				if(active) atomicAdd(&psum[(size_t)ci * SF_ROW + part], v);
			};
#define SF_LOAD_Q(ci, q) do { \
				const int4 *q4_ = (const int4*)(cq + (ci) * HIST); \
				const int4 qa_ = q4_[0], qb_ = q4_[1], qc_ = q4_[2]; \
				q[0] = qa_.x; q[1] = qa_.y; q[2] = qa_.z; q[3] = qa_.w; q[4] = qb_.x; q[5] = qb_.y; q[6] = qb_.z; q[7] = qb_.w; q[8] = qc_.x; q[9] = qc_.y; q[10] = qc_.z; q[11] = qc_.w; } while(0)
			// ---- fixed predictors: the leading candidates (the model stage emits them first, ascending).  All
			// five difference orders come out of one pass: d_k(i) = d_(k-1)(i) - d_(k-1)(i-1), and |d_k| is added
			// up by the |a - b| + c instruction on the two order-(k-1) values (biased into unsigned order; the
			// differences of samples of at most 25 bits stay far from wrapping).  Sample i < k has no order-k
			// residual: those go to separate sums that the frame's first thread drops.
			uint32_t nfixed = 0;
			while(nfixed < gn && FA_CAND_KIND(chdr[nfixed]) == 0) nfixed++;
			if(nfixed && !take_fps) {
				// Per-thread sums in 32 bits where sixteen magnitudes cannot carry out of them: always for 32-bit partition sums,
				// and for samples of at most 25 bits (a fourth difference stays below 2^28); then |a - b| + c in one instruction
				auto fixed_pass = [&](auto zero, auto always_sad) __attribute__((always_inline)) {
					typedef decltype(zero) ACC;
					const bool use_sad = decltype(always_sad)::value || sad_on;
					ACC fT[5], fH[5];
#pragma unroll
					for(int k = 0; k < 5; k++) { fT[k] = 0; fH[k] = 0; }
					const uint32_t B = 0x80000000u;
					auto add_abs = [&](ACC &acc, uint32_t a, uint32_t b) {
						if(sizeof(ACC) == 4 && use_sad) acc = (ACC)sad_u32(a ^ B, b ^ B, (uint32_t)acc);
						else acc += abs_u32((int32_t)(a - b));
					};
					uint32_t px = (uint32_t)xw[HIST - 1], pd1, pd2, pd3;
					{
						const uint32_t x2 = (uint32_t)xw[HIST - 2], x3 = (uint32_t)xw[HIST - 3], x4 = (uint32_t)xw[HIST - 4];
						const uint32_t e1 = x2 - x3, e2 = x3 - x4;      // d1(-2), d1(-3)
						pd1 = px - x2;
						const uint32_t g2 = e1 - e2;                      // d2(-2)
						pd2 = pd1 - e1;
						pd3 = pd2 - g2;
					}
#pragma unroll
					for(int j = 0; j < SF_CH; j++) {
						const uint32_t a = (uint32_t)xw[HIST + j];
						const uint32_t d1 = a - px, d2 = d1 - pd1, d3 = d2 - pd2;
						add_abs(fT[0], a, 0u);
						add_abs(j < 1 ? fH[1] : fT[1], a, px);
						add_abs(j < 2 ? fH[2] : fT[2], d1, pd1);
						add_abs(j < 3 ? fH[3] : fT[3], d2, pd2);
						add_abs(j < 4 ? fH[4] : fT[4], d3, pd3);
						px = a; pd1 = d1; pd2 = d2; pd3 = d3;
					}
					for(uint32_t ci = 0; ci < nfixed; ci++) {
						const uint32_t order = FA_CAND_ORDER(chdr[ci]);
						SUM vT = (SUM)fT[4], vH = (SUM)fH[4];
						if(order == 0u) { vT = (SUM)fT[0]; vH = 0; } else if(order == 1u) { vT = (SUM)fT[1]; vH = (SUM)fH[1]; } else if(order == 2u) { vT = (SUM)fT[2]; vH = (SUM)fH[2]; } else if(order == 3u) { vT = (SUM)fT[3]; vH = (SUM)fH[3]; }
						if(active) atomicAdd(&psum[(size_t)ci * SF_ROW + part], vT + (tid >= 1u ? vH : (SUM)0));
					}
				};
				if constexpr(sizeof(SUM) == 4) fixed_pass((uint32_t)0, std::false_type{});
				else {
					if(bps <= 25u && !(dbg & 0x4000u)) fixed_pass((uint32_t)0, std::true_type{});
					else fixed_pass((SUM)0, std::false_type{});
				}
			}
			if(use_dot2) {
				// packed pairs xp[t] = (x[t], x[t-1]); the unpacked history is dead from here on
				uint32_t xp[SF_W];
				xp[0] = 0;
#pragma unroll
				for(int j = 1; j < SF_W; j++) xp[j] = pack_lo16(xw[j], xw[j - 1]);
				// |x - pred| with both sides moved into unsigned order by 2^(31 - shift): the sums start from 2^31, so that
				// a logical shift yields pred + 2^(31 - shift) directly, and the samples carry the same offset (|x| < 2^15 and
				// shift <= 15 keep x + offset positive).  The offset samples cost an instruction each per shift value, so the
				// candidates are taken shift by shift, not in index order (their place in the arg-min is their index either
				// way): each lane below gn holds the header of candidate `lane`, a ballot collects the candidates that share
				// the shift of the first one not yet done.  A signal's twelve orders have two or three shift values between them.
				uint32_t xs[SF_CH];
				const uint32_t lane_hdr = (lane < gn) ? chdr[lane] : 0u;
				unsigned long long todo = __ballot(lane >= nfixed && lane < gn);
				while(todo) {
					const uint32_t first = (uint32_t)__builtin_ctzll(todo);
					const uint32_t shift = FA_CAND_SHIFT((uint32_t)__builtin_amdgcn_readlane((int)lane_hdr, (int)first));
					unsigned long long grp = __ballot(FA_CAND_SHIFT(lane_hdr) == shift) & todo;
					todo &= ~grp;
					if(sad_on && shift >= 1u) {
						const uint32_t off = 1u << (31u - shift);
#pragma unroll
						for(int j = 0; j < SF_CH; j++) xs[j] = (uint32_t)xw[HIST + j] + off;
					}
					while(grp) {
						const uint32_t ci = (uint32_t)__builtin_ctzll(grp);
						grp &= grp - 1ull;
						const uint32_t hdr = (uint32_t)__builtin_amdgcn_readlane((int)lane_hdr, (int)ci);
						const uint32_t order = FA_CAND_ORDER(hdr);
						accT = 0; accH = 0; accM = 0;
						if(!LIGHT && FA_CAND_WIDE(hdr)) { int32_t q[HIST]; SF_LOAD_Q(ci, q); lpc_pred_each_dispatch<MAC_WIDE>(order, xw, q, shift, by_abs); }
						else {
							uint32_t qp[HIST / 2];
							const uint2 *q2 = (const uint2*)(cqp + ci * (HIST / 2));
							const uint2 pa = q2[0], pb = q2[1], pc = q2[2];
							qp[0] = pa.x; qp[1] = pa.y; qp[2] = pb.x; qp[3] = pb.y; qp[4] = pc.x; qp[5] = pc.y;
							if(sad_on && shift >= 1u) {
								auto by_sum = [&](int sidx, int32_t sum, int P) {
									const uint32_t pb = (uint32_t)sum >> shift;
									if(sidx + SF_CH < P) accM = (SUM)sad_u32(xs[sidx], pb, (uint32_t)accM); else if(sidx < P) accH = (SUM)sad_u32(xs[sidx], pb, (uint32_t)accH); else accT = (SUM)sad_u32(xs[sidx], pb, (uint32_t)accT);
								};
								lpc_sum_each_dot2_dispatch(order, xp, qp, (int32_t)0x80000000u, by_sum);
							}
							else lpc_pred_each_dot2_dispatch(order, xw, xp, qp, shift, by_abs);
						}
						deposit(ci);
					}
				}
			}
			else {
				for(uint32_t ci = nfixed; ci < gn; ci++) {
					const uint32_t hdr = chdr[ci];
					const uint32_t order = FA_CAND_ORDER(hdr), shift = FA_CAND_SHIFT(hdr);
					accT = 0; accH = 0; accM = 0;
					{
						int32_t q[HIST];
						SF_LOAD_Q(ci, q);
						if(!LIGHT && FA_CAND_WIDE(hdr)) lpc_pred_each_dispatch<MAC_WIDE>(order, xw, q, shift, by_abs);
						else if(!LIGHT && bps > 24u) lpc_pred_each_dispatch<MAC_I32>(order, xw, q, shift, by_abs);
						else if(sad_on && shift >= 1u) lpc_pred_each_dispatch<MAC_I24>(order, xw, q, shift, by_sad);
						else lpc_pred_each_dispatch<MAC_I24>(order, xw, q, shift, by_abs);
					}
					deposit(ci);
				}
			}
		}
#undef SF_LOAD_Q
		SF_STAMP(2);
		__syncthreads();
		SF_STAMP(3);

		// ---- eight threads per candidate, thread j of a group taking the finest partitions 8j .. 8j + 7 of its
		// candidate's row: it adds them up to the 4 + 2 + 1 partitions of the next three orders that lie inside its
		// span and prices all fifteen.  The partitions of the three coarsest orders (4 + 2 + 1 more) span several
		// threads: their sums come from three exchange steps inside the group (lane ^ 1, lane ^ 2, lane <-> 7 - lane)
		// and are priced by the lanes {1, 3, 4, 6}, {2, 5} and {7} as the chain passes their level - every pair
		// and every half of the group holds exactly one of these lanes.  Bits per order: plain sums over the
		// group for the four inner levels; for the outer three the same exchange steps taken in the opposite
		// order walk lane 0 through the totals of level 6, 6 + 5, 6 + 5 + 4.  Lane 0 of the group then picks the
		// best order (searched max -> min, strict '<') and leaves estimate and order; the Rice parameters are
		// not kept: the first wave works them out again for the one candidate that wins.
		if(!(dbg & 2u)) {
			const uint32_t j = tid & 7u;
			const uint32_t lvU = (0x64544540u >> (4u * j)) & 15u;                 // level of this lane's outer partition (lane 0: none)
			const bool firstU = j == 1u || j == 2u || j == 7u;                     // ... and whether it is the first partition of that level
			const bool in_frame = j * 8u < parts0;                                 // the span holds partitions of this frame
			const int32_t room = (int32_t)parts0 - (int32_t)(j * 8u);             // finest partitions of the frame from the span's start on
			for(uint32_t ci = tid >> 3; ci < gn; ci += SF_THREADS / 8u) {
				const uint32_t hdr = chdr[ci], info = cinfo[ci];
				const uint32_t order = info & 255u, maxpo_c = (info >> 8) & 15u, minpo_c = (info >> 12) & 15u;
				const bool narrow_c = sizeof(SUM) == 8 && ((info >> 16) & 1u);
				const uint32_t lvl_c = maxpo_f - maxpo_c;
				const uint32_t given_k = FA_CAND_K(hdr);
				const SUM *row = psum + (size_t)ci * SF_ROW + j * 8u;
				SUM f0[8], f1[4], f2[2], f3;
				if(sizeof(SUM) == 4) {
					const uint4 a = ((const uint4*)row)[0], b = ((const uint4*)row)[1];
					f0[0] = (SUM)a.x; f0[1] = (SUM)a.y; f0[2] = (SUM)a.z; f0[3] = (SUM)a.w; f0[4] = (SUM)b.x; f0[5] = (SUM)b.y; f0[6] = (SUM)b.z; f0[7] = (SUM)b.w;
				}
				else {
#pragma unroll
					for(int i = 0; i < 8; i++) f0[i] = row[i];
				}
				// a value at the candidate's own finest order is wrapped to 32 bits when the reference's
				// accumulator there is 32 bits wide; coarser orders add the wrapped values
#define SF_WRAP(L, v) do { if(narrow_c && lvl_c == (uint32_t)(L)) (v) &= (SUM)0xffffffffull; } while(0)
#pragma unroll
				for(int i = 0; i < 8; i++) SF_WRAP(0, f0[i]);
#pragma unroll
				for(int i = 0; i < 4; i++) { f1[i] = f0[2*i] + f0[2*i+1]; SF_WRAP(1, f1[i]); }
#pragma unroll
				for(int i = 0; i < 2; i++) { f2[i] = f1[2*i] + f1[2*i+1]; SF_WRAP(2, f2[i]); }
				f3 = f2[0] + f2[1]; SF_WRAP(3, f3);
				SUM u = block_sum_step<1>(f3), vU; SF_WRAP(4, u); vU = u;
				u = block_sum_step<2>(u); SF_WRAP(5, u); if(lvU == 5u) vU = u;
				u = block_sum_step<3>(u); SF_WRAP(6, u); if(lvU == 6u) vU = u;
#undef SF_WRAP
				// bits of a partition of level L (partition order maxpo_f - L); the first partition of a level is `order` samples
				// short, and order 0 takes the suggested parameter (stream_encoder.c:3982-4029)
				auto bits = [&](SUM sum, uint32_t L, bool first) -> uint32_t {
					uint32_t k, est;
					sf_price<SUM>(sum, (dps0 << L) - (first ? order : 0u), first && L == maxpo_f, given_k, c.k_limit, k, est);
					return est;
				};
				uint32_t e0 = 0, e1 = 0, e2 = 0, e3 = 0, eU;
				if(parts0 >= 8u) {
#pragma unroll
					for(int i = 0; i < 8; i++) e0 += bits(f0[i], 0u, i == 0 && j == 0u);
#pragma unroll
					for(int i = 0; i < 4; i++) e1 += bits(f1[i], 1u, i == 0 && j == 0u);
#pragma unroll
					for(int i = 0; i < 2; i++) e2 += bits(f2[i], 2u, i == 0 && j == 0u);
					e3 = bits(f3, 3u, j == 0u);
				}
				else {
					// fewer than eight finest partitions: the span reaches beyond the frame
#pragma unroll
					for(int i = 0; i < 8; i++) { const uint32_t e = bits(f0[i], 0u, i == 0 && j == 0u); e0 += (i < room) ? e : 0u; }
#pragma unroll
					for(int i = 0; i < 4; i++) { const uint32_t e = bits(f1[i], 1u, i == 0 && j == 0u); e1 += (2 * i < room) ? e : 0u; }
#pragma unroll
					for(int i = 0; i < 2; i++) { const uint32_t e = bits(f2[i], 2u, i == 0 && j == 0u); e2 += (4 * i < room) ? e : 0u; }
					e3 = bits(f3, 3u, j == 0u);
				}
				eU = bits(vU, lvU, firstU);
				if(!in_frame) { e0 = 0; e1 = 0; e2 = 0; e3 = 0; }
				if(!in_frame || lvU == 0u) eU = 0;
				// totals over the group
				e0 = block_sum_step<1>(e0); e1 = block_sum_step<1>(e1); e2 = block_sum_step<1>(e2); e3 = block_sum_step<1>(e3);
				e0 = block_sum_step<2>(e0); e1 = block_sum_step<2>(e1); e2 = block_sum_step<2>(e2); e3 = block_sum_step<2>(e3);
				e0 = block_sum_step<3>(e0); e1 = block_sum_step<3>(e1); e2 = block_sum_step<3>(e2); e3 = block_sum_step<3>(e3);
				const uint32_t r1 = eU + row_partner<4>(eU);          // lane 0: level 6
				const uint32_t r2 = r1 + row_partner<2>(r1);          //         levels 6 + 5
				const uint32_t r3 = r2 + row_partner<1>(r2);          //         levels 6 + 5 + 4
				const uint32_t lvbits[7] = { e0, e1, e2, e3, r3 - r2, r2 - r1, r1 };
				const uint32_t lv_lo = maxpo_f - maxpo_c, lv_hi = maxpo_f - minpo_c;
				uint32_t best = 0, bestlv = 0; bool have = false;
#pragma unroll
				for(int lv = 0; lv < 7; lv++) {
					const uint32_t b = 6u + lvbits[lv];
					const bool take = ((uint32_t)lv - lv_lo <= lv_hi - lv_lo) && (!have || b < best);
					best = take ? b : best; bestlv = take ? (uint32_t)lv : bestlv; have = have || take;
				}
				const uint32_t bestpo = maxpo_f - bestlv;
				uint32_t est = 8u + wasted + order * bps + best;                         // stream_encoder.c:3550
				if(FA_CAND_KIND(hdr)) est = 8u + wasted + 9u + order * (FA_CAND_PREC(hdr) + bps) + best; // :3636
				if(j == 0u) { ckey[ci] = ((unsigned long long)est << 32) | (unsigned long long)(cb + ci); cpo[ci] = bestpo; }
			}
		}
		SF_STAMP(4);
		__syncthreads();
		SF_STAMP(5);

		// ---- first wave: ordered arg-min over the group (estimate, then candidate index)
		if(tid < 64u && !(dbg & 2u)) {
			unsigned long long bk = (lane < gn) ? ckey[lane] : ~0ull;
#pragma unroll
			for(int o = 32; o > 0; o >>= 1) { const unsigned long long t = (unsigned long long)__shfl_xor((long long)bk, o, 64); bk = t < bk ? t : bk; }
			const uint32_t est = (uint32_t)(bk >> 32), widx = (uint32_t)bk - cb;
			// earlier groups hold earlier candidates: replace only when strictly smaller
			if(!have_best || est < best_est) {
				have_best = 1; best_est = est; best_cand = cb + widx; best_po = cpo[widx];
				best_hdr = chdr[widx]; best_q = (lane < (uint32_t)HIST) ? cq[widx * HIST + lane] : 0;
				// Rice parameters of the winner at its best order: lane p starts from finest partition p, best_lv doubling steps
				// leave every lane with the sum of the partition it lies in
				const uint32_t info = cinfo[widx], order = info & 255u, maxpo_c = (info >> 8) & 15u;
				const bool narrow_c = sizeof(SUM) == 8 && ((info >> 16) & 1u);
				const uint32_t lvl_c = maxpo_f - maxpo_c, best_lv = maxpo_f - best_po;
				SUM v = (lane < parts0) ? psum[(size_t)widx * SF_ROW + lane] : (SUM)0;
				if(narrow_c && lvl_c == 0u) v &= (SUM)0xffffffffull;
#define SF_CHAIN(L) do { if(best_lv >= (uint32_t)(L)) { v = block_sum_step<L>(v); if(narrow_c && lvl_c == (uint32_t)(L)) v &= (SUM)0xffffffffull; } } while(0)
				SF_CHAIN(1); SF_CHAIN(2); SF_CHAIN(3); SF_CHAIN(4); SF_CHAIN(5); SF_CHAIN(6);
#undef SF_CHAIN
				uint32_t kw, estw;
				sf_price<SUM>(v, (n >> best_po) - ((lane >> best_lv) == 0u ? order : 0u), best_po == 0u, FA_CAND_K(best_hdr), c.k_limit, kw, estw);
				best_rice2 = __any((lane < parts0 && kw >= 15u) ? 1 : 0) ? 1u : 0u;
				// ... spread to the finest configured order so that the pack stage can index them by sample position alone
				for(uint32_t pe = 0; pe < parts_cap; pe += 64u) {
					const uint32_t kk = (uint32_t)__shfl((int)kw, (int)((((pe + lane) >> (emax - best_po)) << best_lv) & 63u), 64);
					if(pe + lane < parts_cap) win->k[pe + lane] = (uint8_t)kk;
				}
			}
		}
		if(cb + cg < ncand) __syncthreads();
	}
	SF_STAMP(6);

	// ---- final choice against verbatim / constant (stream_encoder.c:3205-3239, :3393-3396), first wave
	if(tid < 64u) {
		const uint32_t verb_bits = 8u + wasted + n * bps;
		const uint32_t base_bits = (c.disable_verbatim && n >= FA_MAX_FIXED_ORDER) ? 0xffffffffu : verb_bits;
		uint32_t type = FA_SUB_VERBATIM, est = base_bits;
		const bool constant = (!c.disable_constant) && n >= FA_MAX_FIXED_ORDER && is_const;
		if(constant) { const uint32_t cbits = 8u + wasted + bps; if(cbits < est) { type = FA_SUB_CONSTANT; est = cbits; } }
		bool use_cand = false;
		if(have_best && ncand > 0 && best_est < est) { use_cand = true; est = best_est; }
		if(est == 0xffffffffu) { type = FA_SUB_VERBATIM; est = verb_bits; }
		if(use_cand) {
			const uint32_t hdr = best_hdr;
			const uint32_t order = FA_CAND_ORDER(hdr);
			if(lane < FLACAMD_MAX_LPC_ORDER) win->q[lane] = (FA_CAND_KIND(hdr) && lane < order) ? best_q : 0;
			if(lane == 0) {
				const uint32_t ty = FA_CAND_KIND(hdr) ? FA_SUB_LPC : FA_SUB_FIXED;
				win->hdr[0] = ty | (order << 2) | (FA_CAND_PREC(hdr) << 8) | (FA_CAND_SHIFT(hdr) << 13) | (best_po << 18) | (best_rice2 << 22) | (wasted << 23) | (FA_CAND_WIDE(hdr) << 28);
				win->hdr[1] = est; win->hdr[2] = bps; win->hdr[3] = best_cand;
			}
		}
		else if(lane == 0) {
			win->hdr[0] = type | (wasted << 23);
			win->hdr[1] = est; win->hdr[2] = bps; win->hdr[3] = 0xffffffffu;
		}
	}
	SF_STAMP(7);
#undef SF_STAMP
}

// 256 threads x 16 samples, capped at 128 registers: four independent workgroups per CU (one wave per
// SIMD each) overlap each other's load, multiply-add and evaluation phases.  Measured alternatives on the
// headline workload: 512 x 8 samples at 2 / 3 workgroups per CU 4.9 / 4.3 ms per hour of audio, this shape
// 3.8 ms; tighter caps (5 or 6 workgroups) gain 2 % and spill 80-100 bytes per lane; evaluating three
// candidates side by side in a wave spills and gains nothing.  Other sample layouts take the general kernel.
template<typename SUM, int PK, int LIGHT, int CGT>
__global__ void __launch_bounds__(SF_THREADS, (PK == 2 || sizeof(SUM) == 8 || (PK == 4 && !LIGHT)) ? 3 : 4) fa_search_fast_kernel(const void *__restrict__ pcm, FaCfg c, FaWork w, uint32_t cg)
{
	sf_body<SUM, PK, LIGHT, CGT>(pcm, c, w, cg);
}
extern "C" size_t fa_search_fast_lds_bytes(const FaCfg *c, uint32_t cg, int sum64)
{
	(void)c;
	size_t b = (size_t)cg * HIST * 4u;               // cq
	b += (size_t)cg * SF_ROW * (sum64 ? 8u : 4u);    // psum
	b += (size_t)cg * 8u;                            // ckey
	b += (size_t)cg * (HIST / 2) * 4u;               // cqp
	b += (size_t)cg * 4u * 3u;                       // chdr, cinfo, cpo
	return (b + 15u) & ~(size_t)15u;
}

// Stereo streams, full-length blocks that fit 256 x 16 samples, with finest partitions that are multiples of
// 16 samples, at most 64 of them, and predictor orders within the register window.
extern "C" int fa_search_fast_ok(const void *pcm, const FaCfg *c)
{
	if((c->channels != 2u && c->channels != 1u) || ((uintptr_t)pcm % 16u) != 0) return 0;
	if(c->channels == 1u && (c->debug & 0x2000000u)) return 0;   // one or two channels, whole aligned rows of samples per load
	if(c->N > (uint32_t)(SF_THREADS * SF_CH) || (c->N % SF_CH) != 0 || c->max_lpc_order > (uint32_t)HIST || (c->debug & 0x800u)) return 0;
	uint32_t tz = 0;
	while(!((c->N >> tz) & 1u)) tz++;
	const uint32_t maxpo = tz < c->max_po ? tz : c->max_po;
	return maxpo <= 6u && ((c->N >> maxpo) % SF_CH) == 0;
}

// `frames` full-length frames starting at frame 0 of the batch
extern "C" void fa_launch_search_fast(const void *pcm, const FaCfg *c, const FaWork *w, uint32_t cg, int sum64, uint32_t frames, hipStream_t st)
{
	if(frames == 0) return;
	const size_t lds = fa_search_fast_lds_bytes(c, cg, sum64);
	const dim3 grid(frames * c->nsig), block(SF_THREADS);
	const int pk = c->channels == 1u ? (c->pcm_format == FLACAMD_PCM_S16 ? 3 : 4) : (c->pcm_format == FLACAMD_PCM_S16 ? 1 : 2);
	// can any candidate take the 64-bit prediction sum (stream_encoder.c:3597-3603: bps + precision + ilog2(order) > 32)
	// or carry more than 24-bit samples?
	uint32_t lg = 0;
	while((2u << lg) <= (c->max_lpc_order ? c->max_lpc_order : 1u)) lg++;
	const uint32_t max_prec = c->prec_search ? FA_MAX_QLP_PRECISION : c->qlp_precision;
	const uint32_t max_bps = c->bps + (c->midside ? 1u : 0u);
	const bool light = !sum64 && max_bps <= 24u && max_bps + max_prec + lg <= 32u;
#define SF_LAUNCH_CG(SUMT, PKV, LV, CGV) do { \
		(void)hipFuncSetAttribute((const void*)fa_search_fast_kernel<SUMT, PKV, LV, CGV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
		hipLaunchKernelGGL((fa_search_fast_kernel<SUMT, PKV, LV, CGV>), grid, block, lds, st, pcm, *c, *w, cg); } while(0)
#define SF_LAUNCH(SUMT, PKV, LV) SF_LAUNCH_CG(SUMT, PKV, LV, 0)
#define SF_LAUNCH_PK(SUMT, LV) do { if(pk == 1) SF_LAUNCH(SUMT, 1, LV); else if(pk == 2) SF_LAUNCH(SUMT, 2, LV); else if(pk == 3) SF_LAUNCH(SUMT, 3, LV); else SF_LAUNCH(SUMT, 4, LV); } while(0)
	// 16-bit stereo at the exhaustive presets (five fixed + twelve linear predictors per signal): the variant with the
	// group size and the production switches as constants
	const bool cd8 = !sum64 && light && pk == 1 && cg == 17u && c->debug == 0u && c->N == 4096u && c->bps == 16u && !c->disable_constant && !c->disable_verbatim
		&& c->fixed_psum == 1u && c->k_limit == 15u && c->max_cand == 17u && c->max_lpc_order == 12u && c->max_po == 6u && c->midside == 1u && c->min_po == 0u && c->multi == 0u
		&& c->nsig == 4u && c->cand_stride == 16u;
	if(cd8) SF_LAUNCH_CG(uint32_t, 1, 1, 17);
	else if(sum64) SF_LAUNCH_PK(unsigned long long, 0);
	else if(light) SF_LAUNCH_PK(uint32_t, 1);
	else SF_LAUNCH_PK(uint32_t, 0);
#undef SF_LAUNCH_PK
#undef SF_LAUNCH
#undef SF_LAUNCH_CG

}
