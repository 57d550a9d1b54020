// Stage 3, register-resident variant of the candidate search (see k_search.hip for what is
// replaced in the reference and for the general kernel).
//
// Applies to full-length blocks of at most 4096 samples whose finest partitions are whole
// multiples of 8 samples and to predictor orders up to HIST: the common streams.  One
// 512-thread workgroup per (frame, signal), 8 samples + HIST of history per thread, read
// straight from the interleaved PCM into registers (left/right/mid/side are formed from the
// same words): no LDS copy of the signal, so the workgroup's LDS holds only the partition
// sums and the small candidate table, and twice as many waves per SIMD stay resident to
// hide the load and barrier latencies around the multiply-add phase.  Every global load
// the workgroup needs is issued before the first barrier.  After the Rice evaluation the
// first wave alone finishes the job with shuffles (best partition order per candidate,
// ordered arg-min over candidates, winner record), so no further barriers are needed.
#include "engine_types.h"
#include "residual.h"

#define SF_THREADS 512
#define SF_CH 8
#define SF_W (HIST + SF_CH)

__device__ __forceinline__ uint32_t sf_limit_po(uint32_t limit, uint32_t blocksize, uint32_t order) // format.c:548-561
{
	while(limit > 0 && (blocksize >> limit) <= order) limit--;
	return limit;
}

// PK = 1: 16-bit stereo (left/right, plus mid/side when analysed), one packed dword per wide sample
template<typename SUM, int PK>
__global__ void __launch_bounds__(SF_THREADS) fa_search_fast_kernel(const void *__restrict__ pcm, FaCfg c, FaWork w, uint32_t cg)
{
	extern __shared__ __align__(16) unsigned char lds_raw[];

	const uint32_t g = blockIdx.x, f = g / c.nsig, s = g - f * c.nsig;
	const uint32_t n = c.N;
	const uint64_t base = (uint64_t)f * c.N;
	const uint32_t tid = threadIdx.x, lane = tid & 63u;
	const uint32_t i0 = tid * SF_CH;
	const uint32_t wasted = (uint32_t)__builtin_amdgcn_readfirstlane((int)w.wasted[g]);
	const uint32_t ncand = (uint32_t)__builtin_amdgcn_readfirstlane((int)w.ncand[g]);
	const uint32_t is_const = w.is_const[g];
	const uint32_t bps = c.bps - wasted + ((c.midside && s == 3u) ? 1u : 0u);

	// frame-level partition order range (stream_encoder.c:2934-2941); every block here is full length
	const uint32_t tz = (uint32_t)__ffs((int)n) - 1u;
	const uint32_t maxpo_f = tz < c.max_po ? tz : c.max_po;
	const uint32_t minpo_f = c.min_po < maxpo_f ? c.min_po : maxpo_f;
	const uint32_t parts0 = 1u << maxpo_f, dps0 = n >> maxpo_f;
	const uint32_t emax = c.max_po < FLACAMD_MAX_PARTITION_ORDER ? c.max_po : FLACAMD_MAX_PARTITION_ORDER;
	const uint32_t parts_cap = 1u << emax, prow = 2u * parts_cap;

	// LDS carve-up
	int32_t *cq = (int32_t*)lds_raw;                                   // [cg][HIST] coefficients
	SUM *psum = (SUM*)(cq + (size_t)cg * HIST);                        // [cg][prow] |residual| sums: finest order first, then each coarser one
	uint32_t *lbits = (uint32_t*)(psum + (size_t)cg * prow);           // [cg][16] bits per partition order
	uint32_t *chdr = lbits + (size_t)cg * 16u;                         // [cg] candidate headers
	uint32_t *cinfo = chdr + cg;                                       // [cg] order | maxpo_c<<8 | minpo_c<<12 | narrow<<16
	uint32_t *cqp = cinfo + cg;                                        // [cg][HIST/2] coefficient pairs for the two-tap instruction
	uint8_t *kbuf = (uint8_t*)(cqp + (size_t)cg * (HIST / 2));         // [cg][prow] Rice parameters

	// ---- this thread's samples i0 - HIST .. i0 + SF_CH - 1, issued before anything waits
	int32_t xw[SF_W];
	const bool active = i0 < n;
	if(PK) {
		uint32_t raw[SF_W];
#pragma unroll
		for(int j = 0; j < SF_W / 4; j++) {
			const int32_t rel = (int32_t)i0 - HIST + 4 * j;
			uint4 v = make_uint4(0u, 0u, 0u, 0u);
			if(active && rel >= 0) v = *(const uint4*)((const uint32_t*)pcm + base + (uint32_t)rel);
			raw[4*j] = v.x; raw[4*j+1] = v.y; raw[4*j+2] = v.z; raw[4*j+3] = v.w;
		}
		// signal = (ca * left + cb * right) >> (cs + wasted): L, R, (L+R)>>1, L-R (stream_encoder.c:2051-2058)
		const int32_t ca = (s == 1u) ? 0 : 1, cb = (s == 0u) ? 0 : (s == 3u ? -1 : 1);
		const uint32_t shv = wasted + (s == 2u ? 1u : 0u);
#pragma unroll
		for(int j = 0; j < SF_W; j++) {
			const int32_t l = (int32_t)(int16_t)(raw[j] & 0xffffu), r = (int32_t)raw[j] >> 16;
			xw[j] = (ca * l + cb * r) >> shv;
		}
	}
	else {
#pragma unroll
		for(int j = 0; j < SF_W; j++) {
			const int32_t rel = (int32_t)i0 - HIST + j;
			xw[j] = (active && rel >= 0) ? (fa_signal(pcm, c, s, base + (uint32_t)rel) >> wasted) : 0;
		}
	}

	// thread-level partial sums: 32 bits where the reference's own accumulators are 32 bits
	typedef SUM PART;
	// samples and coefficients fit 16 bits: two taps per instruction (the side channel of a
	// 16-bit stream has 17 bits and stays on the one-tap path)
	const bool use_dot2 = bps <= 16u && !(c.debug & 0x8000u);
	uint32_t xp[SF_W], xb[SF_CH];
	xp[0] = 0;
#pragma unroll
	for(int j = 1; j < SF_W; j++) xp[j] = pack_lo16(xw[j], xw[j - 1]);
#pragma unroll
	for(int j = 0; j < SF_CH; j++) xb[j] = (uint32_t)xw[HIST + j] ^ 0x80000000u;

	const uint32_t qrows = c.max_lpc_order ? c.max_lpc_order : 1u;
	const uint32_t part = i0 / dps0;
	const bool stamp = (c.debug & 0x1000u) && tid == 0 && g < FA_DBG_WORDS / 8u;
#define SF_STAMP(i) do { if(stamp) w.dbg[(size_t)g * 8u + (i)] = (unsigned long long)clock64(); } while(0)
	SF_STAMP(0);

	// best candidate so far: kept by the first wave, uniform across its lanes
	uint32_t have_best = 0, best_est = 0, best_cand = 0xffffffffu, best_po = 0, best_rice2 = 0, best_hdr = 0;
	int32_t best_q = 0;   // lane j of the first wave: coefficient j of the best candidate
	FaWinner *win = &w.winner[g];

	for(uint32_t cb = 0; cb < ncand; cb += cg) {
		const uint32_t gn = (ncand - cb) < cg ? (ncand - cb) : cg;
		for(uint32_t t = tid; t < gn * parts0; t += SF_THREADS) psum[(size_t)(t / parts0) * prow + (t % parts0)] = 0;
		for(uint32_t t = tid; t < gn * 16u; t += SF_THREADS) lbits[t] = 0;
		// candidate headers and coefficients of this group: one parallel fetch into LDS, for every
		// slot of the group whether used or not, so that the loads do not wait for the candidate count
		for(uint32_t t = tid; t < cg * (HIST / 2 + 1u); t += SF_THREADS) {
			if(t < cg) chdr[t] = (cb + t < c.max_cand) ? w.cand_hdr[(size_t)(cb + t) * c.G + g] : 0u;
			else {
				const uint32_t tt = t - cg, ci = tt / (HIST / 2), m = tt - ci * (HIST / 2);
				const bool ok = cb + ci < c.max_cand;
				// rows at and above the candidate's order are not written by the model stage: read them as zero
				const uint32_t ord = ok ? FA_CAND_ORDER(w.cand_hdr[(size_t)(cb + ci) * c.G + g]) : 0u;
				int32_t q0 = (ok && 2u * m < qrows) ? w.cand_q[((size_t)(cb + ci) * qrows + 2u * m) * c.G + g] : 0;
				int32_t q1 = (ok && 2u * m + 1u < qrows) ? w.cand_q[((size_t)(cb + ci) * qrows + 2u * m + 1u) * c.G + g] : 0;
				if(2u * m >= ord) q0 = 0;
				if(2u * m + 1u >= ord) q1 = 0;
				cq[ci * HIST + 2u * m] = q0; cq[ci * HIST + 2u * m + 1u] = q1;
				cqp[tt] = pack_lo16(q0, q1);
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

		// ---- |residual| sums per finest partition.  Samples before `order` carry no residual: each
		// sample of the chunk goes (statically) to one of three accumulators, of which thread 0 keeps
		// only the tail, thread 1 the tail and the head, every other thread all three.
		if(active && !(c.debug & 1u)) {
			for(uint32_t ci = 0; ci < gn; ci++) {
				const uint32_t hdr = chdr[ci];
				const uint32_t order = FA_CAND_ORDER(hdr);
				PART accT = 0, accH = 0, accM = 0;
				auto by_residual = [&](int sidx, int32_t rv, int P) {
					const uint32_t m = abs_u32(rv);
					if(sidx + SF_CH < P) accM += m; else if(sidx < P) accH += m; else accT += m;
				};
				auto by_abs = [&](int sidx, int32_t x, int32_t pred, int P) { by_residual(sidx, (int32_t)((uint32_t)x - (uint32_t)pred), P); };
				// |x - pred| + acc in one instruction on operands biased into unsigned order; equals the
				// reference's abs(x - pred) whenever x - pred does not wrap, which |pred| <= 2^30 (shift >= 1) ensures
				auto by_sad = [&](int sidx, int32_t, int32_t pred, int P) {
					const uint32_t pb = (uint32_t)pred ^ 0x80000000u;
					if(sidx + SF_CH < P) accM = (PART)sad_u32(xb[sidx], pb, (uint32_t)accM); else if(sidx < P) accH = (PART)sad_u32(xb[sidx], pb, (uint32_t)accH); else accT = (PART)sad_u32(xb[sidx], pb, (uint32_t)accT);
				};
				if(FA_CAND_KIND(hdr) == 0) fixed_each_dispatch(order, xw, by_residual);
				else {
					const uint32_t shift = FA_CAND_SHIFT(hdr);
					const bool sad_ok = sizeof(SUM) == 4 && shift >= 1u && bps <= 24u && !(c.debug & 0x4000u);
					if(FA_CAND_WIDE(hdr) || !sad_ok) {
						int32_t q[HIST];
						const int4 *q4 = (const int4*)(cq + ci * HIST);
						const int4 qa = q4[0], qb = q4[1], qc = q4[2];
						q[0] = qa.x; q[1] = qa.y; q[2] = qa.z; q[3] = qa.w; q[4] = qb.x; q[5] = qb.y; q[6] = qb.z; q[7] = qb.w; q[8] = qc.x; q[9] = qc.y; q[10] = qc.z; q[11] = qc.w;
						if(FA_CAND_WIDE(hdr)) lpc_pred_each_dispatch<MAC_WIDE>(order, xw, q, shift, by_abs);
						else lpc_pred_each_dispatch<MAC_I32>(order, xw, q, shift, by_abs);
					}
					else if(use_dot2) {
						uint32_t qp[HIST / 2];
						const uint2 *q2 = (const uint2*)(cqp + ci * (HIST / 2));
						const uint2 pa = q2[0], pb = q2[1], pc = q2[2];
						qp[0] = pa.x; qp[1] = pa.y; qp[2] = pb.x; qp[3] = pb.y; qp[4] = pc.x; qp[5] = pc.y;
						lpc_pred_each_dot2_dispatch(order, xw, xp, qp, shift, by_sad);
					}
					else {
						int32_t q[HIST];
						const int4 *q4 = (const int4*)(cq + ci * HIST);
						const int4 qa = q4[0], qb = q4[1], qc = q4[2];
						q[0] = qa.x; q[1] = qa.y; q[2] = qa.z; q[3] = qa.w; q[4] = qb.x; q[5] = qb.y; q[6] = qb.z; q[7] = qb.w; q[8] = qc.x; q[9] = qc.y; q[10] = qc.z; q[11] = qc.w;
						lpc_pred_each_dispatch<MAC_I24>(order, xw, q, shift, by_sad);
					}
				}
				const SUM v = (SUM)accT + (tid >= 1u ? (SUM)accH : (SUM)0) + (tid >= 2u ? (SUM)accM : (SUM)0);
				atomicAdd(&psum[(size_t)ci * prow + part], v);
			}
		}
		SF_STAMP(2);
		__syncthreads();
		SF_STAMP(3);

		// ---- sums of the coarser partition orders: pairwise merges, three orders per barrier.
		// A value read at the candidate's own finest order is first wrapped to 32 bits when the
		// reference's accumulator there is 32 bits wide.
		for(uint32_t l0 = 0; l0 < maxpo_f; l0 += 3) {
			const uint32_t lto = (l0 + 3u < maxpo_f) ? l0 + 3u : maxpo_f, span = lto - l0;
			const uint32_t cnt_to = parts0 >> lto, fan = 1u << span;
			for(uint32_t t = tid; t < gn * cnt_to; t += SF_THREADS) {
				const uint32_t ci = t / cnt_to, pg = t - ci * cnt_to;
				const uint32_t info = cinfo[ci];
				const uint32_t lvl_c = maxpo_f - ((info >> 8) & 15u);
				const bool narrow_c = sizeof(SUM) == 8 && ((info >> 16) & 1u);
				SUM *row = psum + (size_t)ci * prow;
				SUM v[8];
				const uint32_t off0 = 2u * parts0 - ((2u * parts0) >> l0);
#pragma unroll
				for(uint32_t i = 0; i < 8; i++) if(i < fan) { v[i] = row[off0 + pg * fan + i]; if(narrow_c && l0 == lvl_c) v[i] &= (SUM)0xffffffffull; }
				uint32_t width = fan;
#pragma unroll
				for(uint32_t d = 1; d <= 3; d++) if(d <= span) {
					width >>= 1;
					const uint32_t lv = l0 + d, off = 2u * parts0 - ((2u * parts0) >> lv);
#pragma unroll
					for(uint32_t i = 0; i < 4; i++) if(i < width) {
						SUM m = v[2*i] + v[2*i+1];
						row[off + pg * width + i] = m;
						if(narrow_c && lv == lvl_c) m &= (SUM)0xffffffffull;
						v[i] = m;
					}
				}
			}
			__syncthreads();
		}

		SF_STAMP(4);
		// ---- Rice parameter and bit estimate of every (candidate, partition order, partition)
		const uint32_t per = 2u * parts0, per_log = maxpo_f + 1u;
		for(uint32_t t = tid; t < ((c.debug & 2u) ? 0u : gn * per); t += SF_THREADS) {
			const uint32_t ci = t >> per_log, idx = t & (per - 1u);
			if(idx == per - 1u) continue;
			// idx in [0, parts0) is order maxpo_f, the next parts0/2 entries order maxpo_f-1, ...
			const uint32_t lvl = per_log - (32u - (uint32_t)__clz((int)(per - idx - 1u)));
			const uint32_t off = per - (per >> lvl);
			const uint32_t po = maxpo_f - lvl, p = idx - off;
			const uint32_t info = cinfo[ci];
			const uint32_t order = info & 255u, maxpo_c = (info >> 8) & 15u, minpo_c = (info >> 12) & 15u;
			if(po > maxpo_c || po < minpo_c) continue;
			const uint32_t nsamp = (n >> po) - (p == 0 ? order : 0u);
			uint32_t k, est;
			if(sizeof(SUM) == 4) {
				const uint32_t sum = (uint32_t)psum[(size_t)ci * prow + idx];
				if(po == 0) k = FA_CAND_K(chdr[ci]);   // stream_encoder.c:3982-4029
				else {                                  // :4052 smallest k with nsamp << k >= sum
					k = 0;
					if(sum > nsamp) {
						k = (uint32_t)__clz((int)nsamp) - (uint32_t)__clz((int)sum);   // nsamp << k has the bit length of sum
						if((nsamp << k) < sum) k++;
					}
					if(k >= c.k_limit) k = c.k_limit - 1u;
				}
				// :3929-3937, unsigned 32-bit arithmetic
				est = 4u + (1u + k) * nsamp + (k ? (sum >> (k - 1u)) : (sum << 1)) - (nsamp >> 1);
			}
			else {
				unsigned long long sum = psum[(size_t)ci * prow + idx];
				if(((info >> 16) & 1u) && po == maxpo_c) sum &= 0xffffffffull;
				if(po == 0) k = FA_CAND_K(chdr[ci]);
				else {
					k = 0;
					if(sum > (unsigned long long)nsamp) {
						const uint32_t bl_s = 64u - (uint32_t)__clzll((long long)sum), bl_n = 32u - (uint32_t)__clz((int)nsamp);
						k = bl_s - bl_n;
						if(((unsigned long long)nsamp << k) < sum) k++;
					}
					if(k >= c.k_limit) k = c.k_limit - 1u;
				}
				est = 4u + (1u + k) * nsamp + (k ? (uint32_t)(sum >> (k - 1u)) : (uint32_t)(sum << 1)) - (nsamp >> 1);
			}
			kbuf[(size_t)ci * prow + idx] = (uint8_t)k;
			atomicAdd(&lbits[ci * 16u + lvl], est);
		}
		__syncthreads();

		SF_STAMP(5);
		// ---- first wave: best partition order of each candidate (searched max -> min, strict '<'),
		// subframe estimate, ordered arg-min over the group (estimate, then candidate index)
		if(tid < 64u) {
			unsigned long long key = ~0ull; uint32_t mypo = 0;
			if(lane < gn) {
				const uint32_t hdr = chdr[lane], info = cinfo[lane];
				const uint32_t order = info & 255u, maxpo_c = (info >> 8) & 15u, minpo_c = (info >> 12) & 15u;
				uint32_t best = 0; bool have = false;
				for(int po = (int)maxpo_c; po >= (int)minpo_c; po--) {
					const uint32_t b = 6u + lbits[lane * 16u + (maxpo_f - (uint32_t)po)];
					if(!have || b < best) { best = b; mypo = (uint32_t)po; have = true; }
				}
				uint32_t est = 8u + wasted + order * bps + best;                         // stream_encoder.c:3550
				if(FA_CAND_KIND(hdr)) est = 8u + wasted + 9u + order * (FA_CAND_PREC(hdr) + bps) + best; // :3636
				key = ((unsigned long long)est << 32) | (unsigned long long)(cb + lane);
			}
			unsigned long long bk = key;
#pragma unroll
			for(int o = 32; o > 0; o >>= 1) { const unsigned long long t = (unsigned long long)__shfl_xor((long long)bk, o, 64); bk = t < bk ? t : bk; }
			const uint32_t est = (uint32_t)(bk >> 32), widx = (uint32_t)bk - cb;
			const uint32_t po_w = (uint32_t)__shfl((int)mypo, (int)widx, 64);
			// earlier groups hold earlier candidates: replace only when strictly smaller
			if(!have_best || est < best_est) {
				have_best = 1; best_est = est; best_cand = cb + widx; best_po = po_w;
				best_hdr = chdr[widx]; best_q = (lane < (uint32_t)HIST) ? cq[widx * HIST + lane] : 0;
				// Rice parameters of the winner, spread to the finest configured order so that the
				// pack stage can index them by sample position alone
				const uint32_t lvl = maxpo_f - po_w;
				const uint32_t off = per - (per >> lvl);
				bool big = false;
				for(uint32_t pe = lane; pe < parts_cap; pe += 64u) {
					const uint8_t k = kbuf[(size_t)widx * prow + off + (pe >> (emax - po_w))];
					win->k[pe] = k; big |= (k >= 15);
				}
				best_rice2 = __any(big ? 1 : 0) ? 1u : 0u;
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

extern "C" size_t fa_search_fast_lds_bytes(const FaCfg *c, uint32_t cg, int sum64)
{
	const size_t emax = c->max_po < FLACAMD_MAX_PARTITION_ORDER ? c->max_po : FLACAMD_MAX_PARTITION_ORDER;
	const size_t prow = (size_t)2 << emax;
	size_t b = (size_t)cg * HIST * 4u;               // cq
	b += (size_t)cg * prow * (sum64 ? 8u : 4u);      // psum
	b += (size_t)cg * 16u * 4u;                      // lbits
	b += (size_t)cg * 4u * 2u;                       // chdr, cinfo
	b += (size_t)cg * (HIST / 2) * 4u;               // cqp
	b += (size_t)cg * prow;                          // kbuf
	return (b + 15u) & ~(size_t)15u;
}

// Full-length blocks that fit 512 x 8 samples, with finest partitions that are multiples of
// 8 samples and predictor orders within the register window.
extern "C" int fa_search_fast_ok(const FaCfg *c)
{
	if(c->N > (uint32_t)(SF_THREADS * SF_CH) || (c->N % SF_CH) != 0 || c->max_lpc_order > (uint32_t)HIST || (c->debug & 0x800u)) return 0;
	uint32_t tz = 0;
	while(!((c->N >> tz) & 1u)) tz++;
	const uint32_t maxpo = tz < c->max_po ? tz : c->max_po;
	return ((c->N >> maxpo) % SF_CH) == 0;
}

// `frames` full-length frames starting at frame 0 of the batch
extern "C" void fa_launch_search_fast(const void *pcm, const FaCfg *c, const FaWork *w, uint32_t cg, int sum64, uint32_t frames, hipStream_t st)
{
	if(frames == 0) return;
	const size_t lds = fa_search_fast_lds_bytes(c, cg, sum64);
	const dim3 grid(frames * c->nsig), block(SF_THREADS);
	const bool pk = c->channels == 2u && c->pcm_format == FLACAMD_PCM_S16 && ((uintptr_t)pcm % 16u) == 0;
#define SF_LAUNCH(SUMT, PKV) do { \
		(void)hipFuncSetAttribute((const void*)fa_search_fast_kernel<SUMT, PKV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
		hipLaunchKernelGGL((fa_search_fast_kernel<SUMT, PKV>), grid, block, lds, st, pcm, *c, *w, cg); } while(0)
	if(sum64) { if(pk) SF_LAUNCH(unsigned long long, 1); else SF_LAUNCH(unsigned long long, 0); }
	else { if(pk) SF_LAUNCH(uint32_t, 1); else SF_LAUNCH(uint32_t, 0); }
#undef SF_LAUNCH
}
