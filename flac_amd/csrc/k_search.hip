// Stage 3 of the encoder: evaluate every residual-coding candidate of a signal and keep
// the one the reference would keep.
//
// Replaces, for one signal of one frame:
//   FLAC__fixed_compute_residual                       fixed.c:352-393
//   FLAC__lpc_compute_residual_from_qlp_coefficients   lpc.c:265-529 (32-bit wrapping sum)
//   ..._from_qlp_coefficients_wide                     lpc.c:531-791 (64-bit sum)
//   precompute_partition_info_sums_                    stream_encoder.c:3779-3847
//   set_partitioned_rice_ / count_rice_bits_in_partition_  :3923-4111
//   find_best_partition_order_                         :3671-3766
//   the keep-best chain of process_subframe_           :3205-3239, :3288-3291, :3376-3381, :3393
//
// One 256-thread workgroup per (frame, signal).  The signal is staged once into LDS
// (padded so that 16-sample register windows load without bank conflicts); each thread
// keeps 16 samples + 12 of history in registers and evaluates ALL candidates on them
// (coefficients are wave-uniform: scalar loads), so the signal is read from HBM/L2 once
// per workgroup, not once per candidate as on the CPU.  |residual| sums go to LDS
// per finest-level partition (one LDS atomic per thread per candidate), the Rice parameter
// / partition-order search runs over (candidate, order, partition) items in parallel, and
// an ordered arg-min (estimate, candidate index) reproduces the reference's strict-'<'
// keep-first tie-breaking.  Nothing but the winner record leaves the CU.
//
// This is integer work bound by VALU issue, not by HBM: ~(order+5) lane-ops per sample
// per LPC candidate.
#include "engine_types.h"
#include "residual.h"
#include "wave_ops.h"
#include "search_price.h"

// format.c:548-561
__device__ __forceinline__ uint32_t limit_po(uint32_t limit, uint32_t blocksize, uint32_t order)
{
	while(limit > 0 && (blocksize >> limit) <= order) limit--;
	return limit;
}

// Candidates per LDS round.  Blocks priced by the lanes of a wave (sp_price_group) keep SP_ROW sums per candidate and take
// `cg` candidates at a time; the others keep the sums of every partition order (2 * parts_cap entries) and take as many
// candidates as fit the same LDS.
__host__ __device__ __forceinline__ uint32_t search_cg_rows(uint32_t cg, uint32_t parts_cap)
{
	const uint32_t prow = 2u * parts_cap;
	if(prow <= SP_ROW) return cg;
	const uint32_t m = cg * SP_ROW / prow;
	return m ? m : 1u;
}
__host__ __device__ __forceinline__ uint32_t search_psum_entries(uint32_t cg, uint32_t parts_cap)
{
	const uint32_t a = cg * SP_ROW, b = search_cg_rows(cg, parts_cap) * 2u * parts_cap;
	return a > b ? a : b;
}

struct SearchShared {
	uint32_t best_est, best_cand, best_po, have_best;
	uint32_t rice2, group_won;
};

template<typename SUM, bool SPILL>
__global__ void __launch_bounds__(256, 4) fa_search_kernel(const void *__restrict__ pcm, FaCfg c, FaWork w, uint32_t cg_arg, uint32_t g0)
{
	extern __shared__ __align__(16) unsigned char lds_raw[];
	__shared__ SearchShared sh;

	// g0: first signal of the range this launch covers; the signals of a frame share an XCD.  With skip_full the launch covers
	// the listed short frames only (the others are coded by the register-resident kernel)
	const uint32_t gb = fa_xcd_signal(blockIdx.x, gridDim.x, c.nsig);
	const uint32_t g = c.skip_full ? w.short_list[gb / c.nsig] * c.nsig + gb % c.nsig : gb + g0, f = g / c.nsig, s = g - f * c.nsig;
	const uint32_t n = fa_frame_len(c, w, f);
	const uint64_t base = (uint64_t)f * c.N;
	const uint32_t tid = threadIdx.x;
	const uint32_t wasted = w.wasted[g];
	const uint32_t bps = c.bps - wasted + ((c.midside && s == 3u) ? 1u : 0u);
	const uint32_t ncand = w.ncand[g];

	// frame-level partition order range (stream_encoder.c:2934-2941)
	uint32_t maxpo_f;
	if(fa_frame_fractional(c, w, f)) maxpo_f = 0;
	else { const uint32_t tz = (uint32_t)__ffs((int)n) - 1u; maxpo_f = tz < c.max_po ? tz : c.max_po; }
	const uint32_t minpo_f = c.min_po < maxpo_f ? c.min_po : maxpo_f;
	const uint32_t parts0 = 1u << maxpo_f, dps0 = n >> maxpo_f;
	const bool fast = ((n & (CH - 1)) == 0) && ((dps0 & (CH - 1)) == 0);
	// the evaluation phase by lanes of a wave (at most 64 finest partitions), or over (candidate, order, partition) items
	const bool wprice = fast && maxpo_f <= 6u && !(c.debug & 0x800000u);

	// LDS carve-up
	// the workgroup's copy of the signal: in LDS, or for blocks too long for that in this signal's row of the spill buffer
	// (written and read by this workgroup only; the barriers below order it)
	const uint32_t xs_words = SPILL ? 0u : xs_words_for(c.N);
	int32_t *xs;
	if constexpr(SPILL) xs = w.xs_spill + (size_t)g * xs_words_for(c.N); else xs = (int32_t*)lds_raw;
	// |residual| sums of every partition at every order: row ci holds the finest order first
	// (parts0 entries), then each coarser order, 2*parts0 - 1 entries in all
	unsigned long long *psum = (unsigned long long*)(lds_raw + (((size_t)xs_words * 4u + 15u) & ~(size_t)15u)); // [cg][2 * parts_cap]
	const uint32_t parts_cap = 1u << (c.max_po < FLACAMD_MAX_PARTITION_ORDER ? c.max_po : FLACAMD_MAX_PARTITION_ORDER);
	const uint32_t prow = wprice ? SP_ROW : 2u * parts_cap;
	const uint32_t cg = wprice ? cg_arg : search_cg_rows(cg_arg, parts_cap);      // candidates per round of this block
	const uint32_t psum_entries = search_psum_entries(cg_arg, parts_cap);
	uint32_t *lbits = (uint32_t*)(psum + psum_entries);                           // [cg][16]
	unsigned long long *ckey = (unsigned long long*)(lbits + (size_t)cg_arg * 16u);    // [cg]
	long long *cinit = (long long*)(ckey + cg_arg);                                // [cg] what a linear predictor's 64-bit sums start from
	uint32_t *cpo = (uint32_t*)(cinit + cg_arg);                                   // [cg]
	uint8_t *kbuf = (uint8_t*)(cpo + cg_arg);                                      // [cg][2*parts_cap] (not kept by the wave pricing)
	uint8_t *best_k = kbuf + psum_entries;                                         // [parts_cap]
	uint32_t *chdr = (uint32_t*)(lds_raw + (((size_t)(best_k + parts_cap - lds_raw) + 3u) & ~(size_t)3u));                              // [cg] candidate headers of the current group
	int32_t *cq = (int32_t*)(chdr + cg_arg);                                       // [cg][HIST] coefficients (orders <= HIST)
	uint32_t *cinfo = (uint32_t*)(cq + (size_t)cg_arg * HIST);                     // [cg] order | maxpo_c<<8 | minpo_c<<12 | narrow<<16
	// The register windows hold the samples plus this constant where that keeps every sample positive and below 2^31
	// (lpc_biased_pred_each); fixed predictors of order >= 1 are differences and do not see it
	const uint32_t bias = (bps <= 26u && !(c.debug & 0x4000u)) ? 0x40000000u : 0u;

	// ---- stage the signal: xs[-32..-1] = 0, xs[0..n) = signal >> wasted, zero up to the next chunk boundary
	if(tid == 0) { sh.have_best = 0; sh.best_est = 0; sh.best_cand = 0xffffffffu; sh.best_po = 0; sh.rice2 = 0; }
	if(tid < 32) xs[xs_idx((int32_t)tid - 32)] = 0;
	{
		const uint32_t nup = (n + CH - 1u) & ~(CH - 1u);
		for(uint32_t i = tid; i < nup; i += 256) xs[xs_idx((int32_t)i)] = (i < n) ? (fa_signal(pcm, c, s, base + i) >> wasted) : 0;
	}
	__syncthreads();

	const uint32_t qrows = c.max_lpc_order ? c.max_lpc_order : 1u;

	for(uint32_t cb = 0; cb < ncand; cb += cg) {
		const uint32_t gn = (ncand - cb) < cg ? (ncand - cb) : cg;
		if(wprice) { for(uint32_t t = tid; t < gn * SP_ROW; t += 256) psum[t] = 0; }
		else for(uint32_t t = tid; t < gn * parts0; t += 256) psum[(size_t)(t / parts0) * prow + (t % parts0)] = 0;
		for(uint32_t t = tid; t < gn * 16u; t += 256) lbits[t] = 0;
		// candidate headers and coefficients of this group: one parallel fetch into LDS
		for(uint32_t t = tid; t < gn; t += 256) {
			const size_t rec = fa_cand_idx(c, g, (cb + t));
			const uint32_t hdr = w.cand_hdr[rec], shift = FA_CAND_SHIFT(hdr);
			chdr[t] = hdr;
			// What the multiply-add phase starts a linear predictor's sums from (it reads this before the next barrier): the
			// constants of lpc_biased_pred_each or lpc_offset_diff_each less the bias's share of every tap, bias times the sum
			// of the coefficients, which the model stage left behind them
			const long long qs = (long long)w.cand_q[rec + qrows], b64 = (long long)bias;
			cinit[t] = (bias && FA_CAND_BOUNDED(hdr)) ? (b64 << shift) - b64 * qs : (long long)(1ull << (31u + shift)) - b64 * (qs - (1ll << shift));
		}
		for(uint32_t t = tid; t < gn * HIST; t += 256) {
			const uint32_t ci = t / HIST, j = t - ci * HIST;
			cq[t] = (j < qrows) ? w.cand_q[fa_cand_idx(c, g, (cb + ci)) + j] : 0;
		}
		__syncthreads();
		if(tid < gn) {
			const uint32_t order = FA_CAND_ORDER(chdr[tid]);
			const uint32_t maxpo_c = limit_po(maxpo_f, n, order);
			const uint32_t minpo_c = minpo_f < maxpo_c ? minpo_f : maxpo_c;
			// the reference's accumulator at the candidate's own finest order wraps at 32 bits when
			// ilog2(partition length) + bps < 32 (stream_encoder.c:3808)
			const uint32_t narrow_c = (fa_ilog2(n >> maxpo_c) + bps < 32u) ? 1u : 0u;
			cinfo[tid] = order | (maxpo_c << 8) | (minpo_c << 12) | (narrow_c << 16);
		}

		// ---- |residual| sums per finest partition
		if(fast && !(c.debug & 1u)) {
			const uint32_t cpp = dps0 / CH; // chunks per finest partition
			for(uint32_t chunk = tid; chunk * CH < n; chunk += 256) {
				int32_t xw[HIST + CH];
#pragma unroll
				for(int j = 0; j < HIST + CH; j++) xw[j] = (int32_t)((uint32_t)xs[xs_idx((int32_t)(chunk * CH) - HIST + j)] + bias);
				const uint32_t part = chunk / cpp;
				for(uint32_t ci = 0; ci < gn; ci++) {
					const uint32_t hdr = chdr[ci];
					const uint32_t order = FA_CAND_ORDER(hdr);
					SUM head = 0, tail = 0;
					auto accum = [&](int sidx, int32_t rv, int P) { const uint32_t m = abs_u32(rv); if(sidx < P) head += m; else tail += m; };
					auto accum_fixed = [&](int sidx, int32_t rv, int P) { accum(sidx, P == 0 ? (int32_t)((uint32_t)rv - bias) : rv, P); };
					if(FA_CAND_KIND(hdr) == 0) fixed_each_dispatch(order, xw, accum_fixed);
					else if(order <= HIST) {
						int32_t q[HIST];
#pragma unroll
						for(int j = 0; j < HIST; j++) q[j] = cq[ci * HIST + j];
						const uint32_t shift = FA_CAND_SHIFT(hdr);
						if(!(c.debug & 0x4000u)) {
							// |x - prediction| of every sample by one |a - b| + c.  Where the model stage has bounded the candidate's
							// predictions (below 2^27: sixteen |residual|s cannot carry out of 32 bits) and the window is biased, on the
							// biased sample and prediction; else on the offset difference (exact for any values, see
							// lpc_offset_diff_each), added up in 32 bits for narrow sums - they wrap like the reference's
							// accumulator - and in 64 bits for wide ones.
							const long long init = cinit[ci];
							if(bias && FA_CAND_BOUNDED(hdr)) {
								uint32_t h32 = 0, t32 = 0;
								auto accum_b = [&](int sidx, uint32_t pb, int P) { if(sidx < P) h32 = sad_u32((uint32_t)xw[HIST + sidx], pb, h32); else t32 = sad_u32((uint32_t)xw[HIST + sidx], pb, t32); };
								lpc_biased_pred_each_dispatch(order, xw, q, shift, init, accum_b);
								head = (SUM)h32; tail = (SUM)t32;
							}
							else if(sizeof(SUM) == 4 || FA_CAND_BOUNDED(hdr)) {
								uint32_t h32 = 0, t32 = 0;
								auto accum_u = [&](int sidx, uint32_t db, int P) { if(sidx < P) h32 = sad_u32(db, 0x80000000u, h32); else t32 = sad_u32(db, 0x80000000u, t32); };
								lpc_offset_diff_each_dispatch(order, xw, q, shift, init, accum_u);
								head = (SUM)h32; tail = (SUM)t32;
							}
							else {
								unsigned long long h64 = 0, t64 = 0;
								auto accum_w = [&](int sidx, uint32_t db, int P) {
									const uint32_t m = sad_u32(db, 0x80000000u, 0u);
									if(sidx < P) h64 += (unsigned long long)m; else t64 += (unsigned long long)m;
								};
								lpc_offset_diff_each_dispatch(order, xw, q, shift, init, accum_w);
								head = (SUM)h64; tail = (SUM)t64;
							}
						}
						else if(FA_CAND_WIDE(hdr)) lpc_each_dispatch<MAC_WIDE>(order, xw, q, shift, accum);
						else if(bps <= 24) lpc_each_dispatch<MAC_I24>(order, xw, q, shift, accum);
						else lpc_each_dispatch<MAC_I32>(order, xw, q, shift, accum);
					}
					else continue; // high orders: generic pass below
					const SUM v = tail + (chunk == 0 ? (SUM)0 : head);
					if(sizeof(SUM) == 4) atomicAdd((uint32_t*)&psum[(size_t)ci * prow + part], (uint32_t)v);
					else atomicAdd(&psum[(size_t)ci * prow + part], (unsigned long long)v);
				}
			}
		}
		// generic per-sample pass: any frame length / partition length, and orders above HIST
		for(uint32_t ci = 0; ci < gn; ci++) {
			const uint32_t hdr = chdr[ci];
			const uint32_t order = FA_CAND_ORDER(hdr), kind = FA_CAND_KIND(hdr);
			if(fast && (kind == 0 || order <= HIST)) continue;
			const uint32_t shift = FA_CAND_SHIFT(hdr), wide = FA_CAND_WIDE(hdr);
			for(uint32_t i = tid + order; i < n; i += 256) {
				int32_t r;
				if(kind == 0) {
					const uint32_t a = (uint32_t)xs[xs_idx((int32_t)i)], b = (uint32_t)xs[xs_idx((int32_t)i - 1)], cc = (uint32_t)xs[xs_idx((int32_t)i - 2)],
					               d = (uint32_t)xs[xs_idx((int32_t)i - 3)], e = (uint32_t)xs[xs_idx((int32_t)i - 4)];
					r = (int32_t)(order == 0 ? a : order == 1 ? a - b : order == 2 ? a - 2u*b + cc : order == 3 ? a - 3u*b + 3u*cc - d : a - 4u*b + 6u*cc - 4u*d + e);
				}
				else {
					int32_t pred;
					if(wide) {
						long long acc = 0;
						for(uint32_t j = 0; j < order; j++) acc += (long long)w.cand_q[fa_cand_idx(c, g, (cb + ci)) + j] * (long long)xs[xs_idx((int32_t)(i - 1u - j))];
						pred = (int32_t)(acc >> shift);
					}
					else {
						uint32_t acc = 0;
						for(uint32_t j = 0; j < order; j++) acc += (uint32_t)w.cand_q[fa_cand_idx(c, g, (cb + ci)) + j] * (uint32_t)xs[xs_idx((int32_t)(i - 1u - j))];
						pred = (int32_t)acc >> shift;
					}
					r = (int32_t)((uint32_t)xs[xs_idx((int32_t)i)] - (uint32_t)pred);
				}
				atomicAdd(&psum[(size_t)ci * prow + i / dps0], (unsigned long long)abs_u32(r));
			}
		}
		__syncthreads();

		if(wprice) {
			// ---- eight lanes price a candidate at every partition order; the first wave keeps the best of the group
			// (ordered arg-min: estimate, then candidate index) and works out the Rice parameters of a new leader
			if(!(c.debug & 2u)) sp_price_group(psum, chdr, cinfo, ckey, cpo, gn, cb, tid, 256u, maxpo_f, dps0, wasted, bps, c.k_limit);
			__syncthreads();
			if(tid < 64u && !(c.debug & 2u)) {
				unsigned long long bk = (tid < gn) ? ckey[tid] : ~0ull;
#pragma unroll
				for(int o = 32; o > 0; o >>= 1) { const unsigned long long t = (unsigned long long)__shfl_xor((long long)bk, o, 64); bk = t < bk ? t : bk; }
				const uint32_t est = (uint32_t)(bk >> 32), widx = (uint32_t)bk - cb;
				// earlier groups hold earlier candidates: replace only when strictly smaller
				if(!sh.have_best || est < sh.best_est) {
					const uint32_t po = cpo[widx], lv = maxpo_f - po;
					const uint32_t kw = sp_winner_parameters(psum + (size_t)widx * SP_ROW, cinfo[widx], chdr[widx], tid, maxpo_f, n, po, c.k_limit);
					if(tid < parts0 && (tid & ((1u << lv) - 1u)) == 0u) best_k[tid >> lv] = (uint8_t)kw;
					if(tid == 0) { sh.have_best = 1; sh.best_est = est; sh.best_cand = cb + widx; sh.best_po = po; }
				}
			}
			__syncthreads();
			continue;
		}

		// ---- sums of the coarser partition orders: pairwise merges, three orders per barrier.
		// A value read at the candidate's own finest order is first wrapped to 32 bits when the
		// reference's accumulator there is 32 bits wide.
		for(uint32_t l0 = 0; l0 < maxpo_f; l0 += 3) {
			const uint32_t lto = (l0 + 3u < maxpo_f) ? l0 + 3u : maxpo_f, span = lto - l0;
			const uint32_t cnt_to = parts0 >> lto, fan = 1u << span;
			for(uint32_t t = tid; t < gn * cnt_to; t += 256) {
				const uint32_t ci = t / cnt_to, pg = t - ci * cnt_to;
				const uint32_t info = cinfo[ci];
				const uint32_t lvl_c = maxpo_f - ((info >> 8) & 15u);
				const bool narrow_c = (info >> 16) & 1u;
				unsigned long long *row = psum + (size_t)ci * prow;
				unsigned long long v[8];
				const uint32_t off0 = 2u * parts0 - ((2u * parts0) >> l0);
#pragma unroll
				for(uint32_t i = 0; i < 8; i++) if(i < fan) { v[i] = row[off0 + pg * fan + i]; if(narrow_c && l0 == lvl_c) v[i] &= 0xffffffffull; }
				uint32_t width = fan;
#pragma unroll
				for(uint32_t d = 1; d <= 3; d++) if(d <= span) {
					width >>= 1;
					const uint32_t lv = l0 + d, off = 2u * parts0 - ((2u * parts0) >> lv);
#pragma unroll
					for(uint32_t i = 0; i < 4; i++) if(i < width) {
						unsigned long long m = v[2*i] + v[2*i+1];
						row[off + pg * width + i] = m;
						if(narrow_c && lv == lvl_c) m &= 0xffffffffull;
						v[i] = m;
					}
				}
			}
			__syncthreads();
		}

		// ---- Rice parameter and bit estimate of every (candidate, partition order, partition).
		// Item t = candidate * per + idx: idx in [0, parts0) is order maxpo_f, the next parts0/2 entries order maxpo_f-1, ...
		// so the items of one (candidate, order) are an aligned block of lanes: their estimates are added up inside the
		// wave and one lane per block adds the total to the candidate's row.
		const uint32_t per = 2u * parts0, per_log = maxpo_f + 1u;
		const uint32_t items = (c.debug & 2u) ? 0u : gn * per;
		for(uint32_t t0 = 0; t0 < items; t0 += 256) {
			const uint32_t t = t0 + tid;
			const uint32_t ci = t >> per_log, idx = t & (per - 1u);
			uint32_t est = 0, lvl = 0, seg = 1;
			bool on = false;
			if(t < items && idx != per - 1u) {
				lvl = per_log - (32u - (uint32_t)__clz((int)(per - idx - 1u)));
				seg = per >> (lvl + 1u);
				const uint32_t off = per - (per >> lvl);
				const uint32_t po = maxpo_f - lvl, p = idx - off;
				const uint32_t info = cinfo[ci];
				const uint32_t order = info & 255u, maxpo_c = (info >> 8) & 15u, minpo_c = (info >> 12) & 15u;
				if(po <= maxpo_c && po >= minpo_c) {
					unsigned long long sum = psum[(size_t)ci * prow + idx];
					if(((info >> 16) & 1u) && po == maxpo_c) sum &= 0xffffffffull;
					const uint32_t nsamp = (n >> po) - (p == 0 ? order : 0u);
					uint32_t k;
					if(po == 0) k = FA_CAND_K(chdr[ci]);   // stream_encoder.c:3982-4029
					else {                                  // :4052 smallest k with nsamp << k >= sum
						k = 0;
						if(sum > (unsigned long long)nsamp) {
							const uint32_t bl_s = 64u - (uint32_t)__clzll((long long)sum), bl_n = 32u - (uint32_t)__clz((int)nsamp);
							k = bl_s - bl_n;
							if(((unsigned long long)nsamp << k) < sum) k++;
						}
						if(k >= c.k_limit) k = c.k_limit - 1u;
					}
					// :3929-3937, unsigned 32-bit arithmetic
					est = 4u + (1u + k) * nsamp + (k ? (uint32_t)(sum >> (k - 1u)) : (uint32_t)(sum << 1)) - (nsamp >> 1);
					kbuf[(size_t)ci * 2u * parts_cap + idx] = (uint8_t)k;
					on = true;
				}
			}
			uint32_t v = est, vn;
			vn = block_sum_step<1>(v); v = seg >= 2u ? vn : v;
			vn = block_sum_step<2>(v); v = seg >= 4u ? vn : v;
			vn = block_sum_step<3>(v); v = seg >= 8u ? vn : v;
			vn = block_sum_step<4>(v); v = seg >= 16u ? vn : v;
			vn = block_sum_step<5>(v); v = seg >= 32u ? vn : v;
			vn = block_sum_step<6>(v); v = seg >= 64u ? vn : v;
			const uint32_t lead = (seg < 64u ? seg : 64u) - 1u;
			if(on && (tid & lead) == 0) atomicAdd(&lbits[ci * 16u + lvl], v);
		}
		__syncthreads();

		// ---- best partition order of each candidate (searched max -> min, strict '<'), subframe estimate
		if(tid < gn) {
			const uint32_t ci = tid;
			const uint32_t hdr = chdr[ci], info = cinfo[ci];
			const uint32_t order = info & 255u, maxpo_c = (info >> 8) & 15u, minpo_c = (info >> 12) & 15u;
			uint32_t best = 0, bestpo = 0; bool have = false;
			for(int po = (int)maxpo_c; po >= (int)minpo_c; po--) {
				const uint32_t b = 6u + lbits[ci * 16u + (maxpo_f - (uint32_t)po)];
				if(!have || b < best) { best = b; bestpo = (uint32_t)po; have = true; }
			}
			uint32_t est = 8u + wasted + order * bps + best;                         // stream_encoder.c:3550
			if(FA_CAND_KIND(hdr)) est = 8u + wasted + 9u + order * (FA_CAND_PREC(hdr) + bps) + best; // :3636
			ckey[ci] = ((unsigned long long)est << 32) | (unsigned long long)(cb + ci);
			cpo[ci] = bestpo;
		}
		__syncthreads();
		if(tid == 0) {
			unsigned long long bk = ~0ull; uint32_t bi = 0;
			for(uint32_t ci = 0; ci < gn; ci++) if(ckey[ci] < bk) { bk = ckey[ci]; bi = ci; }
			const uint32_t est = (uint32_t)(bk >> 32);
			// earlier groups hold earlier candidates: replace only when strictly smaller
			sh.group_won = 0;
			if(!sh.have_best || est < sh.best_est) { sh.have_best = 1; sh.best_est = est; sh.best_cand = cb + bi; sh.best_po = cpo[bi]; sh.group_won = 1; }
		}
		__syncthreads();
		if(sh.group_won) {
			const uint32_t ci = sh.best_cand - cb, po = sh.best_po, lvl = maxpo_f - po;
			uint32_t off = 0, cnt = parts0;
			for(uint32_t l = 0; l < lvl; l++) { off += cnt; cnt >>= 1; }
			for(uint32_t p = tid; p < (1u << po); p += 256) best_k[p] = kbuf[(size_t)ci * 2u * parts_cap + off + p];
		}
		__syncthreads();
	}

	// ---- final choice against verbatim / constant (stream_encoder.c:3205-3239, :3393-3396)
	const uint32_t verb_bits = 8u + wasted + n * bps;
	uint32_t base_bits = (c.disable_verbatim && n >= FA_MAX_FIXED_ORDER) ? 0xffffffffu : verb_bits;
	uint32_t type = FA_SUB_VERBATIM, est = base_bits;
	const bool constant = (!c.disable_constant) && n >= FA_MAX_FIXED_ORDER && (w.is_const[g] & 1u);
	if(constant) { const uint32_t cbits = 8u + wasted + bps; if(cbits < est) { type = FA_SUB_CONSTANT; est = cbits; } }
	bool use_cand = false;
	if(sh.have_best && ncand > 0 && sh.best_est < est) { use_cand = true; est = sh.best_est; }
	if(est == 0xffffffffu) { type = FA_SUB_VERBATIM; est = verb_bits; }

	FaWinner *win = &w.winner[g];
	if(use_cand) {
		const uint32_t bc = sh.best_cand;
		const uint32_t hdr = w.cand_hdr[fa_cand_idx(c, g, bc)];
		const uint32_t order = FA_CAND_ORDER(hdr), po = sh.best_po;
		// Rice parameters spread to the finest configured order: entry e belongs to partition e >> (emax - po),
		// so that the pack stage can index them by sample position alone
		const uint32_t emax = c.max_po < FLACAMD_MAX_PARTITION_ORDER ? c.max_po : FLACAMD_MAX_PARTITION_ORDER;
		bool big = false;
		for(uint32_t pe = tid; pe < parts_cap; pe += 256) { const uint8_t k = best_k[pe >> (emax - po)]; win->k[pe] = k; big |= (k >= 15); }
		if(big) atomicOr(&sh.rice2, 1u);
		if(tid < FLACAMD_MAX_LPC_ORDER) win->q[tid] = (FA_CAND_KIND(hdr) && tid < order) ? w.cand_q[fa_cand_idx(c, g, bc) + tid] : 0;
		__syncthreads();
		if(tid == 0) {
			const uint32_t ty = FA_CAND_KIND(hdr) ? FA_SUB_LPC : FA_SUB_FIXED;
			win->hdr[0] = ty | (order << 2) | (FA_CAND_PREC(hdr) << 8) | (FA_CAND_SHIFT(hdr) << 13) | (po << 18) | (sh.rice2 << 22) | (wasted << 23) | (FA_CAND_WIDE(hdr) << 28);
			win->hdr[1] = est; win->hdr[2] = bps; win->hdr[3] = bc;
		}
	}
	else if(tid == 0) {
		win->hdr[0] = type | (wasted << 23);
		win->hdr[1] = est; win->hdr[2] = bps; win->hdr[3] = 0xffffffffu;
	}
}

extern "C" size_t fa_search_lds_bytes(const FaCfg *c, uint32_t cg)
{
	const size_t xs_words = c->spill ? 0u : xs_words_for(c->N);
	const size_t parts_cap = (size_t)1 << (c->max_po < FLACAMD_MAX_PARTITION_ORDER ? c->max_po : FLACAMD_MAX_PARTITION_ORDER);
	const size_t entries = search_psum_entries(cg, (uint32_t)parts_cap);
	size_t b = (xs_words * 4u + 15u) & ~(size_t)15u;
	b += entries * 8u;                 // psum: SP_ROW sums per candidate, or those of all partition orders for fewer candidates
	b += (size_t)cg * 16u * 4u;        // lbits
	b += (size_t)cg * 8u;              // ckey
	b += (size_t)cg * 4u;              // cpo
	b += entries;                      // kbuf
	b += parts_cap;                    // best_k
	b = (b + 3u) & ~(size_t)3u;
	b += (size_t)cg * 4u;              // chdr
	b += (size_t)cg * HIST * 4u;       // cq
	b += (size_t)cg * 4u;              // cinfo
	b += (size_t)cg * 8u;              // cinit
	return (b + 15u) & ~(size_t)15u;
}

extern "C" int fa_search_fast_ok(const void *pcm, const FaCfg *c);
extern "C" void fa_launch_search_fast(const void *pcm, const FaCfg *c, const FaWork *w, uint32_t cg, int sum64, uint32_t frames, hipStream_t st);

// st_rest: where the general kernel goes.  In a single-stream batch it takes only what the register-resident kernel leaves
// (a short last frame): a workgroup or four that would otherwise sit alone on the chip for their whole run, so the caller
// may give them a stream of their own beside the other kernel (and joins it before the pack stage).
extern "C" void fa_launch_search2(const void *pcm, const FaCfg *c, const FaWork *w, uint32_t cg, int sum64, hipStream_t st, hipStream_t st_rest);
extern "C" void fa_launch_search(const void *pcm, const FaCfg *c, const FaWork *w, uint32_t cg, int sum64, hipStream_t st) { fa_launch_search2(pcm, c, w, cg, sum64, st, st); }
// frames the general kernel has to take although the register-resident one runs (0: none, or it runs alone)
extern "C" uint32_t fa_search_rest_frames(const void *pcm, const FaCfg *c)
{
	if(!fa_search_fast_ok(pcm, c) || c->multi) return 0u;
	return c->last_n == c->N ? 0u : (c->F > 1u ? 1u : 0u);
}
extern "C" void fa_launch_search2(const void *pcm, const FaCfg *c, const FaWork *w, uint32_t cg, int sum64, hipStream_t st, hipStream_t st_rest)
{
	// full-length frames of common streams go to the register-resident kernel; a short last
	// frame and everything else to the general one
	uint32_t fast_frames = 0;
	FaCfg cc = *c;
	if(fa_search_fast_ok(pcm, c)) fast_frames = c->multi ? c->F : (c->last_n == c->N) ? c->F : c->F - 1u;
	fa_launch_search_fast(pcm, c, w, cg, sum64, fast_frames, st);
	// several streams in one batch: short final blocks sit anywhere, both kernels cover all frames and each skips the other's
	cc.skip_full = (c->multi && fast_frames) ? 1u : 0u;
	const uint32_t g0 = c->multi ? 0u : fast_frames * c->nsig, rest = cc.skip_full ? c->nshort * c->nsig : c->G - g0;
	if(rest == 0) return;
	c = &cc;
	const size_t lds = fa_search_lds_bytes(c, cg);
	auto launch = [&](auto kernel) {
		(void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		hipLaunchKernelGGL(kernel, dim3(rest), dim3(256), lds, fast_frames ? st_rest : st, pcm, *c, *w, cg, g0);
	};
	if(c->spill) { if(sum64) launch(fa_search_kernel<unsigned long long, true>); else launch(fa_search_kernel<uint32_t, true>); }
	else { if(sum64) launch(fa_search_kernel<unsigned long long, false>); else launch(fa_search_kernel<uint32_t, false>); }
}
