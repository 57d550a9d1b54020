// Pricing of a group of residual-coding candidates from their finest-partition |residual| sums, by lanes of a wave and
// without barriers or atomics: the evaluation phase of the general search kernel for blocks whose finest partitions
// number at most 64 (partition orders up to 6).
//
// Replaces, per candidate: the merge loop of precompute_partition_info_sums_ (stream_encoder.c:3829-3846),
// set_partitioned_rice_ / count_rice_bits_in_partition_ (:3923-4111) for every partition order and the order loop of
// find_best_partition_order_ (:3671-3766).
#pragma once
#include "engine_types.h"
#include "wave_ops.h"

#define SP_ROW 64u       // entries of a candidate's row of finest-partition sums

// Rice parameter of a partition of nsamp residuals with |residual| sum `sum` (stream_encoder.c:4052: the smallest k
// with nsamp << k >= sum, clipped) and the bits it costs (:3929-3937, unsigned 32-bit arithmetic)
__device__ __forceinline__ void sp_price(unsigned long long sum, uint32_t nsamp, bool use_given_k, uint32_t given_k, uint32_t k_limit, uint32_t &k, uint32_t &est)
{
	k = 0;
	if(sum > (unsigned long long)nsamp) {
		const uint32_t bl_s = 64u - (uint32_t)__clzll((long long)sum), bl_n = 32u - (uint32_t)__clz((int)nsamp);
		k = bl_s - bl_n;
		if(((unsigned long long)nsamp << k) < sum) k++;
	}
	if(k >= k_limit) k = k_limit - 1u;
	if(use_given_k) k = given_k;
	est = 4u + (1u + k) * nsamp + (k ? (uint32_t)(sum >> (k - 1u)) : (uint32_t)(sum << 1)) - (nsamp >> 1);
}

// Eight threads per candidate, thread j of a group taking the finest partitions 8j .. 8j + 7 of its candidate's row: it
// adds them up to the 4 + 2 + 1 partitions of the next three orders that lie inside its span and prices all fifteen.
// The partitions of the three coarsest orders (4 + 2 + 1 more) span several threads: their sums come from three
// exchange steps inside the group (lane ^ 1, lane ^ 2, lane <-> 7 - lane) and are priced by the lanes {1, 3, 4, 6},
// {2, 5} and {7} as the chain passes their level - every pair and every half of the group holds exactly one of these
// lanes.  Bits per order: plain sums over the group for the four inner levels; for the outer three the same exchange
// steps taken in the opposite order walk lane 0 through the totals of level 6, 6 + 5, 6 + 5 + 4.  Lane 0 of the group
// then picks the best order (searched max -> min, strict '<') and leaves (estimate << 32 | candidate index) in ckey and
// the order in cpo; the Rice parameters are not kept (sp_winner_parameters works them out for the one that wins).
//
// psum: rows of SP_ROW sums, zero beyond the frame's parts0 finest partitions.  cinfo: order | maxpo_c << 8 |
// minpo_c << 12 | narrow << 16 per candidate.  Every thread of the workgroup calls this (nthreads a multiple of 64).
__device__ __forceinline__ void sp_price_group(const unsigned long long *psum, const uint32_t *chdr, const uint32_t *cinfo, unsigned long long *ckey, uint32_t *cpo,
                                               uint32_t gn, uint32_t cb, uint32_t tid, uint32_t nthreads, uint32_t maxpo_f, uint32_t dps0, uint32_t wasted, uint32_t bps, uint32_t k_limit)
{
	const uint32_t parts0 = 1u << maxpo_f;
	const uint32_t j = tid & 7u;
	const uint32_t lvU = (0x64544540u >> (4u * j)) & 15u;                 // level of this lane's outer partition (lane 0: none)
	const bool firstU = j == 1u || j == 2u || j == 7u;                     // ... and whether it is the first partition of that level
	const bool in_frame = j * 8u < parts0;                                 // the span holds partitions of this frame
	const int32_t room = (int32_t)parts0 - (int32_t)(j * 8u);             // finest partitions of the frame from the span's start on
	for(uint32_t ci = tid >> 3; ci < gn; ci += nthreads / 8u) {
		const uint32_t hdr = chdr[ci], info = cinfo[ci];
		const uint32_t order = info & 255u, maxpo_c = (info >> 8) & 15u, minpo_c = (info >> 12) & 15u;
		const bool narrow_c = (info >> 16) & 1u;
		const uint32_t lvl_c = maxpo_f - maxpo_c;
		const uint32_t given_k = FA_CAND_K(hdr);
		const unsigned long long *row = psum + (size_t)ci * SP_ROW + j * 8u;
		unsigned long long f0[8], f1[4], f2[2], f3;
#pragma unroll
		for(int i = 0; i < 4; i++) { const ulonglong2 v = ((const ulonglong2*)row)[i]; f0[2*i] = v.x; f0[2*i+1] = v.y; }
		// a value at the candidate's own finest order is wrapped to 32 bits when the reference's accumulator there is
		// 32 bits wide (stream_encoder.c:3808); coarser orders add the wrapped values
#define SP_WRAP(L, v) do { if(narrow_c && lvl_c == (uint32_t)(L)) (v) &= 0xffffffffull; } while(0)
#pragma unroll
		for(int i = 0; i < 8; i++) SP_WRAP(0, f0[i]);
#pragma unroll
		for(int i = 0; i < 4; i++) { f1[i] = f0[2*i] + f0[2*i+1]; SP_WRAP(1, f1[i]); }
#pragma unroll
		for(int i = 0; i < 2; i++) { f2[i] = f1[2*i] + f1[2*i+1]; SP_WRAP(2, f2[i]); }
		f3 = f2[0] + f2[1]; SP_WRAP(3, f3);
		unsigned long long u = block_sum_step<1>(f3), vU; SP_WRAP(4, u); vU = u;
		u = block_sum_step<2>(u); SP_WRAP(5, u); if(lvU == 5u) vU = u;
		u = block_sum_step<3>(u); SP_WRAP(6, u); if(lvU == 6u) vU = u;
#undef SP_WRAP
		// bits of a partition of level L (partition order maxpo_f - L); the first partition of a level is `order` samples
		// short, and order 0 takes the suggested parameter (stream_encoder.c:3982-4029)
		auto bits = [&](unsigned long long sum, uint32_t L, bool first) -> uint32_t {
			uint32_t k, est;
			sp_price(sum, (dps0 << L) - (first ? order : 0u), first && L == maxpo_f, given_k, k_limit, k, est);
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
		}
		else {
			// fewer than eight finest partitions: the span reaches beyond the frame
#pragma unroll
			for(int i = 0; i < 8; i++) { const uint32_t e = bits(f0[i], 0u, i == 0 && j == 0u); e0 += (i < room) ? e : 0u; }
#pragma unroll
			for(int i = 0; i < 4; i++) { const uint32_t e = bits(f1[i], 1u, i == 0 && j == 0u); e1 += (2 * i < room) ? e : 0u; }
#pragma unroll
			for(int i = 0; i < 2; i++) { const uint32_t e = bits(f2[i], 2u, i == 0 && j == 0u); e2 += (4 * i < room) ? e : 0u; }
		}
		e3 = bits(f3, 3u, j == 0u);
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
		uint32_t est = 8u + wasted + order * bps + best;                         // stream_encoder.c:3550
		if(FA_CAND_KIND(hdr)) est = 8u + wasted + 9u + order * (FA_CAND_PREC(hdr) + bps) + best; // :3636
		if(j == 0u) { ckey[ci] = ((unsigned long long)est << 32) | (unsigned long long)(cb + ci); cpo[ci] = maxpo_f - bestlv; }
	}
}

// Rice parameters of a candidate at partition order best_po, by the 64 lanes of one wave: lane p starts from finest
// partition p of the candidate's row, best_lv doubling steps leave every lane with the sum of the partition it lies in.
// Returns the parameter of the partition that finest partition `lane` lies in.
__device__ __forceinline__ uint32_t sp_winner_parameters(const unsigned long long *row, uint32_t info, uint32_t hdr, uint32_t lane, uint32_t maxpo_f, uint32_t n, uint32_t best_po, uint32_t k_limit)
{
	const uint32_t parts0 = 1u << maxpo_f;
	const uint32_t order = info & 255u, maxpo_c = (info >> 8) & 15u;
	const bool narrow_c = (info >> 16) & 1u;
	const uint32_t lvl_c = maxpo_f - maxpo_c, best_lv = maxpo_f - best_po;
	unsigned long long v = (lane < parts0) ? row[lane] : 0ull;
	if(narrow_c && lvl_c == 0u) v &= 0xffffffffull;
#define SP_CHAIN(L) do { if(best_lv >= (uint32_t)(L)) { v = block_sum_step<L>(v); if(narrow_c && lvl_c == (uint32_t)(L)) v &= 0xffffffffull; } } while(0)
	SP_CHAIN(1); SP_CHAIN(2); SP_CHAIN(3); SP_CHAIN(4); SP_CHAIN(5); SP_CHAIN(6);
#undef SP_CHAIN
	uint32_t kw, estw;
	sp_price(v, (n >> best_po) - ((lane >> best_lv) == 0u ? order : 0u), best_po == 0u, FA_CAND_K(hdr), k_limit, kw, estw);
	return kw;
}
