// Cross-lane helpers for 64-lane wavefronts on gfx950.  Everything here stays in the vector
// ALU (DPP moves within a row of 16 lanes, v_permlane16_swap / v_permlane32_swap across rows):
// no trip through the LDS crossbar, so a dependent chain of steps costs ALU latency only.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

// value of the lane paired with this one at exchange distance OFF (1, 2, 4, 8 within a row).
// OFF 1, 2, 8 pair lane ^ OFF; OFF 4 pairs lane i with lane 7 - i of its group of 8 (the two
// differ in bit 2, which is all the callers need).
template<int OFF>
__device__ __forceinline__ uint32_t row_partner(uint32_t v)
{
	static_assert(OFF == 1 || OFF == 2 || OFF == 4 || OFF == 8, "row distance");
	if(OFF == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);    // quad_perm [1,0,3,2]
	if(OFF == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);    // quad_perm [2,3,0,1]
	if(OFF == 4) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false);   // row_half_mirror
	return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, false);                // row_ror:8
}

// One doubling step of an aligned-block sum: on entry every lane holds the sum of its aligned
// block of 2^(L-1) lanes (identical within the block), on return the sum of its block of 2^L.
template<int L>
__device__ __forceinline__ uint32_t block_sum_step(uint32_t v)
{
	static_assert(L >= 1 && L <= 6, "level");
	if(L == 1) return v + row_partner<1>(v);
	if(L == 2) return v + row_partner<2>(v);
	if(L == 3) return v + row_partner<4>(v);
	if(L == 4) return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false);   // row_mirror: the other 8 lanes of the row
	if(L == 5) { const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false); return r[0] + r[1]; }   // [r0,r0,r2,r2] + [r1,r1,r3,r3]
	const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false); return r[0] + r[1];             // [lo,lo] + [hi,hi]
}
template<int L>
__device__ __forceinline__ unsigned long long block_sum_step(unsigned long long v)
{
	uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32), plo, phi;
	if(L == 1) { plo = row_partner<1>(lo); phi = row_partner<1>(hi); }
	else if(L == 2) { plo = row_partner<2>(lo); phi = row_partner<2>(hi); }
	else if(L == 3) { plo = row_partner<4>(lo); phi = row_partner<4>(hi); }
	else if(L == 4) { plo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, 0x140, 0xF, 0xF, false); phi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hi, 0x140, 0xF, 0xF, false); }
	else if(L == 5) {
		const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false), b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
		return (((unsigned long long)b[0] << 32) | a[0]) + (((unsigned long long)b[1] << 32) | a[1]);
	}
	else {
		const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false), b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
		return (((unsigned long long)b[0] << 32) | a[0]) + (((unsigned long long)b[1] << 32) | a[1]);
	}
	return v + (((unsigned long long)phi << 32) | plo);
}

// One step of a butterfly that gathers residue classes: v plus the value of a lane D lanes away
// (a lane of the same class modulo D; every lane of that class holds the same value by then).  Applied
// for D = 32, 16, 8, 4, 2, 1 in that order, lane L holds after the step for D the sum over all lanes
// congruent to L modulo D; lane 0 therefore runs through the sums of the multiples of 32, 16, ... 1.
template<int D>
__device__ __forceinline__ uint32_t class_sum_step(uint32_t v)
{
	static_assert(D == 1 || D == 2 || D == 4 || D == 8 || D == 16 || D == 32, "distance");
	if(D == 32) { const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false); return r[0] + r[1]; }
	if(D == 16) { const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false); return r[0] + r[1]; }
	if(D == 8) return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, false);   // row_ror:8
	if(D == 4) return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xF, 0xF, false);   // row_ror:4
	if(D == 2) return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);    // quad_perm [2,3,0,1]
	return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);               // quad_perm [1,0,3,2]
}

// Sum (or OR) NV = 2^m values across the 64 lanes of a wave with a halving butterfly: every
// exchange step halves the number of values a lane still carries (about NV exchanges instead
// of 6 * NV).  Afterwards lane l holds the total of value (l >> (6 - m)) in v[0].
// Lanes whose bit `off` is clear keep the lower half of their values and take the partner's
// lower half; the others keep and take the upper half.
template<int NV, bool IS_OR>
__device__ __forceinline__ void wave_combine(uint32_t (&v)[NV], uint32_t lane)
{
#define WC_OP(a, b) (IS_OR ? ((a) | (b)) : ((a) + (b)))
	int n = NV;
	// distance 32 and 16: one swap instruction moves both halves at once
	if(n >= 2) {
		const int half = n / 2;
#pragma unroll
		for(int i = 0; i < half; i++) { const auto r = __builtin_amdgcn_permlane32_swap(v[i], v[i + half], false, false); v[i] = WC_OP(r[0], r[1]); }
		n = half;
	}
	else { const auto r = __builtin_amdgcn_permlane32_swap(v[0], v[0], false, false); v[0] = WC_OP(r[0], r[1]); }
	if(n >= 2) {
		const int half = n / 2;
#pragma unroll
		for(int i = 0; i < half; i++) { const auto r = __builtin_amdgcn_permlane16_swap(v[i], v[i + half], false, false); v[i] = WC_OP(r[0], r[1]); }
		n = half;
	}
	else { const auto r = __builtin_amdgcn_permlane16_swap(v[0], v[0], false, false); v[0] = WC_OP(r[0], r[1]); }
	// distances 8, 4, 2, 1: DPP move of the half that is given away.  The selects are bitwise
	// (one v_bfi each): a ?: on two array elements may be turned into an indexed access, which
	// would move the whole array to scratch memory.
#define WC_ROW_STEP(OFF) do { \
		if(n >= 2) { \
			const int half = n / 2; \
			const uint32_t hm = (lane & (uint32_t)(OFF)) ? 0xffffffffu : 0u; \
			_Pragma("unroll") \
			for(int i = 0; i < half; i++) { \
				const uint32_t send = (v[i] & hm) | (v[i + half] & ~hm), keep = (v[i + half] & hm) | (v[i] & ~hm); \
				v[i] = WC_OP(keep, row_partner<OFF>(send)); \
			} \
			n = half; \
		} \
		else v[0] = WC_OP(v[0], row_partner<OFF>(v[0])); \
	} while(0)
	WC_ROW_STEP(8);
	WC_ROW_STEP(4);
	WC_ROW_STEP(2);
	WC_ROW_STEP(1);
#undef WC_ROW_STEP
#undef WC_OP
}
