// Stage 4 of the encoder: channel-assignment decision and bitstream packing of a frame,
// then stage 5: gathering the frames into one contiguous byte stream.
//
// Replaces, for one frame:
//   the assignment arg-min and loose mid/side rule of process_subframes_   stream_encoder.c:2957-2976, :3052-3087
//   FLAC__frame_add_header                                                stream_encoder_framing.c:219-364
//   FLAC__subframe_add_{constant,fixed,lpc,verbatim}, add_residual_partitioned_rice_  :366-553
//   FLAC__bitwriter_write_rice_signed_block / _write_raw_* / _write_utf8_uint32 / zero padding  bitwriter.c:305-380, :544-658, :784-892
//   FLAC__crc8 / FLAC__crc16 over the written bytes                       crc.c:124-142, stream_encoder.c:2890-2905
//
// One 256-thread workgroup per frame.  A subframe is a run of variable-length tokens
// (Rice codes: q zero bits, a one, k low bits).  Each thread owns a contiguous range of
// samples: pass 1 computes the bit length of its range, a workgroup exclusive scan turns
// lengths into bit offsets, pass 2 ORs every token into an LDS bit buffer at its offset
// (zero runs need no writes).  Completed 32-bit words are flushed big-endian to the frame's
// slot in HBM with coalesced stores; the partial last word carries into the next subframe.
// CRC-16 is computed in parallel over the finished bytes (per-thread segment CRC, then a
// GF(2) multiply by x^(8*bytes_after) and an XOR reduction).
#include "engine_types.h"

__device__ __forceinline__ uint32_t pk_xs_idx(int32_t i) { return (uint32_t)(i + 32); }

// ---- CRC helpers (polynomials of crc.c:33-121: CRC-8 0x07, CRC-16 0x8005, init 0, MSB first)
__device__ __forceinline__ uint32_t crc16_byte(uint32_t crc, uint32_t byte)
{
	crc ^= byte << 8;
#pragma unroll
	for(int b = 0; b < 8; b++) crc = (crc & 0x8000u) ? ((crc << 1) ^ 0x8005u) & 0xffffu : (crc << 1) & 0xffffu;
	return crc;
}
__device__ __forceinline__ uint32_t crc8_byte(uint32_t crc, uint32_t byte)
{
	crc ^= byte;
#pragma unroll
	for(int b = 0; b < 8; b++) crc = (crc & 0x80u) ? ((crc << 1) ^ 0x07u) & 0xffu : (crc << 1) & 0xffu;
	return crc;
}
// a(x) * b(x) mod (x^16 + x^15 + x^2 + 1)
__device__ __forceinline__ uint32_t gf16_mul(uint32_t a, uint32_t b)
{
	uint32_t r = 0;
#pragma unroll
	for(int i = 15; i >= 0; i--) {
		r = (r & 0x8000u) ? ((r << 1) ^ 0x8005u) & 0xffffu : (r << 1) & 0xffffu;
		if((b >> i) & 1u) r ^= a;
	}
	return r;
}
// x^(8*nbytes) mod P by square and multiply; x^8 = 0x0100
__device__ __forceinline__ uint32_t gf16_xpow8(uint32_t nbytes)
{
	uint32_t result = 1u, base = 0x0100u;
	while(nbytes) { if(nbytes & 1u) result = gf16_mul(result, base); base = gf16_mul(base, base); nbytes >>= 1; }
	return result;
}

struct PackShared {
	uint32_t wave_tot[4];
	uint32_t assignment;
	uint32_t carry_bits;   // valid bits in bitbuf[0] carried from what was written before
	uint32_t words_out;    // 32-bit words already flushed to the slot
	uint32_t overflow;
	uint32_t crc;
	int32_t  q[FLACAMD_MAX_LPC_ORDER];
	uint8_t  k[FA_MAX_PARTS];
};

// OR `nbits` (1..32) low bits of v into the MSB-first bit buffer at bit position pos
__device__ __forceinline__ void put_bits(uint32_t *buf, uint32_t pos, uint32_t v, uint32_t nbits)
{
	const uint32_t wi = pos >> 5, off = pos & 31u;
	if(nbits < 32u) v &= (1u << nbits) - 1u;
	if(off + nbits <= 32u) atomicOr(&buf[wi], v << (32u - off - nbits));
	else {
		const uint32_t spill = off + nbits - 32u;
		atomicOr(&buf[wi], v >> spill);
		atomicOr(&buf[wi + 1], v << (32u - spill));
	}
}

// exclusive scan of one value per thread over the 256-thread workgroup; returns the
// exclusive prefix and the total through `total`
__device__ __forceinline__ uint32_t block_exscan(uint32_t v, PackShared &sh, uint32_t &total)
{
	const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
	uint32_t inc = v;
#pragma unroll
	for(int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)inc, o, 64); if(lane >= (uint32_t)o) inc += t; }
	__syncthreads();
	if(lane == 63u) sh.wave_tot[wv] = inc;
	__syncthreads();
	uint32_t pre = 0;
#pragma unroll
	for(uint32_t i = 0; i < 4; i++) if(i < wv) pre += sh.wave_tot[i];
	total = sh.wave_tot[0] + sh.wave_tot[1] + sh.wave_tot[2] + sh.wave_tot[3];
	return pre + inc - v;
}

// residual of sample i (i >= order) of the staged signal
__device__ __forceinline__ int32_t residual_at(const int32_t *xs, uint32_t i, uint32_t type, uint32_t order, uint32_t shift, uint32_t wide, const int32_t *q)
{
	if(type == FA_SUB_FIXED) {
		const uint32_t a = (uint32_t)xs[pk_xs_idx((int32_t)i)], b = (uint32_t)xs[pk_xs_idx((int32_t)i - 1)], c = (uint32_t)xs[pk_xs_idx((int32_t)i - 2)],
		               d = (uint32_t)xs[pk_xs_idx((int32_t)i - 3)], e = (uint32_t)xs[pk_xs_idx((int32_t)i - 4)];
		return (int32_t)(order == 0 ? a : order == 1 ? a - b : order == 2 ? a - 2u*b + c : order == 3 ? a - 3u*b + 3u*c - d : a - 4u*b + 6u*c - 4u*d + e);
	}
	int32_t pred;
	if(wide) {
		long long acc = 0;
		for(uint32_t j = 0; j < order; j++) acc += (long long)q[j] * (long long)xs[pk_xs_idx((int32_t)(i - 1u - j))];
		pred = (int32_t)(acc >> shift);
	}
	else {
		uint32_t acc = 0;
		for(uint32_t j = 0; j < order; j++) acc += (uint32_t)q[j] * (uint32_t)xs[pk_xs_idx((int32_t)(i - 1u - j))];
		pred = (int32_t)acc >> shift;
	}
	return (int32_t)((uint32_t)xs[pk_xs_idx((int32_t)i)] - (uint32_t)pred);
}

// Flush the completed words of the bit buffer (bits [0, total_bits)) to the slot and keep
// the partial word as the new carry.  All threads call it.
__device__ __forceinline__ void flush_words(uint32_t *bitbuf, uint32_t total_bits, uint32_t *slot_words, PackShared &sh)
{
	const uint32_t full = total_bits >> 5, wo = sh.words_out;
	__syncthreads();
	for(uint32_t i = threadIdx.x; i < full; i += 256) slot_words[wo + i] = __builtin_bswap32(bitbuf[i]);
	const uint32_t partial = bitbuf[full];
	__syncthreads();
	for(uint32_t i = threadIdx.x; i <= full + 1u; i += 256) bitbuf[i] = 0;
	__syncthreads();
	if(threadIdx.x == 0) { bitbuf[0] = partial; sh.carry_bits = total_bits & 31u; sh.words_out = wo + full; }
	__syncthreads();
}

__global__ void __launch_bounds__(256) fa_pack_kernel(const void *__restrict__ pcm, FaCfg c, FaWork w, uint32_t bitbuf_words)
{
	extern __shared__ __align__(16) unsigned char lds_raw[];
	__shared__ PackShared sh;
	int32_t *xs = (int32_t*)lds_raw;                       // [N + 32]
	uint32_t *bitbuf = (uint32_t*)(xs + c.N + 32u);        // [bitbuf_words]

	const uint32_t f = blockIdx.x, tid = threadIdx.x;
	const uint32_t n = fa_frame_len(c, f);
	const uint64_t base = (uint64_t)f * c.N;
	uint32_t *slot_words = (uint32_t*)(w.slots + (size_t)f * c.slot_bytes);
	uint8_t *slot = w.slots + (size_t)f * c.slot_bytes;
	const uint32_t cap_bits = (bitbuf_words - 2u) * 32u;

	for(uint32_t i = tid; i < bitbuf_words; i += 256) bitbuf[i] = 0;
	if(tid < 32) xs[tid] = 0;
	__syncthreads();

	// ---- channel assignment (stream_encoder.c:3052-3087; loose rule :2958-2967, :3059-3061)
	if(tid == 0) {
		uint32_t a = 0;
		if(c.midside) {
			const uint32_t fa = c.first_frame + f;
			const uint32_t cnt = c.loose ? fa % c.loose_period : 0u;
			// frame whose four estimates decide: this one, or the first of its decision period
			uint32_t decide_ok = 1, fd = f;
			if(cnt != 0) { if(cnt <= f) fd = f - cnt; else decide_ok = 0; }
			if(decide_ok) {
				const FaWinner *wn = &w.winner[(size_t)fd * 4u];
				const uint32_t b0 = wn[0].hdr[1], b1 = wn[1].hdr[1], b2 = wn[2].hdr[1], b3 = wn[3].hdr[1];
				const uint32_t bits[4] = { b0 + b1, b0 + b3, b1 + b3, b2 + b3 };
				uint32_t m = bits[0];
				for(uint32_t i = 1; i < 4; i++) if(bits[i] < m) { m = bits[i]; a = i; }
			}
			else a = c.prev_assignment;
			if(cnt != 0) a = (a == 0) ? 0u : 3u; // followers only choose between independent and mid/side
		}
		sh.assignment = a; sh.carry_bits = 0; sh.words_out = 0; sh.overflow = 0; sh.crc = 0;
		w.assignment[f] = a;

		// ---- frame header (stream_encoder_framing.c:219-364), bytes built serially by one lane
		uint8_t hb[16]; uint32_t nb = 0;
		uint32_t bs_code, bs_hint = 0, sr_code, sr_hint = 0;
		switch(n) {
		case 192: bs_code = 1; break; case 576: bs_code = 2; break; case 1152: bs_code = 3; break; case 2304: bs_code = 4; break; case 4608: bs_code = 5; break;
		case 256: bs_code = 8; break; case 512: bs_code = 9; break; case 1024: bs_code = 10; break; case 2048: bs_code = 11; break; case 4096: bs_code = 12; break;
		case 8192: bs_code = 13; break; case 16384: bs_code = 14; break; case 32768: bs_code = 15; break;
		default: bs_code = bs_hint = (n <= 0x100u) ? 6u : 7u; break;
		}
		switch(c.rate) {
		case 88200: sr_code = 1; break; case 176400: sr_code = 2; break; case 192000: sr_code = 3; break; case 8000: sr_code = 4; break; case 16000: sr_code = 5; break;
		case 22050: sr_code = 6; break; case 24000: sr_code = 7; break; case 32000: sr_code = 8; break; case 44100: sr_code = 9; break; case 48000: sr_code = 10; break; case 96000: sr_code = 11; break;
		default:
			if(c.rate <= 255000u && c.rate % 1000u == 0) sr_code = sr_hint = 12;
			else if(c.rate % 10u == 0) sr_code = sr_hint = 14;
			else if(c.rate <= 0xffffu) sr_code = sr_hint = 13;
			else sr_code = 0;
			break;
		}
		uint32_t bps_code;
		switch(c.bps) { case 8: bps_code = 1; break; case 12: bps_code = 2; break; case 16: bps_code = 4; break; case 20: bps_code = 5; break; case 24: bps_code = 6; break; default: bps_code = 0; break; }
		const uint32_t ch_code = (a == 0) ? c.channels - 1u : 7u + a;
		hb[nb++] = 0xff; hb[nb++] = 0xf8;
		hb[nb++] = (uint8_t)((bs_code << 4) | sr_code);
		hb[nb++] = (uint8_t)((ch_code << 4) | (bps_code << 1));
		{ // bitwriter.c:784-832
			const uint32_t v = c.first_frame + f;
			if(v < 0x80u) hb[nb++] = (uint8_t)v;
			else if(v < 0x800u) { hb[nb++] = (uint8_t)(0xC0u | (v >> 6)); hb[nb++] = (uint8_t)(0x80u | (v & 0x3Fu)); }
			else if(v < 0x10000u) { hb[nb++] = (uint8_t)(0xE0u | (v >> 12)); hb[nb++] = (uint8_t)(0x80u | ((v >> 6) & 0x3Fu)); hb[nb++] = (uint8_t)(0x80u | (v & 0x3Fu)); }
			else if(v < 0x200000u) { hb[nb++] = (uint8_t)(0xF0u | (v >> 18)); hb[nb++] = (uint8_t)(0x80u | ((v >> 12) & 0x3Fu)); hb[nb++] = (uint8_t)(0x80u | ((v >> 6) & 0x3Fu)); hb[nb++] = (uint8_t)(0x80u | (v & 0x3Fu)); }
			else if(v < 0x4000000u) { hb[nb++] = (uint8_t)(0xF8u | (v >> 24)); hb[nb++] = (uint8_t)(0x80u | ((v >> 18) & 0x3Fu)); hb[nb++] = (uint8_t)(0x80u | ((v >> 12) & 0x3Fu)); hb[nb++] = (uint8_t)(0x80u | ((v >> 6) & 0x3Fu)); hb[nb++] = (uint8_t)(0x80u | (v & 0x3Fu)); }
			else { hb[nb++] = (uint8_t)(0xFCu | (v >> 30)); hb[nb++] = (uint8_t)(0x80u | ((v >> 24) & 0x3Fu)); hb[nb++] = (uint8_t)(0x80u | ((v >> 18) & 0x3Fu)); hb[nb++] = (uint8_t)(0x80u | ((v >> 12) & 0x3Fu)); hb[nb++] = (uint8_t)(0x80u | ((v >> 6) & 0x3Fu)); hb[nb++] = (uint8_t)(0x80u | (v & 0x3Fu)); }
		}
		if(bs_hint == 6) hb[nb++] = (uint8_t)(n - 1u);
		else if(bs_hint == 7) { hb[nb++] = (uint8_t)((n - 1u) >> 8); hb[nb++] = (uint8_t)(n - 1u); }
		if(sr_hint == 12) hb[nb++] = (uint8_t)(c.rate / 1000u);
		else if(sr_hint == 13) { hb[nb++] = (uint8_t)(c.rate >> 8); hb[nb++] = (uint8_t)c.rate; }
		else if(sr_hint == 14) { hb[nb++] = (uint8_t)((c.rate / 10u) >> 8); hb[nb++] = (uint8_t)(c.rate / 10u); }
		uint32_t crc = 0;
		for(uint32_t i = 0; i < nb; i++) crc = crc8_byte(crc, hb[i]);
		hb[nb++] = (uint8_t)crc;
		sh.wave_tot[0] = nb; // header length handed to the flush below
		for(uint32_t i = 0; i < nb; i++) bitbuf[i >> 2] |= (uint32_t)hb[i] << (24u - 8u * (i & 3u));
	}
	__syncthreads();
	{
		const uint32_t hdr_bits = sh.wave_tot[0] * 8u;
		__syncthreads();
		flush_words(bitbuf, hdr_bits, slot_words, sh);
	}

	const uint32_t assignment = sh.assignment;
	const uint32_t nsub = c.channels;
	const uint32_t S = (n + 255u) / 256u; // samples per thread

	for(uint32_t sc = 0; sc < nsub; sc++) {
		uint32_t sig = sc;
		if(c.midside) { // stream_encoder.c:3096-3115
			if(assignment == 1) sig = sc == 0 ? 0u : 3u;
			else if(assignment == 2) sig = sc == 0 ? 3u : 1u;
			else if(assignment == 3) sig = sc == 0 ? 2u : 3u;
		}
		const uint32_t g = f * c.nsig + sig;
		const FaWinner *win = &w.winner[g];
		const uint32_t h0 = win->hdr[0], sbps = win->hdr[2];
		const uint32_t type = FA_WIN_TYPE(h0), order = FA_WIN_ORDER(h0), prec = FA_WIN_PREC(h0), shift = FA_WIN_SHIFT(h0), po = FA_WIN_PO(h0),
		               rice2 = FA_WIN_RICE2(h0), wasted = FA_WIN_WASTED(h0), wide = FA_WIN_WIDE(h0);

		// stage the signal and the winner's parameters
		for(uint32_t i = tid; i < n; i += 256) xs[pk_xs_idx((int32_t)i)] = fa_signal(pcm, c, sig, base + i) >> wasted;
		if(tid < FLACAMD_MAX_LPC_ORDER) sh.q[tid] = win->q[tid];
		if(type >= FA_SUB_FIXED) for(uint32_t p = tid; p < (1u << po); p += 256) sh.k[p] = win->k[p];
		__syncthreads();

		const uint32_t carry = sh.carry_bits;
		// bits of the subframe head: type byte, wasted-bits unary, warm-up, lpc parameters, coding method + partition order
		uint32_t head_bits = 8u + (wasted ? wasted : 0u);
		if(type == FA_SUB_FIXED) head_bits += order * sbps + 6u;
		else if(type == FA_SUB_LPC) head_bits += order * sbps + 9u + order * prec + 6u;
		const uint32_t plen = rice2 ? 5u : 4u;
		const uint32_t dps = n >> po;

		const uint32_t i0 = tid * S, i1 = (i0 + S < n) ? i0 + S : n;
		uint32_t mybits = 0;
		if(type == FA_SUB_CONSTANT) mybits = (tid == 0) ? sbps : 0u;
		else if(type == FA_SUB_VERBATIM) mybits = (i1 > i0) ? (i1 - i0) * sbps : 0u;
		else {
			for(uint32_t i = (i0 > order ? i0 : order); i < i1; i++) {
				const uint32_t p = po ? i / dps : 0u, k = sh.k[p];
				const int32_t r = residual_at(xs, i, type, order, shift, wide, sh.q);
				const uint32_t u = ((uint32_t)r << 1) ^ (uint32_t)(r >> 31);
				mybits += (u >> k) + 1u + k;
				if(i == (p * dps > order ? p * dps : order)) mybits += plen; // first residual of its partition
			}
		}
		uint32_t body_bits;
		const uint32_t myoff = block_exscan(mybits, sh, body_bits);
		const uint32_t sub_bits = head_bits + body_bits;
		if(carry + sub_bits > cap_bits) { // cannot happen for estimates that did not wrap; never write past the buffer
			if(tid == 0) { sh.overflow = 1; atomicOr(w.status, 1u); }
			__syncthreads();
			break;
		}

		// ---- write the head (one lane; a few dozen tokens)
		if(tid == 0) {
			uint32_t pos = carry;
			uint32_t tb;
			switch(type) { // stream_encoder_framing.c:371, :383, :421, :468
			case FA_SUB_CONSTANT: tb = 0x00; break; case FA_SUB_VERBATIM: tb = 0x02; break;
			case FA_SUB_FIXED: tb = 0x10u | (order << 1); break; default: tb = 0x40u | ((order - 1u) << 1); break;
			}
			put_bits(bitbuf, pos, tb | (wasted ? 1u : 0u), 8); pos += 8;
			if(wasted) { pos += wasted - 1u; put_bits(bitbuf, pos, 1u, 1); pos += 1; }
			if(type >= FA_SUB_FIXED) {
				for(uint32_t i = 0; i < order; i++) { put_bits(bitbuf, pos, (uint32_t)xs[pk_xs_idx((int32_t)i)], sbps); pos += sbps; }
				if(type == FA_SUB_LPC) {
					put_bits(bitbuf, pos, prec - 1u, 4); pos += 4;
					put_bits(bitbuf, pos, shift, 5); pos += 5;
					for(uint32_t i = 0; i < order; i++) { put_bits(bitbuf, pos, (uint32_t)sh.q[i], prec); pos += prec; }
				}
				put_bits(bitbuf, pos, rice2, 2); pos += 2;
				put_bits(bitbuf, pos, po, 4); pos += 4;
			}
		}
		// ---- write the body
		{
			uint32_t pos = carry + head_bits + myoff;
			if(type == FA_SUB_CONSTANT) { if(tid == 0) put_bits(bitbuf, pos, (uint32_t)xs[pk_xs_idx(0)], sbps); }
			else if(type == FA_SUB_VERBATIM) { for(uint32_t i = i0; i < i1; i++) { put_bits(bitbuf, pos, (uint32_t)xs[pk_xs_idx((int32_t)i)], sbps); pos += sbps; } }
			else {
				for(uint32_t i = (i0 > order ? i0 : order); i < i1; i++) {
					const uint32_t p = po ? i / dps : 0u, k = sh.k[p];
					if(i == (p * dps > order ? p * dps : order)) { put_bits(bitbuf, pos, k, plen); pos += plen; }
					const int32_t r = residual_at(xs, i, type, order, shift, wide, sh.q);
					const uint32_t u = ((uint32_t)r << 1) ^ (uint32_t)(r >> 31);
					pos += u >> k;                                               // unary zeros
					put_bits(bitbuf, pos, (1u << k) | (u & ((1u << k) - 1u)), k + 1u); // stop bit + k low bits
					pos += k + 1u;
				}
			}
		}
		flush_words(bitbuf, carry + sub_bits, slot_words, sh);
	}

	// ---- zero-pad to a byte, CRC-16, footer (stream_encoder.c:2890-2905)
	if(sh.overflow) { if(tid == 0) w.frame_bytes[f] = 0; return; }
	const uint32_t tail_bits = sh.carry_bits, tail_bytes = (tail_bits + 7u) >> 3;
	const uint32_t body_bytes = sh.words_out * 4u + tail_bytes;
	if(tid < tail_bytes) slot[sh.words_out * 4u + tid] = (uint8_t)(bitbuf[0] >> (24u - 8u * tid));
	__syncthreads();
	{
		// each thread: CRC of its segment, then * x^(8 * bytes after the segment)
		const uint32_t seg = ((body_bytes + 255u) / 256u + 3u) & ~3u;
		const uint32_t b0 = tid * seg, b1 = (b0 + seg < body_bytes) ? b0 + seg : body_bytes;
		uint32_t crc = 0;
		if(b0 < body_bytes) {
			for(uint32_t b = b0; b < b1; b++) crc = crc16_byte(crc, slot[b]);
			crc = gf16_mul(crc, gf16_xpow8(body_bytes - b1));
		}
#pragma unroll
		for(int o = 32; o > 0; o >>= 1) crc ^= (uint32_t)__shfl_xor((int)crc, o, 64);
		if((tid & 63u) == 0) atomicXor(&sh.crc, crc);
	}
	__syncthreads();
	if(tid == 0) {
		const uint32_t crc = sh.crc;
		slot[body_bytes] = (uint8_t)(crc >> 8); slot[body_bytes + 1u] = (uint8_t)crc;
		w.frame_bytes[f] = body_bytes + 2u;
	}
}

// ---- stage 5: exclusive scan of the frame sizes, then copy slots to their final offsets

__global__ void __launch_bounds__(1024) fa_scan_kernel(FaCfg c, FaWork w)
{
	__shared__ unsigned long long wave_tot[16];
	__shared__ unsigned long long carry;
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
	if(tid == 0) carry = 0;
	__syncthreads();
	for(uint32_t b = 0; b < c.F; b += 1024) {
		const uint32_t i = b + tid;
		const unsigned long long v = i < c.F ? w.frame_bytes[i] : 0;
		unsigned long long inc = v;
#pragma unroll
		for(int o = 1; o < 64; o <<= 1) { const unsigned long long t = (unsigned long long)__shfl_up((long long)inc, o, 64); if(lane >= (uint32_t)o) inc += t; }
		if(lane == 63u) wave_tot[wv] = inc;
		__syncthreads();
		unsigned long long pre = carry;
		for(uint32_t k = 0; k < wv; k++) pre += wave_tot[k];
		if(i < c.F) w.offsets[i] = pre + inc - v;
		__syncthreads();
		if(tid == 1023u) carry = pre + inc;
		__syncthreads();
	}
	if(tid == 0) w.offsets[c.F] = carry;
}

// One workgroup per frame: byte-granular destination, 4-byte loads where the source allows
__global__ void __launch_bounds__(256) fa_gather_kernel(FaCfg c, FaWork w, uint8_t *__restrict__ out, unsigned long long out_cap, uint32_t *__restrict__ frame_bytes_out)
{
	const uint32_t f = blockIdx.x, tid = threadIdx.x;
	const uint32_t nb = w.frame_bytes[f];
	const unsigned long long off = w.offsets[f];
	if(tid == 0 && frame_bytes_out) frame_bytes_out[f] = nb;
	if(off + nb > out_cap) { if(tid == 0) atomicOr(w.status, 2u); return; }
	const uint8_t *src = w.slots + (size_t)f * c.slot_bytes;
	uint8_t *dst = out + off;
	// align the destination to 4 bytes, then move words assembled from the source
	const uint32_t lead = (uint32_t)((4u - ((uintptr_t)dst & 3u)) & 3u);
	const uint32_t head = lead < nb ? lead : nb;
	if(tid < head) dst[tid] = src[tid];
	const uint32_t nwords = (nb - head) >> 2;
	const uint32_t *sw = (const uint32_t*)src; // slot is 16-byte aligned
	for(uint32_t i = tid; i < nwords; i += 256) {
		const uint32_t sb = head + i * 4u;              // source byte offset
		const uint32_t w0 = sw[sb >> 2], w1 = sw[(sb >> 2) + 1u];
		const uint32_t sh8 = (sb & 3u) * 8u;
		const uint32_t v = sh8 ? (w0 >> sh8) | (w1 << (32u - sh8)) : w0;
		((uint32_t*)(dst + head))[i] = v;
	}
	const uint32_t done = head + nwords * 4u;
	if(tid < nb - done) dst[done + tid] = src[done + tid];
}

extern "C" size_t fa_pack_lds_bytes(const FaCfg *c, uint32_t *bitbuf_words_out)
{
	// one subframe never exceeds N * (bps + 2) bits (estimate below verbatim + N/2 slack), plus head and carry
	const uint32_t maxbps = c->bps + 1u;
	const size_t bits = (size_t)c->N * (maxbps + 1u) + 8u + 32u + 32u * (maxbps + 15u) + 64u + 64u;
	const uint32_t words = (uint32_t)((bits + 31u) / 32u) + 4u;
	if(bitbuf_words_out) *bitbuf_words_out = words;
	return (((size_t)c->N + 32u) * 4u + (size_t)words * 4u + 15u) & ~(size_t)15u;
}

extern "C" void fa_launch_pack(const void *pcm, const FaCfg *c, const FaWork *w, hipStream_t st)
{
	uint32_t words = 0;
	const size_t lds = fa_pack_lds_bytes(c, &words);
	hipFuncSetAttribute((const void*)fa_pack_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	hipLaunchKernelGGL(fa_pack_kernel, dim3(c->F), dim3(256), lds, st, pcm, *c, *w, words);
}

extern "C" void fa_launch_gather(const FaCfg *c, const FaWork *w, void *out, size_t out_cap, uint32_t *frame_bytes_out, hipStream_t st)
{
	hipLaunchKernelGGL(fa_scan_kernel, dim3(1), dim3(1024), 0, st, *c, *w);
	hipLaunchKernelGGL(fa_gather_kernel, dim3(c->F), dim3(256), 0, st, *c, *w, (uint8_t*)out, (unsigned long long)out_cap, frame_bytes_out);
}
