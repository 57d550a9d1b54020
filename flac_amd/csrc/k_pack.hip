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
#include "residual.h"
#include <type_traits>

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


// The codes of a frame header that depend on blocksize and sample rate, and its length in
// bytes (stream_encoder_framing.c:219-364; UTF-8 frame number bitwriter.c:784-832).
struct FrameHdrFields { uint32_t bs_code, bs_hint, sr_code, sr_hint, bps_code, nbytes; };
__device__ __forceinline__ FrameHdrFields fa_frame_header_fields(const FaCfg &c, uint32_t n, uint32_t frame_number)
{
	FrameHdrFields h;
	h.bs_hint = 0; h.sr_hint = 0;
	switch(n) {
	case 192: h.bs_code = 1; break; case 576: h.bs_code = 2; break; case 1152: h.bs_code = 3; break; case 2304: h.bs_code = 4; break; case 4608: h.bs_code = 5; break;
	case 256: h.bs_code = 8; break; case 512: h.bs_code = 9; break; case 1024: h.bs_code = 10; break; case 2048: h.bs_code = 11; break; case 4096: h.bs_code = 12; break;
	case 8192: h.bs_code = 13; break; case 16384: h.bs_code = 14; break; case 32768: h.bs_code = 15; break;
	default: h.bs_code = h.bs_hint = (n <= 0x100u) ? 6u : 7u; break;
	}
	switch(c.rate) {
	case 88200: h.sr_code = 1; break; case 176400: h.sr_code = 2; break; case 192000: h.sr_code = 3; break; case 8000: h.sr_code = 4; break; case 16000: h.sr_code = 5; break;
	case 22050: h.sr_code = 6; break; case 24000: h.sr_code = 7; break; case 32000: h.sr_code = 8; break; case 44100: h.sr_code = 9; break; case 48000: h.sr_code = 10; break; case 96000: h.sr_code = 11; break;
	default:
		if(c.rate <= 255000u && c.rate % 1000u == 0) h.sr_code = h.sr_hint = 12;
		else if(c.rate % 10u == 0) h.sr_code = h.sr_hint = 14;
		else if(c.rate <= 0xffffu) h.sr_code = h.sr_hint = 13;
		else h.sr_code = 0;
		break;
	}
	switch(c.bps) { case 8: h.bps_code = 1; break; case 12: h.bps_code = 2; break; case 16: h.bps_code = 4; break; case 20: h.bps_code = 5; break; case 24: h.bps_code = 6; break; default: h.bps_code = 0; break; }
	const uint32_t v = frame_number;
	const uint32_t utf8 = v < 0x80u ? 1u : v < 0x800u ? 2u : v < 0x10000u ? 3u : v < 0x200000u ? 4u : v < 0x4000000u ? 5u : 6u;
	h.nbytes = 4u + utf8 + (h.bs_hint == 6 ? 1u : h.bs_hint == 7 ? 2u : 0u) + (h.sr_hint == 12 ? 1u : h.sr_hint ? 2u : 0u) + 1u;
	return h;
}

// The bytes of the frame header including its CRC-8, handed one by one to put(index, byte);
// returns their count (<= 16).
template<class PUT>
__device__ __forceinline__ uint32_t fa_frame_header(const FaCfg &c, uint32_t n, uint32_t frame_number, uint32_t assignment, PUT &&put)
{
	const FrameHdrFields h = fa_frame_header_fields(c, n, frame_number);
	const uint32_t ch_code = (assignment == 0) ? c.channels - 1u : 7u + assignment;
	uint32_t nb = 0, crc = 0;
	auto hb = [&](uint32_t byte) { byte &= 0xffu; crc = crc8_byte(crc, byte); put(nb, byte); nb++; };
	hb(0xff); hb(0xf8); hb((h.bs_code << 4) | h.sr_code); hb((ch_code << 4) | (h.bps_code << 1));
	{
		const uint32_t v = frame_number;
		if(v < 0x80u) hb(v);
		else if(v < 0x800u) { hb(0xC0u | (v >> 6)); hb(0x80u | (v & 0x3Fu)); }
		else if(v < 0x10000u) { hb(0xE0u | (v >> 12)); hb(0x80u | ((v >> 6) & 0x3Fu)); hb(0x80u | (v & 0x3Fu)); }
		else if(v < 0x200000u) { hb(0xF0u | (v >> 18)); hb(0x80u | ((v >> 12) & 0x3Fu)); hb(0x80u | ((v >> 6) & 0x3Fu)); hb(0x80u | (v & 0x3Fu)); }
		else if(v < 0x4000000u) { hb(0xF8u | (v >> 24)); hb(0x80u | ((v >> 18) & 0x3Fu)); hb(0x80u | ((v >> 12) & 0x3Fu)); hb(0x80u | ((v >> 6) & 0x3Fu)); hb(0x80u | (v & 0x3Fu)); }
		else { hb(0xFCu | (v >> 30)); hb(0x80u | ((v >> 24) & 0x3Fu)); hb(0x80u | ((v >> 18) & 0x3Fu)); hb(0x80u | ((v >> 12) & 0x3Fu)); hb(0x80u | ((v >> 6) & 0x3Fu)); hb(0x80u | (v & 0x3Fu)); }
	}
	if(h.bs_hint == 6) hb(n - 1u);
	else if(h.bs_hint == 7) { hb((n - 1u) >> 8); hb(n - 1u); }
	if(h.sr_hint == 12) hb(c.rate / 1000u);
	else if(h.sr_hint == 13) { hb(c.rate >> 8); hb(c.rate); }
	else if(h.sr_hint == 14) { hb((c.rate / 10u) >> 8); hb(c.rate / 10u); }
	hb(crc);
	return nb;
}

// Channel assignment of frame f (stream_encoder.c:3052-3087; loose rule :2958-2967, :3059-3061).
__device__ __forceinline__ uint32_t fa_assignment(const FaCfg &c, const FaWork &w, uint32_t f)
{
	uint32_t a = 0;
	if(c.midside) {
		const uint32_t fa = fa_frame_num(c, w, f);
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
	return a;
}

struct PackShared {
	uint32_t wave_tot[4];
	uint32_t assignment;
	uint32_t carry_bits;   // valid bits in bitbuf[0] carried from what was written before
	uint32_t words_out;    // 32-bit words already flushed to the slot
	uint32_t overflow;
	uint32_t crc;          // CRC-16 of everything flushed so far
	uint32_t crc_chunk;    // CRC-16 of the words being flushed
	uint32_t hdr_bytes;
	int32_t  q[FLACAMD_MAX_LPC_ORDER];
	uint8_t  k[FA_MAX_PARTS];
	uint16_t crc4[4][256]; // CRC-16 of one byte followed by 0..3 zero bytes
};

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

// Per-thread bit writer: a run of tokens at increasing bit positions.  Words that lie
// strictly inside the thread's run are stored whole; the first and the last word may be
// shared with the neighbouring threads and are OR-ed atomically.  The buffer is zero
// beforehand, so zero runs are skipped without touching memory.
struct BitRun {
	uint32_t *buf; uint32_t pos, wi, first_wi, cur;
	__device__ __forceinline__ void begin(uint32_t *b, uint32_t p) { buf = b; pos = p; wi = first_wi = p >> 5; cur = 0; }
	__device__ __forceinline__ void store_word() { if(cur) { if(wi == first_wi) atomicOr(&buf[wi], cur); else buf[wi] = cur; } }
	__device__ __forceinline__ void zeros(uint32_t nz)
	{
		pos += nz;
		const uint32_t nw = pos >> 5;
		if(nw != wi) { store_word(); wi = nw; cur = 0; }
	}
	// nbits in 1..32, v already masked to nbits
	__device__ __forceinline__ void put(uint32_t v, uint32_t nbits)
	{
		const uint32_t off = pos & 31u;
		pos += nbits;
		if(off + nbits < 32u) cur |= v << (32u - off - nbits);
		else {
			const uint32_t spill = off + nbits - 32u;
			cur |= spill ? (v >> spill) : v;
			store_word();
			wi++;
			cur = spill ? (v << (32u - spill)) : 0u;
		}
	}
	__device__ __forceinline__ void end() { if(cur) atomicOr(&buf[wi], cur); }
};

// CRC-16 of n 32-bit big-endian words starting at p, register starting at 0 (slice by four)
__device__ __forceinline__ uint32_t crc16_words(const uint32_t *p, uint32_t n, const uint16_t (&t)[4][256])
{
	uint32_t crc = 0;
	for(uint32_t i = 0; i < n; i++) {
		const uint32_t v = p[i] ^ (crc << 16);
		crc = (uint32_t)t[3][v >> 24] ^ (uint32_t)t[2][(v >> 16) & 255u] ^ (uint32_t)t[1][(v >> 8) & 255u] ^ (uint32_t)t[0][v & 255u];
	}
	return crc;
}

// Flush the completed words of the bit buffer (bits [0, total_bits)) to the slot, fold them
// into the running CRC-16, and keep the partial word as the new carry.  All threads call it.
__device__ __forceinline__ void flush_words(uint32_t *bitbuf, uint32_t total_bits, uint32_t *slot_words, PackShared &sh, const FaWork &w, uint32_t segw, uint32_t dbg)
{
	const uint32_t full = total_bits >> 5, wo = sh.words_out, tid = threadIdx.x;
	__syncthreads();
	if(!(dbg & 0x100u)) for(uint32_t i = tid; i < full; i += 256) slot_words[wo + i] = __builtin_bswap32(bitbuf[i]);
	// CRC of the chunk: thread j takes the segw words that end segw*j words before the end
	if(!(dbg & 0x20u)) {
		const uint32_t e1 = (segw * tid < full) ? full - segw * tid : 0u, b0 = (e1 > segw) ? e1 - segw : 0u;
		uint32_t crc = 0;
		if(e1 > 0) crc = gf16_mul(crc16_words(bitbuf + b0, e1 - b0, sh.crc4), w.crc16_xpw[segw * tid]);
#pragma unroll
		for(int o = 32; o > 0; o >>= 1) crc ^= (uint32_t)__shfl_xor((int)crc, o, 64);
		if((tid & 63u) == 0 && crc) atomicXor(&sh.crc_chunk, crc);
	}
	const uint32_t partial = bitbuf[full];
	__syncthreads();
	for(uint32_t i = tid; i <= full + 1u; i += 256) bitbuf[i] = 0;
	__syncthreads();
	if(tid == 0) {
		bitbuf[0] = partial; sh.carry_bits = total_bits & 31u; sh.words_out = wo + full;
		sh.crc = gf16_mul(sh.crc, w.crc16_xpw[full]) ^ sh.crc_chunk;
		sh.crc_chunk = 0;
	}
	__syncthreads();
}

template<bool SPILL>
__global__ void __launch_bounds__(256, 3) fa_pack_kernel(const void *__restrict__ pcm, FaCfg c, FaWork w, uint32_t bitbuf_words)
{
	extern __shared__ __align__(16) unsigned char lds_raw[];
	__shared__ PackShared sh;
	// padded signal of the subframe being written, xs_words_for(N) words: in LDS, or for blocks too long for that in the
	// frame's first row of the spill buffer (this workgroup's alone)
	int32_t *xs; uint32_t *bitbuf;                              // bitbuf: [bitbuf_words]
	if constexpr(SPILL) { xs = w.xs_spill + (size_t)blockIdx.x * c.nsig * xs_words_for(c.N); bitbuf = (uint32_t*)lds_raw; }
	else { xs = (int32_t*)lds_raw; bitbuf = (uint32_t*)(xs + xs_words_for(c.N)); }

	const uint32_t f = blockIdx.x, tid = threadIdx.x;
	const uint32_t n = fa_frame_len(c, w, f);
	const uint64_t base = (uint64_t)f * c.N;
	uint32_t *slot_words = (uint32_t*)(w.slots + (size_t)f * c.slot_bytes);
	uint8_t *slot = w.slots + (size_t)f * c.slot_bytes;
	const uint32_t cap_bits = (bitbuf_words - 2u) * 32u;
	const uint32_t segw = (bitbuf_words + 255u) / 256u;
	const uint32_t kspread = c.max_po < FLACAMD_MAX_PARTITION_ORDER ? c.max_po : FLACAMD_MAX_PARTITION_ORDER;

	for(uint32_t i = tid; i < bitbuf_words; i += 256) bitbuf[i] = 0;
	if(tid < 32) xs[xs_idx((int32_t)tid - 32)] = 0;
	for(uint32_t i = tid; i < 1024; i += 256) sh.crc4[i >> 8][i & 255u] = w.crc16_table4[i];
	__syncthreads();

	// ---- channel assignment (stream_encoder.c:3052-3087; loose rule :2958-2967, :3059-3061)
	if(tid == 0) {
		const uint32_t a = fa_assignment(c, w, f);
		sh.assignment = a; sh.carry_bits = 0; sh.words_out = 0; sh.overflow = 0; sh.crc = 0; sh.crc_chunk = 0;
		w.assignment[f] = a;

		// ---- frame header (stream_encoder_framing.c:219-364), built by one lane
		const uint32_t nb = fa_frame_header(c, n, fa_frame_num(c, w, f), a, [&](uint32_t i, uint32_t byte) { bitbuf[i >> 2] |= byte << (24u - 8u * (i & 3u)); });
		sh.hdr_bytes = nb;
	}
	__syncthreads();
	flush_words(bitbuf, sh.hdr_bytes * 8u, slot_words, sh, w, segw, c.debug);

	const uint32_t assignment = sh.assignment;
	const uint32_t nsub = c.channels;
	const uint32_t nrounds = (n + 256u * CH - 1u) / (256u * CH);

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
		{
			const uint32_t nup = (n + CH - 1u) & ~(CH - 1u);
			if(!(c.debug & 0x200u)) for(uint32_t i = tid; i < nup; i += 256) xs[xs_idx((int32_t)i)] = (i < n) ? (fa_signal(pcm, c, sig, base + i) >> wasted) : 0;
		}
		if(tid < FLACAMD_MAX_LPC_ORDER) sh.q[tid] = win->q[tid];
		// the winner's Rice parameters are stored spread to the finest configured order
		if(type >= FA_SUB_FIXED) for(uint32_t p = tid; p < (1u << po); p += 256) sh.k[p] = win->k[p << (kspread - po)];
		__syncthreads();

		// bits of the subframe head: type byte, wasted-bits unary, warm-up, lpc parameters, coding method + partition order
		uint32_t head_bits = 8u + wasted;
		if(type == FA_SUB_FIXED) head_bits += order * sbps + 6u;
		else if(type == FA_SUB_LPC) head_bits += order * sbps + 9u + order * prec + 6u;
		const uint32_t plen = rice2 ? 5u : 4u;
		const uint32_t dps = n >> po;
		const uint32_t carry = sh.carry_bits;
		const uint32_t smask = sbps < 32u ? (1u << sbps) - 1u : 0xffffffffu;

		// ---- head (one lane; a few dozen tokens)
		if(tid == 0 && carry + head_bits <= cap_bits && !(c.debug & 0x40u)) {
			BitRun br; br.begin(bitbuf, carry);
			uint32_t tb;
			switch(type) { // stream_encoder_framing.c:371, :383, :421, :468
			case FA_SUB_CONSTANT: tb = 0x00; break; case FA_SUB_VERBATIM: tb = 0x02; break;
			case FA_SUB_FIXED: tb = 0x10u | (order << 1); break; default: tb = 0x40u | ((order - 1u) << 1); break;
			}
			br.put(tb | (wasted ? 1u : 0u), 8);
			if(wasted) { br.zeros(wasted - 1u); br.put(1u, 1); }
			if(type >= FA_SUB_FIXED) {
				for(uint32_t i = 0; i < order; i++) br.put((uint32_t)xs[xs_idx((int32_t)i)] & smask, sbps);
				if(type == FA_SUB_LPC) {
					br.put(prec - 1u, 4); br.put(shift, 5);
					for(uint32_t i = 0; i < order; i++) br.put((uint32_t)sh.q[i] & ((1u << prec) - 1u), prec);
				}
				br.put(rice2, 2); br.put(po, 4);
			}
			br.end();
		}

		uint32_t bitpos = carry + head_bits;
		bool ovf = bitpos > cap_bits;
		for(uint32_t r = 0; r < nrounds && !ovf; r++) {
			const uint32_t i0 = (r * 256u + tid) * CH;
			uint32_t u[CH];       // folded residual / verbatim sample bits
			uint32_t meta[CH];    // bit length of the token; bit 31: a partition parameter precedes it; bits 24..28: k
			uint32_t mybits = 0;
#pragma unroll
			for(int s = 0; s < CH; s++) { u[s] = 0; meta[s] = 0; }
			if(i0 < n) {
				if(type == FA_SUB_CONSTANT) { if(i0 == 0) { u[0] = (uint32_t)xs[xs_idx(0)] & smask; meta[0] = sbps; mybits = sbps; } }
				else if(type == FA_SUB_VERBATIM) {
#pragma unroll
					for(int s = 0; s < CH; s++) if(i0 + (uint32_t)s < n) { u[s] = (uint32_t)xs[xs_idx((int32_t)(i0 + s))] & smask; meta[s] = sbps; mybits += sbps; }
				}
				else {
					int32_t res[CH];
					if(c.debug & 0x80u) {
#pragma unroll
						for(int s = 0; s < CH; s++) res[s] = (int32_t)((i0 + s) & 7u);
					}
					else if(order <= HIST) {
						int32_t xw[HIST + CH];
#pragma unroll
						for(int j = 0; j < HIST + CH; j++) xw[j] = xs[xs_idx((int32_t)i0 - HIST + j)];
						auto keep = [&](int s, int32_t rv, int) { res[s] = rv; };
						if(type == FA_SUB_FIXED) fixed_each_dispatch(order, xw, keep);
						else {
							int32_t q[HIST];
#pragma unroll
							for(int j = 0; j < HIST; j++) q[j] = sh.q[j];
							if(wide) lpc_each_dispatch<MAC_WIDE>(order, xw, q, shift, keep);
							else if(sbps <= 24) lpc_each_dispatch<MAC_I24>(order, xw, q, shift, keep);
							else lpc_each_dispatch<MAC_I32>(order, xw, q, shift, keep);
						}
					}
					else {
#pragma unroll
						for(int s = 0; s < CH; s++) { const uint32_t i = i0 + (uint32_t)s; res[s] = (i >= order && i < n) ? residual_generic(xs, i, true, order, shift, wide, sh.q) : 0; }
					}
					// partition walk: sample i lies in partition i / dps (partition 0 starts at `order`)
					uint32_t p = po ? i0 / dps : 0u;
					uint32_t next = po ? (p + 1u) * dps : 0xffffffffu;
					uint32_t kcur = sh.k[p];
#pragma unroll
					for(int s = 0; s < CH; s++) {
						const uint32_t i = i0 + (uint32_t)s;
						if(i == next) { p++; next += dps; kcur = sh.k[p]; }
						if(i >= order && i < n) {
							const uint32_t uu = ((uint32_t)res[s] << 1) ^ (uint32_t)(res[s] >> 31);
							const bool first = (i == (p ? p * dps : order));
							const uint32_t len = (uu >> kcur) + 1u + kcur;
							u[s] = uu; meta[s] = len | (kcur << 24) | (first ? 0x80000000u : 0u);
							mybits += len + (first ? plen : 0u);
						}
					}
				}
			}
			uint32_t round_bits;
			const uint32_t myoff = block_exscan(mybits, sh, round_bits);
			if(bitpos + round_bits > cap_bits) {
				// long blocks: the buffer holds part of a subframe; write out what is complete and go on behind the carry
				flush_words(bitbuf, bitpos, slot_words, sh, w, segw, c.debug);
				bitpos = sh.carry_bits;
				if(bitpos + round_bits > cap_bits) { ovf = true; break; }
			}
			if(mybits && !(c.debug & 0x10u)) {
				BitRun br; br.begin(bitbuf, bitpos + myoff);
				if(type <= FA_SUB_VERBATIM) {
#pragma unroll
					for(int s = 0; s < CH; s++) if(meta[s]) br.put(u[s], meta[s]);
				}
				else {
#pragma unroll
					for(int s = 0; s < CH; s++) if(meta[s]) {
						const uint32_t k = (meta[s] >> 24) & 31u;
						if(meta[s] & 0x80000000u) br.put(k, plen);
						br.zeros(u[s] >> k);                                         // unary part
						br.put((1u << k) | (u[s] & ((1u << k) - 1u)), k + 1u);      // stop bit + k low bits
					}
				}
				br.end();
			}
			bitpos += round_bits;
		}
		if(ovf) { // cannot happen for estimates that did not wrap; never write past the buffer
			if(tid == 0) { sh.overflow = 1; atomicOr(w.status, 1u); }
			__syncthreads();
			break;
		}
		flush_words(bitbuf, bitpos, slot_words, sh, w, segw, c.debug);
	}

	// ---- zero-pad to a byte, CRC-16, footer (stream_encoder.c:2890-2905)
	if(sh.overflow) { if(tid == 0) w.frame_bytes[f] = 0; return; }
	if(tid == 0) {
		const uint32_t tail_bits = sh.carry_bits, tail_bytes = (tail_bits + 7u) >> 3;
		const uint32_t body_bytes = sh.words_out * 4u + tail_bytes;
		uint32_t crc = sh.crc;
		for(uint32_t i = 0; i < tail_bytes; i++) {
			const uint32_t byte = (bitbuf[0] >> (24u - 8u * i)) & 0xffu;
			slot[sh.words_out * 4u + i] = (uint8_t)byte;
			crc = ((crc << 8) & 0xffffu) ^ sh.crc4[0][(crc >> 8) ^ byte];
		}
		slot[body_bytes] = (uint8_t)(crc >> 8); slot[body_bytes + 1u] = (uint8_t)crc;
		w.frame_bytes[f] = body_bytes + 2u;
	}
}

// ---- fast path: mono / stereo streams, blocksize <= 4096, predictor orders <= HIST ----------
//
// One 256-thread workgroup per frame, 16 samples per thread, every subframe of the frame in
// the same pass.  Nothing depends on a previous step's global load: the winners' headers are
// read first (wave-uniform), then every thread fetches its 16 + HIST wide samples straight
// from the interleaved PCM into registers (no LDS copy of the signal) together with its Rice
// parameter, forms the signals it needs (left/right/mid/side) from the same registers,
// computes residual tokens and their bit lengths, and ONE workgroup scan over (bits of
// subframe 0, bits of subframe 1) places every token of the frame.  The subframe heads
// (type, warm-up samples, coefficients) are written token-parallel by the first wave.  The
// whole frame is assembled in LDS and flushed once: coalesced word stores, CRC-16 per thread
// segment folded with x^(32 w) multipliers.  Four barriers per frame.
#define PF_THREADS 256
#define PF_LB_OWN (1ull << 62)
#define PF_LB_ALL (2ull << 62)
#define PF_LB_MASK (3ull << 62)
#define PF_CH 16
#define PF_W (HIST + PF_CH)
#define PF_LB_SPINS 64u   // polls of a silent predecessor before a frame goes to its slot instead of its place (see the look-back)

struct PackFastShared {
	uint32_t wave_tot[PF_THREADS / 64][2];
	uint32_t crc_acc;
	unsigned long long out_off;   // where the frame goes in the caller's buffer (FaCfg::direct_out)
	uint16_t crc4[4][256];
};

struct SubDesc { uint32_t type, order, prec, shift, po, rice2, wasted, wide, sbps, sig, g, head_bits, nwarm; };

// OR a token into the bit buffer where any word may be shared with other writers
__device__ __forceinline__ void put_shared(uint32_t *buf, uint32_t pos, uint32_t v, uint32_t nbits)
{
	const uint32_t off = pos & 31u, wi = pos >> 5;
	if(off + nbits <= 32u) atomicOr(&buf[wi], v << (32u - off - nbits));
	else { const uint32_t spill = off + nbits - 32u; atomicOr(&buf[wi], v >> spill); atomicOr(&buf[wi + 1u], v << (32u - spill)); }
}

// four consecutive wide samples starting at batch sample `wide` (a multiple of 4) into a[4], b[4]
// (channel 0 and channel 1); samples at or past `end` read as zero
template<int NSUB>
__device__ __forceinline__ void pf_load4(const void *__restrict__ pcm, uint32_t fmt, uint64_t wide, uint64_t end, int32_t (&a)[4], int32_t (&b)[4])
{
	if(wide + 4u <= end) {
		if(NSUB == 2) {
			if(fmt == FLACAMD_PCM_S16) {
				const uint4 v = *(const uint4*)((const uint32_t*)pcm + wide);
				const uint32_t t[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
				for(int j = 0; j < 4; j++) { a[j] = (int32_t)(int16_t)(t[j] & 0xffffu); b[j] = (int32_t)t[j] >> 16; }
			}
			else {
				const uint4 v0 = *(const uint4*)((const int32_t*)pcm + wide * 2u), v1 = *(const uint4*)((const int32_t*)pcm + wide * 2u + 4u);
				a[0] = (int32_t)v0.x; b[0] = (int32_t)v0.y; a[1] = (int32_t)v0.z; b[1] = (int32_t)v0.w;
				a[2] = (int32_t)v1.x; b[2] = (int32_t)v1.y; a[3] = (int32_t)v1.z; b[3] = (int32_t)v1.w;
			}
		}
		else {
			if(fmt == FLACAMD_PCM_S16) {
				const uint2 v = *(const uint2*)((const int16_t*)pcm + wide);
				a[0] = (int32_t)(int16_t)(v.x & 0xffffu); a[1] = (int32_t)v.x >> 16; a[2] = (int32_t)(int16_t)(v.y & 0xffffu); a[3] = (int32_t)v.y >> 16;
			}
			else {
				const uint4 v = *(const uint4*)((const int32_t*)pcm + wide);
				a[0] = (int32_t)v.x; a[1] = (int32_t)v.y; a[2] = (int32_t)v.z; a[3] = (int32_t)v.w;
			}
#pragma unroll
			for(int j = 0; j < 4; j++) b[j] = 0;
		}
	}
	else {
#pragma unroll
		for(int j = 0; j < 4; j++) {
			const bool ok = wide + (uint64_t)j < end;
			a[j] = ok ? fa_pcm(pcm, fmt, NSUB, wide + j, 0) : 0;
			b[j] = (ok && NSUB == 2) ? fa_pcm(pcm, fmt, NSUB, wide + j, 1) : 0;
		}
	}
}

// One wave per frame: the channel assignment and the pack record (see FaPackRec).
template<int NSUB>
__global__ void __launch_bounds__(64) fa_assign_kernel(FaCfg c, FaWork w)
{
	const uint32_t f = blockIdx.x, lane = threadIdx.x;
	const uint32_t a = fa_assignment(c, w, f);
	FaPackRec *rec = &w.packrec[f];
#pragma unroll
	for(int sc = 0; sc < NSUB; sc++) {
		uint32_t sig = (uint32_t)sc;
		if(c.midside) { // stream_encoder.c:3096-3115
			if(a == 1) sig = sc == 0 ? 0u : 3u;
			else if(a == 2) sig = sc == 0 ? 3u : 1u;
			else if(a == 3) sig = sc == 0 ? 2u : 3u;
		}
		const FaWinner *win = &w.winner[f * c.nsig + sig];
		if(lane == 0) { rec->hdr0[sc] = win->hdr[0]; rec->sbps[sc] = win->hdr[2]; rec->sig[sc] = sig; }
		if(lane < 12u) rec->q[sc][lane] = win->q[lane];
		((uint32_t*)rec->k[sc])[lane] = ((const uint32_t*)win->k)[lane];   // 256 bytes
	}
	if(lane == 0) { rec->assignment = a; w.assignment[f] = a; }
}

// 256 threads x 16 samples capped at 128 registers: four workgroups per CU (512 x 8 at two / three workgroups per CU
// was a third slower).
// PACKED16: 16-bit stereo input (FLACAMD_PCM_S16, NSUB == 2), fetched as one register per wide sample.
// NARROW: no subframe of the stream can need the 64-bit prediction sum or samples beyond 24 bits (decided on the host).
// The kernel is a chain of load, scan, emission, look-back, CRC and flush with barriers between, which more resident
// workgroups hide.  With both (the 64-bit path is where the register count peaks) and the first subframe written out
// before the second is looked at, it needs 79 registers: six workgroups per CU, nothing spilled.  Per hour of audio:
// 0.90 ms at four workgroups per CU, 0.80 at five, 0.75 at six, 0.73 at seven (five registers in scratch memory, which
// is 0.2 GB of traffic), 1.00 at eight.
// CD8: the stream is 16-bit / 44.1 kHz stereo in blocks of 4096 with mid/side and partition orders up to 6, one stream per
// batch, frames written in place (`flac -8`, -5 ... on CD audio; the launcher checks every field): those parameters are
// constants of this variant instead of values that occupy scalar registers across the whole kernel.
template<int NSUB, bool PACKED16, bool NARROW, bool CD8>
__global__ void __launch_bounds__(PF_THREADS, PACKED16 ? (NARROW ? 6 : 5) : 4) fa_pack_fast_kernel(const void *__restrict__ pcm, FaCfg c_arg, FaWork w, uint32_t bitbuf_words)
{
	FaCfg c = c_arg;
	if(CD8) {
		c.N = 4096u; c.channels = 2u; c.bps = 16u; c.rate = 44100u; c.nsig = 4u; c.midside = 1u; c.multi = 0u; c.max_po = 6u; c.debug = 0u;
		c.pcm_format = FLACAMD_PCM_S16; c.direct_out = 1u;
	}
	extern __shared__ __align__(16) unsigned char lds_raw[];
	__shared__ PackFastShared sh;
	uint32_t *bitbuf = (uint32_t*)lds_raw;

	const uint32_t f = blockIdx.x, tid = threadIdx.x;
	const uint32_t n = fa_frame_len(c, w, f);
	const uint64_t base = (uint64_t)f * c.N;
	uint8_t *slot = w.slots + (size_t)f * c.slot_bytes;
	uint32_t *slot_words = (uint32_t*)slot;
	const uint32_t cap_bits = (bitbuf_words - 2u) * 32u;
	const uint32_t i0 = tid * PF_CH;
	const uint32_t kspread = c.max_po < FLACAMD_MAX_PARTITION_ORDER ? c.max_po : FLACAMD_MAX_PARTITION_ORDER;

	for(uint32_t i = tid; i < bitbuf_words; i += PF_THREADS) bitbuf[i] = 0;
	for(uint32_t i = tid; i < 512u; i += PF_THREADS) ((uint32_t*)sh.crc4)[i] = ((const uint32_t*)w.crc16_table4)[i];
	if(tid == 0) sh.crc_acc = 0;

	// ---- wave-uniform decisions, all from the frame's pack record: one round of loads
	const FaPackRec *rec = &w.packrec[f];
	const uint32_t a = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec->assignment);
	SubDesc d[NSUB];
	int32_t qsel[NSUB][HIST];
	uint32_t kpre[NSUB];
#pragma unroll
	for(int sc = 0; sc < NSUB; sc++) {
		const uint32_t h0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec->hdr0[sc]);
		SubDesc &s = d[sc];
		s.sig = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec->sig[sc]); s.g = f * c.nsig + s.sig;
		s.sbps = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec->sbps[sc]);
		s.type = FA_WIN_TYPE(h0); s.order = FA_WIN_ORDER(h0); s.prec = FA_WIN_PREC(h0); s.shift = FA_WIN_SHIFT(h0); s.po = FA_WIN_PO(h0);
		s.rice2 = FA_WIN_RICE2(h0); s.wasted = FA_WIN_WASTED(h0); s.wide = FA_WIN_WIDE(h0);
		s.nwarm = s.type == FA_SUB_CONSTANT ? 1u : s.type >= FA_SUB_FIXED ? s.order : 0u;
		s.head_bits = 8u + s.wasted + s.nwarm * s.sbps + (s.type == FA_SUB_LPC ? 9u + s.order * s.prec : 0u) + (s.type >= FA_SUB_FIXED ? 6u : 0u);
#pragma unroll
		for(int j = 0; j < HIST; j++) qsel[sc][j] = __builtin_amdgcn_readfirstlane(rec->q[sc][j]);
		// the Rice parameters are stored spread to the finest configured order: indexed by position, not by the winner's order
		const uint32_t e = i0 / (c.N >> kspread);
		kpre[sc] = rec->k[sc][e < FA_MAX_PARTS ? e : FA_MAX_PARTS - 1u];
	}
	const FrameHdrFields hf = fa_frame_header_fields(c, n, fa_frame_num(c, w, f));

	// ---- this thread's samples i0 - HIST .. i0 + PF_CH - 1 of both channels.  16-bit stereo is fetched packed (left | right << 16,
	// one register per wide sample) by each subframe in turn, which forms its signal in the same registers: the second
	// fetch hits the cache, and holding the words across the first subframe's arithmetic instead costs the registers that
	// decide whether five workgroups fit a CU
	auto fetch_packed = [&](const uint32_t *src, uint32_t (&raw)[PF_W]) {
#pragma unroll
		for(int j = 0; j < PF_W / 4; j++) {
			const int32_t rel = (int32_t)i0 - HIST + 4 * j;
			uint4 v = make_uint4(0u, 0u, 0u, 0u);
			if(rel >= 0 && (uint32_t)rel + 4u <= n) v = *(const uint4*)(src + base + (uint32_t)rel);
			else if(rel >= 0 && (uint32_t)rel < n) { // the tail of a short last block: never read past the end of the stream
				const uint32_t *q = src + base + (uint32_t)rel;
				v.x = q[0];
				if((uint32_t)rel + 1u < n) v.y = q[1];
				if((uint32_t)rel + 2u < n) v.z = q[2];
			}
			raw[4 * j] = v.x; raw[4 * j + 1] = v.y; raw[4 * j + 2] = v.z; raw[4 * j + 3] = v.w;
		}
	};
	// (other sample formats: both channels unpacked, likewise once per subframe)
	auto fetch_wide = [&](const void *src, int32_t (&xa)[PF_W], int32_t (&xb)[PF_W]) {
		const uint64_t end = base + n;
#pragma unroll
		for(int j = 0; j < PF_W / 4; j++) {
			int32_t ta[4], tb[4];
			const int32_t rel = (int32_t)i0 - HIST + 4 * j;
			if(rel < 0 || (uint32_t)rel >= n) {
#pragma unroll
				for(int q = 0; q < 4; q++) { ta[q] = 0; tb[q] = 0; }
			}
			else pf_load4<NSUB>(src, c.pcm_format, base + (uint32_t)rel, end, ta, tb);
#pragma unroll
			for(int q = 0; q < 4; q++) { xa[4 * j + q] = ta[q]; xb[4 * j + q] = tb[q]; }
		}
	};

	// ---- tokens of one subframe: folded residuals u[], the Rice parameter of this thread's
	// partition, where a partition parameter precedes a token, and the bits of the run
	uint32_t u[NSUB][PF_CH], kk[NSUB], bits[NSUB];
	int first_s[NSUB];
	bool bad = false;
	auto analyse = [&](auto sc_const) {
		constexpr int sc = decltype(sc_const)::value;
		const SubDesc &s = d[sc];
		uint32_t mybits = 0;
		kk[sc] = 0; first_s[sc] = -1;
#pragma unroll
		for(int q = 0; q < PF_CH; q++) u[sc][q] = 0;
		if(i0 < n && s.type != FA_SUB_CONSTANT) {
			const uint32_t dps = n >> s.po;
			if(s.po && (dps % PF_CH)) bad = true;
			const uint32_t p = s.po ? i0 / dps : 0u;
			if(s.type >= FA_SUB_FIXED) kk[sc] = kpre[sc];
			// which signal is the same for the whole workgroup: one branch around the loop, not a chain of selects per sample
			int32_t xw[PF_W];
			const uint32_t which = c.midside ? s.sig : (uint32_t)sc;
			if constexpr(PACKED16) {
				uint32_t raw[PF_W];
				const uint32_t *src = (const uint32_t*)pcm;
				if(sc == 1) asm volatile("" : "+s"(src));   // a fetch of its own, not the first subframe's values kept
				fetch_packed(src, raw);
				if(which == 0u) {
#pragma unroll
					for(int j = 0; j < PF_W; j++) xw[j] = (int32_t)(int16_t)(raw[j] & 0xffffu) >> s.wasted;
				}
				else if(which == 1u) {
#pragma unroll
					for(int j = 0; j < PF_W; j++) xw[j] = (int32_t)raw[j] >> (16u + s.wasted);
				}
				else if(which == 2u) {
#pragma unroll
					for(int j = 0; j < PF_W; j++) xw[j] = dot2_i16(raw[j], 0x00010001u, 0) >> (1u + s.wasted);   // left + right by one two-term multiply-add on the halves
				}
				else {
#pragma unroll
					for(int j = 0; j < PF_W; j++) xw[j] = dot2_i16(raw[j], 0xffff0001u, 0) >> s.wasted;            // left - right likewise
				}
			}
			else {
			int32_t xa[PF_W], xb[PF_W];
			const void *src = pcm;
			if(sc == 1) asm volatile("" : "+s"(src));
			fetch_wide(src, xa, xb);
			if(which == 0u) {
#pragma unroll
				for(int j = 0; j < PF_W; j++) xw[j] = xa[j] >> s.wasted;
			}
			else if(which == 1u) {
#pragma unroll
				for(int j = 0; j < PF_W; j++) xw[j] = xb[j] >> s.wasted;
			}
			else if(which == 2u) {
#pragma unroll
				for(int j = 0; j < PF_W; j++) xw[j] = ((int32_t)((uint32_t)xa[j] + (uint32_t)xb[j]) >> 1) >> s.wasted;
			}
			else {
#pragma unroll
				for(int j = 0; j < PF_W; j++) xw[j] = (int32_t)((uint32_t)xa[j] - (uint32_t)xb[j]) >> s.wasted;
			}
			}
			const uint32_t smask = s.sbps < 32u ? (1u << s.sbps) - 1u : 0xffffffffu;
			if(s.type == FA_SUB_VERBATIM) {
#pragma unroll
				for(int q = 0; q < PF_CH; q++) if(i0 + (uint32_t)q < n) { u[sc][q] = (uint32_t)xw[HIST + q] & smask; mybits += s.sbps; }
			}
			else {
				int32_t res[PF_CH];
				auto keep = [&](int q, int32_t rv, int) { res[q] = rv; };
				if(c.debug & 0x80u) {
#pragma unroll
					for(int q = 0; q < PF_CH; q++) res[q] = (int32_t)((i0 + q) & 7u);
				}
				else if(s.type == FA_SUB_FIXED) fixed_each_dispatch(s.order, xw, keep);
				else {
					int32_t qc[HIST];
#pragma unroll
					for(int j = 0; j < HIST; j++) qc[j] = qsel[sc][j];
					if(!NARROW && s.wide) lpc_each_dispatch<MAC_WIDE>(s.order, xw, qc, s.shift, keep);
					else if(NARROW || s.sbps <= 24) lpc_each_dispatch<MAC_I24>(s.order, xw, qc, s.shift, keep);
					else lpc_each_dispatch<MAC_I32>(s.order, xw, qc, s.shift, keep);
				}
				const uint32_t k = kk[sc];
				const uint32_t pstart = p ? p * dps : s.order;   // first sample that carries a residual in this partition
				if(pstart >= i0 && pstart < i0 + PF_CH) { first_s[sc] = (int)(pstart - i0); mybits += s.rice2 ? 5u : 4u; }
#pragma unroll
				for(int q = 0; q < PF_CH; q++) {
					const uint32_t i = i0 + (uint32_t)q;
					if(i >= s.order && i < n) {
						const uint32_t uu = ((uint32_t)res[q] << 1) ^ (uint32_t)(res[q] >> 31);
						u[sc][q] = uu;
						mybits += (uu >> k) + 1u + k;
					}
				}
			}
		}
		bits[sc] = mybits;
	};
	// ---- residual tokens of one subframe into the frame image
	uint32_t ex[NSUB], tot[NSUB];
	uint32_t start[NSUB];   // bit position of the head of each subframe
	auto emit = [&](auto sc_const) {
		constexpr int sc = decltype(sc_const)::value;
		const SubDesc &s = d[sc];
		if(bits[sc] != 0) {
			BitRun br; br.begin(bitbuf, start[sc] + s.head_bits + ex[sc]);
			if(s.type == FA_SUB_VERBATIM) {
#pragma unroll
				for(int q = 0; q < PF_CH; q++) if(i0 + (uint32_t)q < n) br.put(u[sc][q], s.sbps);
			}
			else {
				const uint32_t k = kk[sc], plen = s.rice2 ? 5u : 4u;
#pragma unroll
				for(int q = 0; q < PF_CH; q++) {
					const uint32_t i = i0 + (uint32_t)q;
					if(i >= s.order && i < n) {
						if(q == first_s[sc]) br.put(k, plen);
						br.zeros(u[sc][q] >> k);                                          // unary part
						br.put((1u << k) | (u[sc][q] & ((1u << k) - 1u)), k + 1u);       // stop bit + k low bits
					}
				}
			}
			br.end();
		}
	};
	// exclusive scan of one subframe's bits per thread over the workgroup (one barrier; the first also orders the zeroing
	// of bitbuf and the table load before their use)
	auto scan = [&](auto sc_const) {
		constexpr int sc = decltype(sc_const)::value;
		const uint32_t lane = tid & 63u, wv = tid >> 6;
		uint32_t inc = bits[sc];
#pragma unroll
		for(int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)inc, o, 64); if(lane >= (uint32_t)o) inc += t; }
		if(lane == 63u) sh.wave_tot[wv][sc] = inc;
		__syncthreads();
		uint32_t pre = 0, all = 0;
#pragma unroll
		for(uint32_t i = 0; i < PF_THREADS / 64; i++) { const uint32_t t = sh.wave_tot[i][sc]; all += t; if(i < wv) pre += t; }
		ex[sc] = pre + inc - bits[sc]; tot[sc] = all;
	};
	// The first subframe's tokens are written before the second subframe is looked at (a scan each instead of one for
	// both): its sixteen folded residuals per thread are gone from the registers by then
	const bool emit_on = !(c.debug & 0x10u);
	analyse(std::integral_constant<int, 0>{});
	scan(std::integral_constant<int, 0>{});
	start[0] = hf.nbytes * 8u;
	uint32_t total_bits = start[0] + d[0].head_bits + tot[0];
	bool over = total_bits > cap_bits;      // (the same for every thread)
	if constexpr(NSUB == 2) {
		if(!over && emit_on) emit(std::integral_constant<int, 0>{});
		analyse(std::integral_constant<int, 1>{});
		scan(std::integral_constant<int, 1>{});
		start[1] = total_bits;
		total_bits += d[1].head_bits + tot[1];
		over = over || total_bits > cap_bits;
	}
	if(__syncthreads_or(bad ? 1 : 0) || over) { // cannot happen for estimates that did not wrap; never write past the buffer
		if(tid == 0) {
			atomicOr(w.status, 1u); w.frame_bytes[f] = 0;
			if(c.direct_out) __hip_atomic_store(&w.lb[f], PF_LB_OWN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the frames behind this one must not wait for it
		}
		return;
	}
	// Development switch (FLACAMD_DEBUG bit 22), the counterpart of the reference's SPOTCHECK_ESTIMATE build
	// (stream_encoder.c:3444-3473): every coded subframe's estimate - the number the candidates were compared on - against the
	// bits it takes in the stream.  Totals in the debug words: subframes, those whose estimate is not the exact count, the
	// sum and the maximum of |estimate - bits|, the bits.
	if((c.debug & 0x400000u) && tid == 0) {
#pragma unroll
		for(int sc = 0; sc < NSUB; sc++) {
			const uint32_t est = w.winner[d[sc].g].hdr[1], act = d[sc].head_bits + tot[sc];
			const uint32_t diff = est > act ? est - act : act - est;
			atomicAdd(&w.dbg[0], 1ull); atomicAdd(&w.dbg[1], diff ? 1ull : 0ull); atomicAdd(&w.dbg[2], (unsigned long long)diff);
			atomicMax(&w.dbg[3], (unsigned long long)diff); atomicAdd(&w.dbg[4], (unsigned long long)act);
		}
	}
	// The frame's size is known before the rest of it is written: tell the frames behind (see the look-back below)
	const uint32_t frame_nb = ((total_bits + 7u) >> 3) + 2u;
	if(c.direct_out && tid == 0)
		__hip_atomic_store(&w.lb[f], (f == 0u ? PF_LB_ALL : PF_LB_OWN) | (unsigned long long)frame_nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

	// ---- frame header (one lane of the last wave), subframe heads (token-parallel, first wave)
	if(tid == PF_THREADS - 1u) {
		fa_frame_header(c, n, fa_frame_num(c, w, f), a, [&](uint32_t i, uint32_t byte) { atomicOr(&bitbuf[i >> 2], byte << (24u - 8u * (i & 3u))); });
	}
	if(tid < 32u * NSUB && !(c.debug & 0x40u)) {
		const uint32_t sc = tid >> 5, t = tid & 31u;
		// the descriptor of this lane's subframe, selected field by field (an indexed struct would live in scratch)
		const bool hi = NSUB == 2 && sc == 1u;
		SubDesc s;
#define PF_SEL(fld) s.fld = hi ? d[NSUB - 1].fld : d[0].fld
		PF_SEL(type); PF_SEL(order); PF_SEL(prec); PF_SEL(shift); PF_SEL(po); PF_SEL(rice2); PF_SEL(wasted); PF_SEL(wide); PF_SEL(sbps); PF_SEL(sig); PF_SEL(g); PF_SEL(head_bits); PF_SEL(nwarm);
#undef PF_SEL
		const uint32_t pos0 = hi ? start[NSUB - 1] : start[0];
		const uint32_t wtok = s.wasted ? 1u : 0u;
		const uint32_t lpc = s.type == FA_SUB_LPC ? 1u : 0u, coded = s.type >= FA_SUB_FIXED ? 1u : 0u;
		const uint32_t ntok = 1u + wtok + s.nwarm + lpc * (1u + s.order) + coded;
		if(t < ntok) {
			uint32_t pos = pos0, v, nb;
			if(t == 0) { // stream_encoder_framing.c:371, :383, :421, :468
				uint32_t tb;
				switch(s.type) { case FA_SUB_CONSTANT: tb = 0x00; break; case FA_SUB_VERBATIM: tb = 0x02; break; case FA_SUB_FIXED: tb = 0x10u | (s.order << 1); break; default: tb = 0x40u | ((s.order - 1u) << 1); break; }
				v = tb | wtok; nb = 8;
			}
			else if(t < 1u + wtok) { pos += 8u; v = 1u; nb = s.wasted; }               // wasted - 1 zeros and a one
			else if(t < 1u + wtok + s.nwarm) {                                          // warm-up samples (or the constant)
				const uint32_t j = t - 1u - wtok;
				pos += 8u + s.wasted + j * s.sbps;
				const uint64_t wide = base + j;
				const int32_t l = fa_pcm(pcm, c.pcm_format, NSUB, wide, 0), r = NSUB == 2 ? fa_pcm(pcm, c.pcm_format, NSUB, wide, 1) : 0;
				int32_t x;
				if(c.midside) x = s.sig == 0 ? l : s.sig == 1 ? r : s.sig == 2 ? ((int32_t)((uint32_t)l + (uint32_t)r) >> 1) : (int32_t)((uint32_t)l - (uint32_t)r);
				else x = sc == 0 ? l : r;
				v = (uint32_t)(x >> s.wasted) & (s.sbps < 32u ? (1u << s.sbps) - 1u : 0xffffffffu); nb = s.sbps;
			}
			else if(lpc && t == 1u + wtok + s.nwarm) { pos += 8u + s.wasted + s.nwarm * s.sbps; v = ((s.prec - 1u) << 5) | (s.shift & 31u); nb = 9; }
			else if(lpc && t < 2u + wtok + s.nwarm + s.order) {
				const uint32_t j = t - 2u - wtok - s.nwarm;
				pos += 8u + s.wasted + s.nwarm * s.sbps + 9u + j * s.prec;
				v = (uint32_t)rec->q[hi ? NSUB - 1 : 0][j] & ((1u << s.prec) - 1u); nb = s.prec;
			}
			else { pos += s.head_bits - 6u; v = (s.rice2 << 4) | s.po; nb = 6; }
			put_shared(bitbuf, pos, v, nb);
		}
	}

	// ---- residual tokens of the last subframe
	if(emit_on) {
		if constexpr(NSUB == 2) emit(std::integral_constant<int, 1>{});
		else emit(std::integral_constant<int, 0>{});
	}
	__syncthreads();

	// ---- where the frame goes (FaCfg::direct_out).  Frames lie end to end in the caller's buffer, so frame f starts at the
	// sum of the sizes of the frames before it.  Every workgroup has published its own size (state "own"); the first wave
	// walks back over its predecessors 64 at a time, adding sizes, until it meets one that already knows the total up to and
	// including itself (state "all"), then publishes its own total.  Workgroups are handed out in index order, so a
	// predecessor that has not published yet is normally running (or done) and will publish soon; the wait is bounded.
	const uint32_t full = total_bits >> 5;
	if(c.direct_out && tid < 64u) {
		unsigned long long before = 0;
		bool gave_up = false;
		if(f > 0u) {
			int32_t hi = (int32_t)f - 1;
			uint32_t waits = 0;
			for(;;) {
				const int32_t j = hi - (int32_t)tid;
				const unsigned long long stt = j >= 0 ? __hip_atomic_load(&w.lb[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : PF_LB_ALL;   // before frame 0: nothing, known
				const uint32_t flag = (uint32_t)(stt >> 62);
				const unsigned long long knows_all = __ballot(flag == 2u), silent = __ballot(flag == 0u);
				const uint32_t stop = knows_all ? (uint32_t)__builtin_ctzll(knows_all) : 64u;   // nearest predecessor that knows its total
				const unsigned long long upto = stop >= 63u ? ~0ull : ((2ull << stop) - 1ull);
				if(silent & upto) {
					// A short bound on the wait: the frame then goes to its slot and fa_place_late_kernel moves it afterwards.  When
					// this kernel shares the chip with others (another encoder's pack kernel, another process), an XCD may be so full
					// of other work that it launches its share of the workgroups late: frames on the other XCDs would spin for them
					// while holding the very CU slots the others are waiting for.  Nothing here waits for long.
					if(++waits > PF_LB_SPINS) { gave_up = true; break; }
					__builtin_amdgcn_s_sleep(2);
					continue;
				}
				unsigned long long v = tid <= stop ? (stt & ~PF_LB_MASK) : 0ull;
#pragma unroll
				for(int o = 32; o > 0; o >>= 1) v += (unsigned long long)__shfl_xor((long long)v, o, 64);
				before += v;
				if(knows_all) break;
				hi -= 64;
			}
			if(tid == 0 && !gave_up) __hip_atomic_store(&w.lb[f], PF_LB_ALL | (before + frame_nb), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		if((c.debug & 0x20000000u) && f % 3u == 1u) gave_up = true;   // development switch: every third frame takes the late path
		if(tid == 0) sh.out_off = gave_up ? ~0ull : before + (c.multi == 2u ? w.frame_shift[f] : 0u);
	}

	// ---- whole words to the slot (unless the frame goes straight to the caller), CRC-16 of thread segments combined with x^(32 w)
	if(!c.direct_out && !(c.debug & 0x100u)) for(uint32_t i = tid; i < full; i += PF_THREADS) slot_words[i] = __builtin_bswap32(bitbuf[i]);
	if(!(c.debug & 0x20u)) {
		const uint32_t segw = (full + PF_THREADS - 1u) / PF_THREADS;
		const uint32_t e1 = (segw * tid < full) ? full - segw * tid : 0u, b0 = (e1 > segw) ? e1 - segw : 0u;
		uint32_t crc = 0;
		if(e1 > 0) crc = gf16_mul(crc16_words(bitbuf + b0, e1 - b0, sh.crc4), w.crc16_xpw[segw * tid]);
#pragma unroll
		for(int o = 32; o > 0; o >>= 1) crc ^= (uint32_t)__shfl_xor((int)crc, o, 64);
		if((tid & 63u) == 0 && crc) atomicXor(&sh.crc_acc, crc);
	}
	__syncthreads();

	// ---- zero-pad to a byte, CRC-16, footer (stream_encoder.c:2890-2905): the bytes after the last whole word, by one lane;
	// with FaCfg::direct_out the whole words go to their place meanwhile
	const uint32_t tail_bits = total_bits & 31u, tail_bytes = (tail_bits + 7u) >> 3;
	const uint32_t body_bytes = full * 4u + tail_bytes, nb = body_bytes + 2u;
	const unsigned long long off = c.direct_out ? sh.out_off : 0ull;
	const bool late = c.direct_out && off == ~0ull;   // its place was not known in time: into the slot, moved later
	const bool place = c.direct_out && !late && off + nb <= w.out_cap;
	if(c.direct_out && !late && !place && tid == 0) atomicOr(w.status, 2u);
	uint8_t *dst = place ? w.out + off : slot;
	if(late) {
		for(uint32_t i = tid; i < full; i += PF_THREADS) slot_words[i] = __builtin_bswap32(bitbuf[i]);
		if(tid == 0) ((uint32_t*)(w.lb + c.F))[f] = 1u;   // the list of late frames follows the look-back states
	}
	if(tid == 0) {
		uint32_t crc = sh.crc_acc;
		const uint32_t last = bitbuf[full];
		const bool store = place || !c.direct_out || late;
		for(uint32_t i = 0; i < tail_bytes; i++) {
			const uint32_t byte = (last >> (24u - 8u * i)) & 0xffu;
			if(store) dst[full * 4u + i] = (uint8_t)byte;
			crc = ((crc << 8) & 0xffffu) ^ sh.crc4[0][(crc >> 8) ^ byte];
		}
		if(store) { dst[body_bytes] = (uint8_t)(crc >> 8); dst[body_bytes + 1u] = (uint8_t)crc; }
		w.frame_bytes[f] = nb;
		if(c.direct_out && w.frame_bytes_out) w.frame_bytes_out[f] = nb;
	}
	if(place) {
		// bytes up to the first aligned word of the destination, aligned words put together from the image's big-endian
		// words, the bytes that are left of the whole words
		const uint32_t wb = full * 4u;
		const uint32_t lead = (uint32_t)((4u - ((uintptr_t)dst & 3u)) & 3u), head = lead < wb ? lead : wb;
		auto image_byte = [&](uint32_t bb) { return (uint8_t)(bitbuf[bb >> 2] >> (24u - 8u * (bb & 3u))); };
		if(tid < head) dst[tid] = image_byte(tid);
		const uint32_t nwords = (wb - head) >> 2;
		uint32_t *dw = (uint32_t*)(dst + head);
		const uint32_t sh8 = head * 8u;   // the image is word-aligned, so every destination word starts `head` bytes into an image word
		// a wave's 64 words start on a 256-byte boundary of the destination (the first wave's round begins short), so that
		// every store fills whole cache lines and only a frame's first and last line are shared with its neighbours
		const int32_t skew = (int32_t)(((uintptr_t)dw & 255u) >> 2);
		for(int32_t i = (int32_t)tid - skew; i < (int32_t)nwords; i += PF_THREADS) {
			if(i < 0) continue;
			const uint32_t w0 = __builtin_bswap32(bitbuf[i]), w1 = __builtin_bswap32(bitbuf[i + 1]);
			dw[i] = sh8 ? (w0 >> sh8) | (w1 << (32u - sh8)) : w0;
		}
		const uint32_t done = head + nwords * 4u;
		if(tid < wb - done) dst[done + tid] = image_byte(done + tid);
	}
}

// ---- stage 5: exclusive scan of the frame sizes, then copy slots to their final offsets

__global__ void __launch_bounds__(1024) fa_scan_kernel(FaCfg c, FaWork w)
{
	// Workgroup b owns frames [1024 b, 1024 b + 1024): it sums every frame size ahead of them itself
	// (coalesced, independent loads that hit L2) and scans its own 1024, so that no workgroup waits for another.
	__shared__ unsigned long long wave_ahead[16];
	__shared__ uint32_t wave_own[16];
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6, b = blockIdx.x;
	unsigned long long ahead = 0;
#pragma unroll 8
	for(uint32_t i = 0; i < b; i++) ahead += w.frame_bytes[i * 1024u + tid];
	const uint32_t f = b * 1024u + tid;
	const uint32_t v = f < c.F ? w.frame_bytes[f] : 0u;
	uint32_t inc = v; // 64 frames of at most 2^24 bytes each fit 32 bits
#pragma unroll
	for(int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)inc, o, 64); if(lane >= (uint32_t)o) inc += t; }
#pragma unroll
	for(int o = 32; o >= 1; o >>= 1) ahead += (unsigned long long)__shfl_xor((long long)ahead, o, 64);
	if(lane == 63u) wave_own[wv] = inc;
	if(lane == 0u) wave_ahead[wv] = ahead;
	__syncthreads();
	unsigned long long base = 0;
#pragma unroll
	for(uint32_t k = 0; k < 16u; k++) { base += wave_ahead[k]; if(k < wv) base += wave_own[k]; }
	if(f < c.F) {
		w.offsets[f] = base + (inc - v);
		if(f + 1u == c.F) w.offsets[c.F] = base + inc;
	}
}

// One frame from its slot to its place, by a workgroup of 256: byte-granular destination, 4-byte loads where the source allows
__device__ __forceinline__ void fa_move_frame(const FaCfg &c, const FaWork &w, uint32_t f, uint8_t *__restrict__ out, unsigned long long out_cap, uint32_t *__restrict__ frame_bytes_out);

__global__ void __launch_bounds__(256) fa_gather_kernel(FaCfg c, FaWork w, uint8_t *__restrict__ out, unsigned long long out_cap, uint32_t *__restrict__ frame_bytes_out)
{
	fa_move_frame(c, w, blockIdx.x, out, out_cap, frame_bytes_out);
}

// After a pack kernel that wrote its frames in place: the frames that went to their slots because their place was not
// known in time (FaWork::lb, the flags behind the look-back states), sixteen frames to a workgroup.  Normally none.
__global__ void __launch_bounds__(256) fa_place_late_kernel(FaCfg c, FaWork w, uint8_t *__restrict__ out, unsigned long long out_cap, uint32_t *__restrict__ frame_bytes_out)
{
	const uint32_t *late = (const uint32_t*)(w.lb + c.F);
	for(uint32_t k = 0; k < 16u; k++) {
		const uint32_t f = blockIdx.x * 16u + k;
		if(f < c.F && late[f]) fa_move_frame(c, w, f, out, out_cap, frame_bytes_out);   // (the same for the whole workgroup)
	}
}

__device__ __forceinline__ void fa_move_frame(const FaCfg &c, const FaWork &w, uint32_t f, uint8_t *__restrict__ out, unsigned long long out_cap, uint32_t *__restrict__ frame_bytes_out)
{
	const uint32_t tid = threadIdx.x;
	const uint32_t nb = w.frame_bytes[f];
	const unsigned long long off = w.offsets[f] + (c.multi == 2u ? w.frame_shift[f] : 0u);
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
	uint32_t words = (uint32_t)((bits + 31u) / 32u) + 4u;
	// long blocks: 128 KB of bit buffer, flushed whenever the next 4096 tokens might not fit
	if(c->spill && words > 32768u) words = 32768u;
	if(bitbuf_words_out) *bitbuf_words_out = words;
	return ((c->spill ? 0u : (size_t)xs_words_for(c->N) * 4u) + (size_t)words * 4u + 15u) & ~(size_t)15u;
}

// the fast kernel applies to mono / stereo streams whose blocks fit 256 x 16 samples, whose
// partitions are whole multiples of a thread's 16 samples and whose predictors fit the register window
static bool fa_pack_fast_ok(const void *pcm, const FaCfg *c)
{
	return c->channels <= 2u && c->N <= (uint32_t)(PF_THREADS * PF_CH) && c->N % ((uint32_t)PF_CH << c->max_po) == 0 && c->max_lpc_order <= (uint32_t)HIST
		&& ((uintptr_t)pcm % 16u) == 0 && !(c->debug & 0x400u);
}

static uint32_t fa_pack_fast_words(const FaCfg *c)
{
	const uint32_t maxbps = c->bps + 1u;
	const size_t sub = (size_t)c->N * (maxbps + 1u) + 8u + 32u + 32u * (maxbps + 15u) + 64u;
	return (uint32_t)((16u * 8u + c->channels * sub + 64u + 31u) / 32u) + 4u;
}

extern "C" void fa_launch_pack_general(const void *pcm, const FaCfg *c, const FaWork *w, hipStream_t st, int large);

extern "C" void fa_launch_pack(const void *pcm, const FaCfg *c, const FaWork *w, hipStream_t st)
{
	if(fa_pack_fast_ok(pcm, c)) {
		const uint32_t words = fa_pack_fast_words(c);
		const size_t lds = ((size_t)words * 4u + 15u) & ~(size_t)15u;
		if(c->channels == 2) hipLaunchKernelGGL(fa_assign_kernel<2>, dim3(c->F), dim3(64), 0, st, *c, *w);
		else hipLaunchKernelGGL(fa_assign_kernel<1>, dim3(c->F), dim3(64), 0, st, *c, *w);
		// can any subframe take the 64-bit prediction sum (stream_encoder.c:3597-3603: bps + precision + ilog2(order) > 32)
		// or carry more than 24-bit samples?
		uint32_t lg = 0;
		while((2u << lg) <= (c->max_lpc_order ? c->max_lpc_order : 1u)) lg++;
		const uint32_t max_prec = c->prec_search ? FA_MAX_QLP_PRECISION : c->qlp_precision;
		const uint32_t max_bps = c->bps + (c->midside ? 1u : 0u);
		const bool narrow = max_bps <= 24u && max_bps + max_prec + lg <= 32u;
#define PF_LAUNCH(...) do { \
			(void)hipFuncSetAttribute((const void*)fa_pack_fast_kernel<__VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
			hipLaunchKernelGGL((fa_pack_fast_kernel<__VA_ARGS__>), dim3(c->F), dim3(PF_THREADS), lds, st, pcm, *c, *w, words); } while(0)
		const bool cd8 = narrow && c->N == 4096u && c->channels == 2u && c->bps == 16u && c->rate == 44100u && c->nsig == 4u && c->midside == 1u && c->multi == 0u
			&& c->max_po == 6u && c->debug == 0u && c->pcm_format == FLACAMD_PCM_S16 && c->direct_out == 1u;
		if(c->channels == 2 && c->pcm_format == FLACAMD_PCM_S16) { if(cd8) PF_LAUNCH(2, true, true, true); else if(narrow) PF_LAUNCH(2, true, true, false); else PF_LAUNCH(2, true, false, false); }
		else if(c->channels == 2) PF_LAUNCH(2, false, false, false);
		else PF_LAUNCH(1, false, false, false);
#undef PF_LAUNCH
		return;
	}
	fa_launch_pack_general(pcm, c, w, st, 0);
}

// the general kernel for any stream: one subframe at a time through LDS, flushed whenever the next tokens might not fit.
// large: as much bit buffer as LDS holds (FA_PACK_LARGE_WORDS at most), for frames far beyond the verbatim size: the
// 4096 tokens of one round have to fit between two flushes.
extern "C" void fa_launch_pack_general(const void *pcm, const FaCfg *c, const FaWork *w, hipStream_t st, int large)
{
	uint32_t words = 0;
	size_t lds = fa_pack_lds_bytes(c, &words);
	if(large) {
		const size_t xsb = c->spill ? 0u : (size_t)xs_words_for(c->N) * 4u;
		const size_t room = 150u * 1024u > xsb ? (150u * 1024u - xsb) / 4u : 0u;
		if(room > words) words = room > FA_PACK_LARGE_WORDS ? FA_PACK_LARGE_WORDS : (uint32_t)room;
		lds = (xsb + (size_t)words * 4u + 15u) & ~(size_t)15u;
	}
	if(c->spill) {
		(void)hipFuncSetAttribute((const void*)fa_pack_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		hipLaunchKernelGGL(fa_pack_kernel<true>, dim3(c->F), dim3(256), lds, st, pcm, *c, *w, words);
	}
	else {
		(void)hipFuncSetAttribute((const void*)fa_pack_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		hipLaunchKernelGGL(fa_pack_kernel<false>, dim3(c->F), dim3(256), lds, st, pcm, *c, *w, words);
	}
}

// the frame sizes' exclusive scan alone (frames already in place: FaCfg::direct_out)
extern "C" void fa_launch_offsets(const FaCfg *c, const FaWork *w, hipStream_t st)
{
	hipLaunchKernelGGL(fa_scan_kernel, dim3((c->F + 1023u) / 1024u), dim3(1024), 0, st, *c, *w);
	// frames written in place: those that went to their slots instead now have their offsets
	if(c->direct_out) hipLaunchKernelGGL(fa_place_late_kernel, dim3((c->F + 15u) / 16u), dim3(256), 0, st, *c, *w, w->out, (unsigned long long)w->out_cap, w->frame_bytes_out);
}

// can the pack stage put the frames of this batch straight into the caller's buffer?
extern "C" int fa_pack_direct_ok(const void *pcm, const FaCfg *c) { return fa_pack_fast_ok(pcm, c) && !(c->debug & 0x80000u); }

extern "C" void fa_launch_gather(const FaCfg *c, const FaWork *w, void *out, size_t out_cap, uint32_t *frame_bytes_out, hipStream_t st)
{
	hipLaunchKernelGGL(fa_scan_kernel, dim3((c->F + 1023u) / 1024u), dim3(1024), 0, st, *c, *w);
	hipLaunchKernelGGL(fa_gather_kernel, dim3(c->F), dim3(256), 0, st, *c, *w, (uint8_t*)out, (unsigned long long)out_cap, frame_bytes_out);
}
