/* TEST INFRASTRUCTURE ONLY - see flac_oracle.h.
 *
 * CPU restatement of the reference libFLAC 1.2.1 frame coder.  Written from scratch in a
 * "one frame in, one frame out" shape (the shape the HIP engine uses), not in the
 * reference's object/bitbuffer shape.  Every routine cites the reference lines whose
 * arithmetic it restates; paths are relative to /root/reference/src/libFLAC/.
 *
 * Build: gcc -O2 -std=c99 -ffp-contract=off (no -ffast-math, no -march=native): float
 * results must equal the reference build's, which has no FMA and no reassociation.
 */
#include "flac_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_LN2
#define M_LN2 0.69314718055994530942
#endif
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

#define MAX_FIXED_ORDER 4u
#define MIN_QLP_PRECISION 5u
#define MAX_QLP_PRECISION 15u
#define MAX_RICE_PARTITION_ORDER 15u

/* ------------------------------------------------------------------ small helpers */

static unsigned ilog2u(uint32_t v) { unsigned l = 0; while(v >>= 1) l++; return l; } /* bitmath.c:38-46 */
static unsigned umin(unsigned a, unsigned b) { return a < b ? a : b; }
static unsigned umax(unsigned a, unsigned b) { return a > b ? a : b; }
static uint64_t u64min(uint64_t a, uint64_t b) { return a < b ? a : b; }

/* ------------------------------------------------------------------ CRC (crc.c:124-142) */
/* CRC-8: poly x^8+x^2+x+1 (0x07); CRC-16: poly x^16+x^15+x^2+1 (0x8005); init 0, MSB first. */
uint8_t fo_crc8(const uint8_t *d, size_t n)
{
	uint8_t c = 0;
	while(n--) { int b; c ^= *d++; for(b = 0; b < 8; b++) c = (uint8_t)((c & 0x80) ? (c << 1) ^ 0x07 : (c << 1)); }
	return c;
}
uint16_t fo_crc16(const uint8_t *d, size_t n)
{
	uint16_t c = 0;
	while(n--) { int b; c ^= (uint16_t)(*d++ << 8); for(b = 0; b < 8; b++) c = (uint16_t)((c & 0x8000) ? (c << 1) ^ 0x8005 : (c << 1)); }
	return c;
}

/* ------------------------------------------------------------------ MD5 (md5.c) */
typedef struct { uint32_t h[4]; uint64_t nbytes; uint8_t buf[64]; } md5_t;
static const uint32_t md5_k[64] = {
	0xd76aa478,0xe8c7b756,0x242070db,0xc1bdceee,0xf57c0faf,0x4787c62a,0xa8304613,0xfd469501,
	0x698098d8,0x8b44f7af,0xffff5bb1,0x895cd7be,0x6b901122,0xfd987193,0xa679438e,0x49b40821,
	0xf61e2562,0xc040b340,0x265e5a51,0xe9b6c7aa,0xd62f105d,0x02441453,0xd8a1e681,0xe7d3fbc8,
	0x21e1cde6,0xc33707d6,0xf4d50d87,0x455a14ed,0xa9e3e905,0xfcefa3f8,0x676f02d9,0x8d2a4c8a,
	0xfffa3942,0x8771f681,0x6d9d6122,0xfde5380c,0xa4beea44,0x4bdecfa9,0xf6bb4b60,0xbebfbc70,
	0x289b7ec6,0xeaa127fa,0xd4ef3085,0x04881d05,0xd9d4d039,0xe6db99e5,0x1fa27cf8,0xc4ac5665,
	0xf4292244,0x432aff97,0xab9423a7,0xfc93a039,0x655b59c3,0x8f0ccc92,0xffeff47d,0x85845dd1,
	0x6fa87e4f,0xfe2ce6e0,0xa3014314,0x4e0811a1,0xf7537e82,0xbd3af235,0x2ad7d2bb,0xeb86d391 };
static const uint8_t md5_s[64] = { 7,12,17,22,7,12,17,22,7,12,17,22,7,12,17,22, 5,9,14,20,5,9,14,20,5,9,14,20,5,9,14,20,
	4,11,16,23,4,11,16,23,4,11,16,23,4,11,16,23, 6,10,15,21,6,10,15,21,6,10,15,21,6,10,15,21 };
static void md5_block(md5_t *m, const uint8_t *p)
{
	uint32_t w[16], a = m->h[0], b = m->h[1], c = m->h[2], d = m->h[3]; int i;
	for(i = 0; i < 16; i++) w[i] = (uint32_t)p[4*i] | (uint32_t)p[4*i+1] << 8 | (uint32_t)p[4*i+2] << 16 | (uint32_t)p[4*i+3] << 24;
	for(i = 0; i < 64; i++) {
		uint32_t f, t; int g;
		if(i < 16) { f = (b & c) | (~b & d); g = i; }
		else if(i < 32) { f = (d & b) | (~d & c); g = (5*i + 1) & 15; }
		else if(i < 48) { f = b ^ c ^ d; g = (3*i + 5) & 15; }
		else { f = c ^ (b | ~d); g = (7*i) & 15; }
		t = d; d = c; c = b;
		f += a + md5_k[i] + w[g];
		b += (f << md5_s[i]) | (f >> (32 - md5_s[i]));
		a = t;
	}
	m->h[0] += a; m->h[1] += b; m->h[2] += c; m->h[3] += d;
}
static void md5_init(md5_t *m) { m->h[0] = 0x67452301; m->h[1] = 0xefcdab89; m->h[2] = 0x98badcfe; m->h[3] = 0x10325476; m->nbytes = 0; }
static void md5_update(md5_t *m, const uint8_t *p, size_t n)
{
	size_t fill = (size_t)(m->nbytes & 63);
	m->nbytes += n;
	if(fill) { size_t t = 64 - fill; if(t > n) t = n; memcpy(m->buf + fill, p, t); p += t; n -= t; if(fill + t < 64) return; md5_block(m, m->buf); }
	while(n >= 64) { md5_block(m, p); p += 64; n -= 64; }
	if(n) memcpy(m->buf, p, n);
}
static void md5_final(md5_t *m, uint8_t out[16])
{
	uint64_t bits = m->nbytes * 8; uint8_t pad[72]; size_t fill = (size_t)(m->nbytes & 63), n = (fill < 56 ? 56 - fill : 120 - fill); int i;
	memset(pad, 0, sizeof pad); pad[0] = 0x80;
	for(i = 0; i < 8; i++) pad[n + i] = (uint8_t)(bits >> (8*i));
	md5_update(m, pad, n + 8);
	for(i = 0; i < 16; i++) out[i] = (uint8_t)(m->h[i >> 2] >> (8 * (i & 3)));
}
/* md5.c:273-389 format_input_: channels interleaved, each sample little-endian in
 * (bps+7)/8 bytes; md5.c:394-420 accumulates that byte stream. */
static void md5_pcm_update(md5_t *m, const int32_t *pcm, uint64_t nsamples, uint32_t channels, uint32_t bps)
{
	const unsigned bytes = (bps + 7) / 8; uint8_t tmp[4096]; size_t fill = 0; uint64_t i, total = nsamples * channels;
	for(i = 0; i < total; i++) {
		uint32_t v = (uint32_t)pcm[i]; unsigned b;
		for(b = 0; b < bytes; b++) tmp[fill++] = (uint8_t)(v >> (8*b));
		if(fill + 4 > sizeof tmp) { md5_update(m, tmp, fill); fill = 0; }
	}
	if(fill) md5_update(m, tmp, fill);
}
void fo_md5_pcm(const int32_t *pcm, uint64_t nsamples, uint32_t channels, uint32_t bps, uint8_t digest[16])
{
	md5_t m; md5_init(&m); md5_pcm_update(&m, pcm, nsamples, channels, bps); md5_final(&m, digest);
}

/* ------------------------------------------------------------------ presets & parameter resolution */

void fo_params_preset(fo_params *p, unsigned level)
{
	/* stream_encoder.c:127-148: {mid_side, loose, max_lpc, qlp_prec, prec_search, escape, exhaustive, min_po, max_po, search_dist} */
	static const struct { uint8_t ms, loose, lpc, exh, maxpo; } t[9] = {
		{0,0,0,0,3},{1,1,0,0,3},{1,0,0,0,3},{0,0,6,0,4},{1,1,8,0,4},{1,0,8,0,5},{1,0,8,0,6},{1,0,8,1,6},{1,0,12,1,6} };
	if(level > 8) level = 8;
	p->do_mid_side = t[level].ms; p->loose_mid_side = t[level].loose; p->max_lpc_order = t[level].lpc;
	p->qlp_coeff_precision = 0; p->do_qlp_coeff_prec_search = 0; p->do_exhaustive_model_search = t[level].exh;
	p->min_residual_partition_order = 0; p->max_residual_partition_order = t[level].maxpo;
	/* stream_encoder.c:1481-1483: every preset uses one tukey(0.5) window */
	p->num_apodizations = 1; p->apod_type[0] = FO_WIN_TUKEY; p->apod_param[0] = 0.5f;
}

int fo_params_resolve(fo_params *p)
{
	/* stream_encoder.c:667-678 */
	if(p->channels == 0 || p->channels > FO_MAX_CHANNELS) return 1;
	if(p->channels != 2) { p->do_mid_side = 0; p->loose_mid_side = 0; }
	else if(!p->do_mid_side) p->loose_mid_side = 0;
	if(p->bits_per_sample >= 32) p->do_mid_side = 0;
	if(p->bits_per_sample < 4 || p->bits_per_sample > 24) return 2; /* :680 (reference codec max is 24) */
	if(p->sample_rate == 0 || p->sample_rate > 655350) return 3;
	/* :686-700 */
	if(p->blocksize == 0) p->blocksize = p->max_lpc_order == 0 ? 1152 : 4096;
	if(p->blocksize < 16 || p->blocksize > 65535) return 4;
	if(p->max_lpc_order > FO_MAX_LPC_ORDER) return 5;
	if(p->blocksize < p->max_lpc_order) return 6;
	/* :702-735 automatic coefficient precision */
	if(p->qlp_coeff_precision == 0) {
		const unsigned bs = p->blocksize;
		if(p->bits_per_sample < 16) p->qlp_coeff_precision = umax(MIN_QLP_PRECISION, 2 + p->bits_per_sample / 2);
		else if(p->bits_per_sample == 16)
			p->qlp_coeff_precision = bs <= 192 ? 7 : bs <= 384 ? 8 : bs <= 576 ? 9 : bs <= 1152 ? 10 : bs <= 2304 ? 11 : bs <= 4608 ? 12 : 13;
		else
			p->qlp_coeff_precision = bs <= 384 ? MAX_QLP_PRECISION - 2 : bs <= 1152 ? MAX_QLP_PRECISION - 1 : MAX_QLP_PRECISION;
	}
	else if(p->qlp_coeff_precision < MIN_QLP_PRECISION || p->qlp_coeff_precision > MAX_QLP_PRECISION) return 7;
	/* :763-766 */
	if(p->max_residual_partition_order > MAX_RICE_PARTITION_ORDER) p->max_residual_partition_order = MAX_RICE_PARTITION_ORDER;
	if(p->min_residual_partition_order >= p->max_residual_partition_order) p->min_residual_partition_order = p->max_residual_partition_order;
	if(p->num_apodizations == 0) { p->num_apodizations = 1; p->apod_type[0] = FO_WIN_TUKEY; p->apod_param[0] = 0.5f; }
	return 0;
}

/* ------------------------------------------------------------------ windows (window.c:49-223) */
/* Expression shapes (float vs double sub-expressions) follow the reference exactly,
 * because the stored float depends on them. */
void fo_window(uint32_t type, float param, int32_t L, float *w)
{
	const int32_t N = L - 1; int32_t n;
	switch(type) {
	case FO_WIN_BARTLETT: /* :49-66 */
		if(L & 1) { for(n = 0; n <= N/2; n++) w[n] = 2.0f * n / (float)N; for(; n <= N; n++) w[n] = 2.0f - 2.0f * n / (float)N; }
		else { for(n = 0; n <= L/2-1; n++) w[n] = 2.0f * n / (float)N; for(; n <= N; n++) w[n] = 2.0f - 2.0f * (N-n) / (float)N; }
		break;
	case FO_WIN_BARTLETT_HANN: /* :68-75 */
		for(n = 0; n < L; n++) w[n] = (float)(0.62f - 0.48f * fabs((float)n/(float)N+0.5f) + 0.38f * cos(2.0f * M_PI * ((float)n/(float)N+0.5f)));
		break;
	case FO_WIN_BLACKMAN: /* :77-84 */
		for(n = 0; n < L; n++) w[n] = (float)(0.42f - 0.5f * cos(2.0f * M_PI * n / N) + 0.08f * cos(4.0f * M_PI * n / N));
		break;
	case FO_WIN_BLACKMAN_HARRIS_4TERM: /* :87-94 */
		for(n = 0; n <= N; n++) w[n] = (float)(0.35875f - 0.48829f * cos(2.0f * M_PI * n / N) + 0.14128f * cos(4.0f * M_PI * n / N) - 0.01168f * cos(6.0f * M_PI * n / N));
		break;
	case FO_WIN_CONNES: { /* :96-107 */
		const double N2 = (double)N / 2.;
		for(n = 0; n <= N; n++) { double k = ((double)n - N2) / N2; k = 1.0f - k * k; w[n] = (float)(k * k); }
		break; }
	case FO_WIN_FLATTOP: /* :109-116 */
		for(n = 0; n < L; n++) w[n] = (float)(1.0f - 1.93f * cos(2.0f * M_PI * n / N) + 1.29f * cos(4.0f * M_PI * n / N) - 0.388f * cos(6.0f * M_PI * n / N) + 0.0322f * cos(8.0f * M_PI * n / N));
		break;
	case FO_WIN_GAUSS: { /* :118-128 */
		const double N2 = (double)N / 2.;
		for(n = 0; n <= N; n++) { const double k = ((double)n - N2) / (param * N2); w[n] = (float)exp(-0.5f * k * k); }
		break; }
	case FO_WIN_HAMMING: /* :130-137 */
		for(n = 0; n < L; n++) w[n] = (float)(0.54f - 0.46f * cos(2.0f * M_PI * n / N));
		break;
	case FO_WIN_HANN: /* :139-146 */
		for(n = 0; n < L; n++) w[n] = (float)(0.5f - 0.5f * cos(2.0f * M_PI * n / N));
		break;
	case FO_WIN_KAISER_BESSEL: /* :148-155 */
		for(n = 0; n < L; n++) w[n] = (float)(0.402f - 0.498f * cos(2.0f * M_PI * n / N) + 0.098f * cos(4.0f * M_PI * n / N) - 0.001f * cos(6.0f * M_PI * n / N));
		break;
	case FO_WIN_NUTTALL: /* :157-164 */
		for(n = 0; n < L; n++) w[n] = (float)(0.3635819f - 0.4891775f*cos(2.0f*M_PI*n/N) + 0.1365995f*cos(4.0f*M_PI*n/N) - 0.0106411f*cos(6.0f*M_PI*n/N));
		break;
	case FO_WIN_RECTANGLE: /* :166-172 */
		for(n = 0; n < L; n++) w[n] = 1.0f;
		break;
	case FO_WIN_TRIANGLE: /* :174-190; "L+1/2" is L+0 in C: the odd-length branch never reaches its second loop */
		if(L & 1) { for(n = 1; n <= L+1/2; n++) w[n-1] = 2.0f * n / ((float)L + 1.0f); for(; n <= L; n++) w[n-1] = - (float)(2 * (L - n + 1)) / ((float)L + 1.0f); }
		else { for(n = 1; n <= L/2; n++) w[n-1] = 2.0f * n / (float)L; for(; n <= L; n++) w[n-1] = ((float)(2 * (L - n)) + 1.0f) / (float)L; }
		break;
	case FO_WIN_WELCH: { /* :213-223 */
		const double N2 = (double)N / 2.;
		for(n = 0; n <= N; n++) { const double k = ((double)n - N2) / N2; w[n] = (float)(1.0f - k * k); }
		break; }
	case FO_WIN_TUKEY: default: /* :192-211 */
		if(param <= 0.0) fo_window(FO_WIN_RECTANGLE, 0, L, w);
		else if(param >= 1.0) fo_window(FO_WIN_HANN, 0, L, w);
		else {
			const int32_t Np = (int32_t)(param / 2.0f * L) - 1;
			fo_window(FO_WIN_RECTANGLE, 0, L, w);
			if(Np > 0) for(n = 0; n <= Np; n++) {
				w[n] = (float)(0.5f - 0.5f * cos(M_PI * n / Np));
				w[L-Np-1+n] = (float)(0.5f - 0.5f * cos(M_PI * (n+Np) / Np));
			}
		}
		break;
	}
}

/* ------------------------------------------------------------------ MSB-first bit sink */
typedef struct { uint8_t *p; size_t cap; uint64_t nbits; int overflow; } bitsink;

static void put_bits(bitsink *s, uint32_t v, unsigned n) /* bitwriter.c:343-380 semantics: n low bits of v, MSB first */
{
	while(n) {
		const unsigned room = 8 - (unsigned)(s->nbits & 7), take = n < room ? n : room;
		const size_t byte = (size_t)(s->nbits >> 3);
		const uint32_t chunk = (v >> (n - take)) & ((1u << take) - 1u);
		if(byte >= s->cap) { s->overflow = 1; return; }
		if(room == 8) s->p[byte] = 0;
		s->p[byte] |= (uint8_t)(chunk << (room - take));
		s->nbits += take; n -= take;
	}
}
static void put_sbits(bitsink *s, int32_t v, unsigned n) { put_bits(s, n < 32 ? ((uint32_t)v & ((1u << n) - 1u)) : (uint32_t)v, n); } /* bitwriter.c write_raw_int32 */
static void put_zeros(bitsink *s, uint32_t n) { while(n > 24) { put_bits(s, 0, 24); n -= 24; } if(n) put_bits(s, 0, n); }
static void put_unary(bitsink *s, uint32_t v) { put_zeros(s, v); put_bits(s, 1, 1); } /* v zeros then a one */

/* bitwriter.c:784-832 "UTF-8" coding of the frame number (up to 31 bits here) */
static void put_utf8(bitsink *s, uint32_t v)
{
	if(v < 0x80) put_bits(s, v, 8);
	else if(v < 0x800) { put_bits(s, 0xC0 | (v >> 6), 8); put_bits(s, 0x80 | (v & 0x3F), 8); }
	else if(v < 0x10000) { put_bits(s, 0xE0 | (v >> 12), 8); put_bits(s, 0x80 | ((v >> 6) & 0x3F), 8); put_bits(s, 0x80 | (v & 0x3F), 8); }
	else if(v < 0x200000) { put_bits(s, 0xF0 | (v >> 18), 8); put_bits(s, 0x80 | ((v >> 12) & 0x3F), 8); put_bits(s, 0x80 | ((v >> 6) & 0x3F), 8); put_bits(s, 0x80 | (v & 0x3F), 8); }
	else if(v < 0x4000000) { put_bits(s, 0xF8 | (v >> 24), 8); put_bits(s, 0x80 | ((v >> 18) & 0x3F), 8); put_bits(s, 0x80 | ((v >> 12) & 0x3F), 8); put_bits(s, 0x80 | ((v >> 6) & 0x3F), 8); put_bits(s, 0x80 | (v & 0x3F), 8); }
	else { put_bits(s, 0xFC | (v >> 30), 8); put_bits(s, 0x80 | ((v >> 24) & 0x3F), 8); put_bits(s, 0x80 | ((v >> 18) & 0x3F), 8); put_bits(s, 0x80 | ((v >> 12) & 0x3F), 8); put_bits(s, 0x80 | ((v >> 6) & 0x3F), 8); put_bits(s, 0x80 | (v & 0x3F), 8); }
}

/* ------------------------------------------------------------------ analysis primitives */

/* stream_encoder.c:4113-4135: trailing zero bits common to every sample; the signal is
 * shifted down in place. */
static unsigned strip_wasted_bits(int32_t *x, unsigned n)
{
	uint32_t acc = 0; unsigned i, shift = 0;
	for(i = 0; i < n; i++) acc |= (uint32_t)x[i];
	if(acc == 0) return 0;
	while(!(acc & 1)) { acc >>= 1; shift++; }
	if(shift) for(i = 0; i < n; i++) x[i] >>= shift;
	return shift;
}

/* fixed.c:224-350: sums of |k-th difference| for k = 0..4 over samples [4, n).
 * narrow: 32-bit wrapping accumulators (fixed.c:234), else 64-bit (:296); the choice is
 * stream_encoder.c:894,949 (bps + ilog2(blocksize) + 1 > 30 -> wide).
 * Differences themselves wrap in 32 bits like the reference's FLAC__int32 temporaries. */
static unsigned fixed_best_order(const int32_t *x, unsigned n, int wide, float bits_per_sample[5], uint64_t err_out[5])
{
	uint64_t e[5] = {0,0,0,0,0}; unsigned i, k, order; const unsigned len = n - MAX_FIXED_ORDER;
	for(i = MAX_FIXED_ORDER; i < n; i++) {
		const uint32_t a = (uint32_t)x[i], b = (uint32_t)x[i-1], c = (uint32_t)x[i-2], d = (uint32_t)x[i-3], f = (uint32_t)x[i-4];
		const int32_t d0 = (int32_t)a, d1 = (int32_t)(a - b), d2 = (int32_t)(a - 2*b + c), d3 = (int32_t)(a - 3*b + 3*c - d), d4 = (int32_t)(a - 4*b + 6*c - 4*d + f);
		const int32_t dd[5] = { d0, d1, d2, d3, d4 };
		for(k = 0; k < 5; k++) { const uint32_t m = dd[k] < 0 ? (uint32_t)0 - (uint32_t)dd[k] : (uint32_t)dd[k]; e[k] += m; }
	}
	if(!wide) for(k = 0; k < 5; k++) e[k] &= 0xffffffffu;
	/* fixed.c:245-254 */
	if(e[0] < u64min(u64min(u64min(e[1], e[2]), e[3]), e[4])) order = 0;
	else if(e[1] < u64min(u64min(e[2], e[3]), e[4])) order = 1;
	else if(e[2] < u64min(e[3], e[4])) order = 2;
	else if(e[3] < e[4]) order = 3;
	else order = 4;
	/* fixed.c:265-269 */
	for(k = 0; k < 5; k++) bits_per_sample[k] = (float)((e[k] > 0) ? log(M_LN2 * (double)e[k] / (double)len) / M_LN2 : 0.0);
	if(err_out) for(k = 0; k < 5; k++) err_out[k] = e[k];
	return order;
}

/* fixed.c:352-393 with 32-bit wrap */
static void fixed_residual(const int32_t *x, unsigned n, unsigned order, int32_t *res)
{
	unsigned i;
	for(i = order; i < n; i++) {
		const uint32_t a = (uint32_t)x[i];
		uint32_t r;
		switch(order) {
		case 0: r = a; break;
		case 1: r = a - (uint32_t)x[i-1]; break;
		case 2: r = a - 2u*(uint32_t)x[i-1] + (uint32_t)x[i-2]; break;
		case 3: r = a - 3u*(uint32_t)x[i-1] + 3u*(uint32_t)x[i-2] - (uint32_t)x[i-3]; break;
		default: r = a - 4u*(uint32_t)x[i-1] + 6u*(uint32_t)x[i-2] - 4u*(uint32_t)x[i-3] + (uint32_t)x[i-4]; break;
		}
		res[i - order] = (int32_t)r;
	}
}

/* lpc.c:56-61 then lpc.c:63-110: float window multiply, float autocorrelation with the
 * products accumulated strictly in sample order (no FMA, no reassociation). */
static void windowed_autocorrelation(const int32_t *x, const float *win, unsigned n, unsigned lag, float *scratch, float *autoc)
{
	unsigned i, l;
	for(i = 0; i < n; i++) scratch[i] = (float)x[i] * win[i];
	for(l = 0; l < lag; l++) autoc[l] = 0.0f;
	for(i = 0; i < n; i++) {
		const float d = scratch[i]; const unsigned lim = (n - i < lag) ? n - i : lag;
		for(l = 0; l < lim; l++) { const float prod = d * scratch[i + l]; autoc[l] = autoc[l] + prod; }
	}
}

/* lpc.c:112-154 Levinson-Durbin in double; coefficients of every order kept as float.
 * Returns the usable max order (smaller when the error hits exactly 0, lpc.c:149). */
static unsigned levinson(const float *autoc, unsigned max_order, float coef[][FO_MAX_LPC_ORDER], double *err_out)
{
	double lpc[FO_MAX_LPC_ORDER], err = autoc[0]; unsigned i, j;
	for(i = 0; i < max_order; i++) {
		double r = -autoc[i+1];
		for(j = 0; j < i; j++) r -= lpc[j] * autoc[i-j];
		r /= err;
		lpc[i] = r;
		for(j = 0; j < (i >> 1); j++) { const double t = lpc[j]; lpc[j] += r * lpc[i-1-j]; lpc[i-1-j] += r * t; }
		if(i & 1) lpc[j] += lpc[j] * r;
		err *= (1.0 - r * r);
		for(j = 0; j <= i; j++) coef[i][j] = (float)(-lpc[j]);
		err_out[i] = err;
		if(err == 0.0) return i + 1;
	}
	return max_order;
}

/* lpc.c:1325-1351 */
static double expected_bits_with_scale(double lpc_error, double scale)
{
	if(lpc_error > 0.0) { const double b = 0.5 * log(scale * lpc_error) / M_LN2; return b >= 0.0 ? b : 0.0; }
	return lpc_error < 0.0 ? 1e32 : 0.0;
}
static double expected_bits(double lpc_error, unsigned total_samples) { return expected_bits_with_scale(lpc_error, 0.5 * M_LN2 * M_LN2 / (double)total_samples); }

/* lpc.c:1353-1375 */
static unsigned guess_lpc_order(const double *err, unsigned max_order, unsigned total_samples, unsigned overhead_bits_per_order)
{
	const double scale = 0.5 * M_LN2 * M_LN2 / (double)total_samples; double best = (double)0xffffffffu; unsigned best_i = 0, i;
	for(i = 0; i < max_order; i++) {
		const double bits = expected_bits_with_scale(err[i], scale) * (double)(total_samples - (i+1)) + (double)((i+1) * overhead_bits_per_order);
		if(bits < best) { best = bits; best_i = i; }
	}
	return best_i + 1;
}

/* lpc.c:156-263. Returns 0 ok, 1 shift below range, 2 all-zero coefficients. */
static int quantize_coefficients(const float *c, unsigned order, unsigned precision, int32_t *q, int *shift_out)
{
	double cmax = 0.0, error = 0.0; int32_t qmax, qmin; int shift, log2cmax; unsigned i;
	precision--; qmax = (int32_t)1 << precision; qmin = -qmax; qmax--;
	for(i = 0; i < order; i++) { const double d = fabs((double)c[i]); if(d > cmax) cmax = d; }
	if(cmax <= 0.0) return 2;
	(void)frexp(cmax, &log2cmax); log2cmax--;
	shift = (int)precision - log2cmax - 1;
	if(shift > 15) shift = 15; else if(shift < -16) return 1;
	if(shift >= 0) {
		for(i = 0; i < order; i++) {
			int32_t v;
			error += c[i] * (1 << shift);           /* float * int -> float product, widened (lpc.c:202) */
			v = error >= 0.0 ? (int32_t)(error + 0.5) : (int32_t)(error - 0.5);
			if(v > qmax) v = qmax; else if(v < qmin) v = qmin;
			error -= v; q[i] = v;
		}
	}
	else {
		const int nshift = -shift;
		for(i = 0; i < order; i++) {
			int32_t v;
			error += c[i] / (1 << nshift);          /* lpc.c:237 */
			v = error >= 0.0 ? (int32_t)(error + 0.5) : (int32_t)(error - 0.5);
			if(v > qmax) v = qmax; else if(v < qmin) v = qmin;
			error -= v; q[i] = v;
		}
		shift = 0;
	}
	*shift_out = shift;
	return 0;
}

/* lpc.c:265-529 (32-bit wrapping sum) and :531-791 (64-bit sum); selector stream_encoder.c:3597-3603 */
static void lpc_residual(const int32_t *x, unsigned n, const int32_t *q, unsigned order, int shift, int wide, int32_t *res)
{
	unsigned i, j;
	for(i = order; i < n; i++) {
		if(wide) {
			int64_t sum = 0;
			for(j = 0; j < order; j++) sum += (int64_t)q[j] * (int64_t)x[i-1-j];
			res[i - order] = (int32_t)((uint32_t)x[i] - (uint32_t)(int32_t)(sum >> shift));
		}
		else {
			uint32_t sum = 0;
			for(j = 0; j < order; j++) sum += (uint32_t)q[j] * (uint32_t)x[i-1-j];
			res[i - order] = (int32_t)((uint32_t)x[i] - (uint32_t)((int32_t)sum >> shift));
		}
	}
}

/* ------------------------------------------------------------------ Rice partition search */

typedef struct {
	unsigned order;       /* partition order */
	unsigned rice2;
	uint32_t bits;        /* estimated bits of the whole residual section incl. method+order fields */
	uint32_t *param;      /* 1 << order entries */
} rice_choice;

/* format.c:548-561 */
static unsigned limit_partition_order(unsigned limit, unsigned blocksize, unsigned predictor_order)
{
	while(limit > 0 && (blocksize >> limit) <= predictor_order) limit--;
	return limit;
}

/* stream_encoder.c:3923-3945 (all arithmetic in unsigned 32 bits, as there) */
static uint32_t rice_bits_estimate(unsigned k, uint32_t nsamples, uint64_t abs_sum)
{
	return 4u + (1u + k) * nsamples + (k ? (uint32_t)(abs_sum >> (k - 1)) : (uint32_t)(abs_sum << 1)) - (nsamples >> 1);
}

/* stream_encoder.c:3671-3766 find_best_partition_order_, with
 * :3779-3847 precompute_partition_info_sums_ and :3948-4111 set_partitioned_rice_.
 * sums[] scratch needs 2<<max_po entries, params scratch 2 * (1<<max_po). */
static uint32_t best_rice_partitioning(const int32_t *res, unsigned residual_samples, unsigned predictor_order,
	unsigned suggested_k, unsigned k_limit, unsigned min_po, unsigned max_po, unsigned bps,
	uint64_t *sums, uint32_t *param_scratch, rice_choice *out)
{
	const unsigned blocksize = residual_samples + predictor_order;
	unsigned po, parts, i, have = 0, cur = 0;
	uint32_t best_bits = 0; unsigned best_po = 0;
	uint64_t *level;

	max_po = limit_partition_order(max_po, blocksize, predictor_order);
	min_po = umin(min_po, max_po);

	/* sums at the finest order (:3803-3830): 32-bit wrapping accumulator when
	 * ilog2(partition length) + bps < 32, else 64-bit */
	{
		const unsigned plen = blocksize >> max_po; const int narrow = ilog2u(plen) + bps < 32;
		unsigned s = 0, end = 0u - predictor_order;
		parts = 1u << max_po;
		for(i = 0; i < parts; i++) {
			uint64_t acc = 0;
			end += plen;
			for(; s < end; s++) { const int32_t r = res[s]; acc += (r < 0) ? (uint32_t)0 - (uint32_t)r : (uint32_t)r; if(narrow) acc &= 0xffffffffu; }
			sums[i] = acc;
		}
	}
	/* merge pairs for the coarser orders (:3832-3846), 64-bit adds */
	{
		unsigned from = 0, to = parts, p = parts; int o;
		for(o = (int)max_po - 1; o >= (int)min_po; o--) { p >>= 1; for(i = 0; i < p; i++) { sums[to++] = sums[from] + sums[from+1]; from += 2; } }
	}

	level = sums;
	for(po = max_po; ; po--) {
		uint32_t *param = param_scratch + (cur ? (1u << max_po) : 0);
		uint32_t bits = 2u + 4u; int usable = 1;
		parts = 1u << po;
		if(po == 0) { /* :3982-4029: the caller's suggested parameter, not one derived from the sum */
			param[0] = suggested_k;
			bits += rice_bits_estimate(suggested_k, residual_samples, level[0]);
		}
		else {
			for(i = 0; i < parts; i++) { /* :4035-4106 */
				unsigned n = blocksize >> po, k = 0; uint64_t m;
				if(i == 0) { if(n <= predictor_order) { usable = 0; break; } n -= predictor_order; }
				for(m = n; m < level[i]; m <<= 1) k++;  /* :4052 */
				if(k >= k_limit) k = k_limit - 1;
				param[i] = k;
				bits += rice_bits_estimate(k, n, level[i]);
			}
		}
		if(!usable) break;
		level += parts;
		if(!have || bits < best_bits) { best_bits = bits; best_po = po; have = 1; cur = !cur; } /* :3729 strict '<' keeps the finer order on ties */
		if(po == min_po) break;
	}
	{
		const uint32_t *win = param_scratch + (cur ? 0 : (1u << max_po)); /* the buffer NOT about to be overwritten */
		out->order = best_po; out->bits = best_bits; out->rice2 = 0;
		for(i = 0; i < (1u << best_po); i++) { out->param[i] = win[i]; if(win[i] >= 15) out->rice2 = 1; } /* :3757-3762 */
	}
	return best_bits;
}

/* ------------------------------------------------------------------ one subframe: candidate search */

typedef struct {
	unsigned type, order, wasted_bits, bps;
	unsigned qlp_precision; int qlp_shift; int32_t qlp[FO_MAX_LPC_ORDER];
	uint32_t bits;
	rice_choice rice;
	int32_t *residual;     /* residual_samples values (fixed / lpc) */
	const int32_t *signal; /* the (wasted-bit-stripped) signal */
} subframe_plan;

typedef struct {
	const fo_params *p;
	const float *windows;      /* num_apodizations x stream blocksize */
	unsigned stream_blocksize;
	int wide_fixed;            /* stream_encoder.c:894 */
	float *fscratch;
	int32_t *res_tmp;
	uint64_t *sums;
	uint32_t *param_scratch, *param_tmp;
} coder_ctx;

static void plan_take(subframe_plan *best, const subframe_plan *cand, unsigned residual_samples, unsigned nparams)
{
	uint32_t *param = best->rice.param; int32_t *resbuf = best->residual;
	*best = *cand;
	best->rice.param = param; best->residual = resbuf;
	if(cand->rice.param) memcpy(param, cand->rice.param, sizeof(uint32_t) * nparams);
	if(cand->residual) memcpy(resbuf, cand->residual, sizeof(int32_t) * residual_samples);
}

/* stream_encoder.c:3169-3402 process_subframe_.  Candidates are tried in the reference's
 * order and a candidate replaces the best only when strictly smaller. */
static void choose_subframe(coder_ctx *c, const int32_t *x, unsigned n, unsigned bps, unsigned wasted,
	unsigned min_po, unsigned max_po, subframe_plan *best, float *autoc_out, uint64_t *fixed_err_out)
{
	const fo_params *p = c->p;
	const unsigned k_limit = p->bits_per_sample > 16 ? 31 : 15; /* :3200 uses the STREAM bps */
	subframe_plan cand;
	float fixed_bits[5]; unsigned guess_fixed;

	memset(&cand, 0, sizeof cand);
	cand.wasted_bits = wasted; cand.bps = bps; cand.signal = x;
	best->wasted_bits = wasted; best->bps = bps; best->signal = x;

	/* verbatim baseline (:3205-3209, estimate :3660) */
	best->type = FO_SUB_VERBATIM; best->order = 0;
	best->bits = (p->disable_verbatim && n >= MAX_FIXED_ORDER) ? 0xffffffffu : 8u + wasted + n * bps;

	if(n >= MAX_FIXED_ORDER) {
		int constant = 0; unsigned i;
		guess_fixed = fixed_best_order(x, n, c->wide_fixed, fixed_bits, fixed_err_out);
		if(!p->disable_constant && fixed_bits[1] == 0.0f) { /* :3215-3232 */
			constant = 1;
			for(i = 1; i < n; i++) if(x[i] != x[0]) { constant = 0; break; }
		}
		if(constant) {
			const uint32_t bits = 8u + wasted + bps; /* :3487 */
			if(bits < best->bits) { best->type = FO_SUB_CONSTANT; best->order = 0; best->bits = bits; }
		}
		else {
			if(!p->disable_fixed || (p->max_lpc_order == 0 && best->bits == 0xffffffffu)) { /* :3241-3293 */
				unsigned lo, hi, o;
				if(p->do_exhaustive_model_search) { lo = 0; hi = MAX_FIXED_ORDER; } else lo = hi = guess_fixed;
				if(hi >= n) hi = n - 1;
				for(o = lo; o <= hi; o++) {
					unsigned k; uint32_t rbits;
					if(fixed_bits[o] >= (float)bps) continue;
					k = (fixed_bits[o] > 0.0) ? (unsigned)(fixed_bits[o] + 0.5) : 0; /* float + double 0.5 -> double, :3256 */
					k++;
					if(k >= k_limit) k = k_limit - 1;
					fixed_residual(x, n, o, c->res_tmp);
					cand.type = FO_SUB_FIXED; cand.order = o; cand.residual = c->res_tmp; cand.rice.param = c->param_tmp;
					rbits = best_rice_partitioning(c->res_tmp, n - o, o, k, k_limit, min_po, max_po, bps, c->sums, c->param_scratch, &cand.rice);
					cand.bits = 8u + wasted + o * bps + rbits; /* :3550 */
					if(cand.bits < best->bits) plan_take(best, &cand, n - o, 1u << cand.rice.order);
				}
			}
			if(p->max_lpc_order > 0) { /* :3297-3387 */
				unsigned max_order = p->max_lpc_order >= n ? n - 1 : p->max_lpc_order;
				if(max_order > 0) {
					unsigned a;
					for(a = 0; a < p->num_apodizations; a++) {
						float autoc[FO_MAX_LPC_ORDER + 1]; float coef[FO_MAX_LPC_ORDER][FO_MAX_LPC_ORDER]; double err[FO_MAX_LPC_ORDER];
						unsigned lo, o;
						/* the window table was built for the stream blocksize; a short last
						 * block uses its first n entries (stream_encoder.c:3305, :1324) */
						windowed_autocorrelation(x, c->windows + (size_t)a * c->stream_blocksize, n, max_order + 1, c->fscratch, autoc);
						if(autoc_out && a == 0) { unsigned l; for(l = 0; l <= max_order; l++) autoc_out[l] = autoc[l]; }
						if(autoc[0] == 0.0f) continue;
						max_order = levinson(autoc, max_order, coef, err); /* note: a reduced max_order persists for later windows (:3309 passes &max_lpc_order) */
						if(p->do_exhaustive_model_search) lo = 1;
						else {
							const unsigned g = guess_lpc_order(err, max_order, n, bps + (p->do_qlp_coeff_prec_search ? MIN_QLP_PRECISION : p->qlp_coeff_precision));
							lo = max_order = g;
						}
						if(max_order >= n) max_order = n - 1;
						for(o = lo; o <= max_order; o++) {
							const double eb = expected_bits(err[o-1], n - o);
							unsigned k, plo, phi, prec;
							if(eb >= (double)bps) continue;
							k = (eb > 0.0) ? (unsigned)(eb + 0.5) : 0;
							k++;
							if(k >= k_limit) k = k_limit - 1;
							if(p->do_qlp_coeff_prec_search) { /* :3341-3350 */
								plo = MIN_QLP_PRECISION;
								if(bps <= 17) { phi = umin(32 - bps - o, MAX_QLP_PRECISION); phi = umax(phi, plo); }
								else phi = MAX_QLP_PRECISION;
							}
							else plo = phi = p->qlp_coeff_precision;
							for(prec = plo; prec <= phi; prec++) {
								unsigned eff = prec; int shift, wide; uint32_t rbits;
								if(bps <= 16) eff = umin(eff, 32 - bps - ilog2u(o)); /* :3587-3591 */
								if(quantize_coefficients(coef[o-1], o, eff, cand.qlp, &shift) != 0) continue; /* :3593-3595 */
								wide = !(bps + eff + ilog2u(o) <= 32); /* :3597 */
								lpc_residual(x, n, cand.qlp, o, shift, wide, c->res_tmp);
								cand.type = FO_SUB_LPC; cand.order = o; cand.qlp_precision = eff; cand.qlp_shift = shift;
								cand.residual = c->res_tmp; cand.rice.param = c->param_tmp;
								rbits = best_rice_partitioning(c->res_tmp, n - o, o, k, k_limit, min_po, max_po, bps, c->sums, c->param_scratch, &cand.rice);
								cand.bits = 8u + wasted + 4u + 5u + o * (eff + bps) + rbits; /* :3636 */
								if(cand.bits > 0 && cand.bits < best->bits) plan_take(best, &cand, n - o, 1u << cand.rice.order);
							}
						}
					}
				}
			}
		}
	}
	if(best->bits == 0xffffffffu) { best->type = FO_SUB_VERBATIM; best->bits = 8u + wasted + n * bps; } /* :3393-3396 */
}

/* ------------------------------------------------------------------ bitstream writers */

/* stream_encoder_framing.c:219-364 */
static void write_frame_header(bitsink *s, unsigned blocksize, unsigned rate, unsigned channels, unsigned assignment, unsigned bps, uint32_t frame_number)
{
	const size_t start = (size_t)(s->nbits >> 3);
	unsigned code, bs_hint = 0, sr_hint = 0;
	put_bits(s, 0x3ffe, 14); put_bits(s, 0, 1); put_bits(s, 0, 1); /* sync, reserved, fixed-blocksize stream */
	switch(blocksize) {
	case 192: code = 1; break; case 576: code = 2; break; case 1152: code = 3; break; case 2304: code = 4; break; case 4608: code = 5; break;
	case 256: code = 8; break; case 512: code = 9; break; case 1024: code = 10; break; case 2048: code = 11; break; case 4096: code = 12; break;
	case 8192: code = 13; break; case 16384: code = 14; break; case 32768: code = 15; break;
	default: code = bs_hint = (blocksize <= 0x100) ? 6 : 7; break;
	}
	put_bits(s, code, 4);
	switch(rate) {
	case 88200: code = 1; break; case 176400: code = 2; break; case 192000: code = 3; break; case 8000: code = 4; break; case 16000: code = 5; break;
	case 22050: code = 6; break; case 24000: code = 7; break; case 32000: code = 8; break; case 44100: code = 9; break; case 48000: code = 10; break; case 96000: code = 11; break;
	default:
		if(rate <= 255000 && rate % 1000 == 0) code = sr_hint = 12;
		else if(rate % 10 == 0) code = sr_hint = 14;
		else if(rate <= 0xffff) code = sr_hint = 13;
		else code = 0;
		break;
	}
	put_bits(s, code, 4);
	put_bits(s, assignment == 0 ? channels - 1 : 7 + assignment, 4); /* 8 left/side, 9 right/side, 10 mid/side */
	switch(bps) { case 8: code = 1; break; case 12: code = 2; break; case 16: code = 4; break; case 20: code = 5; break; case 24: code = 6; break; default: code = 0; break; }
	put_bits(s, code, 3); put_bits(s, 0, 1);
	put_utf8(s, frame_number);
	if(bs_hint) put_bits(s, blocksize - 1, bs_hint == 6 ? 8 : 16);
	if(sr_hint == 12) put_bits(s, rate / 1000, 8); else if(sr_hint == 13) put_bits(s, rate, 16); else if(sr_hint == 14) put_bits(s, rate / 10, 16);
	put_bits(s, fo_crc8(s->p + start, (size_t)(s->nbits >> 3) - start), 8);
}

/* bitwriter.c:544-658 as a bit sequence: zig-zag fold, (u >> k) zeros, a one, k low bits */
static void write_rice_block(bitsink *s, const int32_t *r, unsigned n, unsigned k)
{
	unsigned i;
	for(i = 0; i < n; i++) {
		const uint32_t u = ((uint32_t)r[i] << 1) ^ (uint32_t)(r[i] >> 31);
		put_zeros(s, u >> k);
		put_bits(s, (u & ((1u << k) - 1u)) | (1u << k), k + 1);
	}
}

/* stream_encoder_framing.c:366-553 */
static void write_subframe(bitsink *s, const subframe_plan *sf, unsigned n)
{
	unsigned i; const unsigned w = sf->wasted_bits ? 1 : 0;
	switch(sf->type) {
	case FO_SUB_CONSTANT: put_bits(s, 0x00 | w, 8); break;
	case FO_SUB_VERBATIM: put_bits(s, 0x02 | w, 8); break;
	case FO_SUB_FIXED:    put_bits(s, 0x10 | (sf->order << 1) | w, 8); break;
	default:              put_bits(s, 0x40 | ((sf->order - 1) << 1) | w, 8); break;
	}
	if(w) put_unary(s, sf->wasted_bits - 1);
	if(sf->type == FO_SUB_CONSTANT) { put_sbits(s, sf->signal[0], sf->bps); return; }
	if(sf->type == FO_SUB_VERBATIM) { for(i = 0; i < n; i++) put_sbits(s, sf->signal[i], sf->bps); return; }
	for(i = 0; i < sf->order; i++) put_sbits(s, sf->signal[i], sf->bps);
	if(sf->type == FO_SUB_LPC) {
		put_bits(s, sf->qlp_precision - 1, 4); put_sbits(s, sf->qlp_shift, 5);
		for(i = 0; i < sf->order; i++) put_sbits(s, sf->qlp[i], sf->qlp_precision);
	}
	put_bits(s, sf->rice.rice2 ? 1 : 0, 2); put_bits(s, sf->rice.order, 4);
	{
		const unsigned plen = sf->rice.rice2 ? 5 : 4, parts = 1u << sf->rice.order; unsigned done = 0;
		for(i = 0; i < parts; i++) {
			unsigned cnt = n >> sf->rice.order;
			if(sf->rice.order == 0) cnt = n - sf->order; else if(i == 0) cnt -= sf->order;
			put_bits(s, sf->rice.param[i], plen);
			write_rice_block(s, sf->residual + done, cnt, sf->rice.param[i]);
			done += cnt;
		}
	}
}

/* ------------------------------------------------------------------ one frame */

typedef struct { unsigned loose_count, loose_period, last_assignment; } ms_state;

/* stream_encoder.c:2866-2923 process_frame_ + :2925-3167 process_subframes_.
 * chan[c] points at n samples of channel c (this routine works on private copies). */
static size_t encode_frame(coder_ctx *c, const int32_t *const chan[], unsigned n, int fractional, uint32_t frame_number,
	ms_state *ms, subframe_plan plans[4], int32_t *sig[4], uint8_t *out, size_t cap, fo_frame_info *info)
{
	const fo_params *p = c->p;
	unsigned max_po, min_po, ch, i, assignment = 0, do_indep = 1, do_ms = 0, nsub;
	const subframe_plan *emit[FO_MAX_CHANNELS];
	bitsink s;

	if(fractional) max_po = 0; /* :2934-2941 */
	else {
		unsigned b = n; max_po = 0;
		while(!(b & 1)) { max_po++; b >>= 1; }
		max_po = umin(umin(max_po, MAX_RICE_PARTITION_ORDER), p->max_residual_partition_order);
	}
	min_po = umin(p->min_residual_partition_order, max_po);

	if(p->do_mid_side) { /* :2957-2976 */
		if(p->loose_mid_side && ms->loose_count != 0) { do_indep = (ms->last_assignment == 0); do_ms = !do_indep; }
		else { do_indep = 1; do_ms = 1; }
	}

	if(info) { memset(info, 0, sizeof *info); info->blocksize = n; info->frame_number = frame_number; }

	if(p->channels == 2 && p->do_mid_side) {
		/* plans 0,1 = left,right; 2,3 = mid,side (:1992-1999: mid = (l+r)>>1 arithmetic, side = l-r) */
		for(i = 0; i < n; i++) { const int32_t l = chan[0][i], r = chan[1][i]; sig[0][i] = l; sig[1][i] = r; sig[2][i] = (int32_t)((uint32_t)l + (uint32_t)r) >> 1; sig[3][i] = (int32_t)((uint32_t)l - (uint32_t)r); }
		for(ch = 0; ch < 4; ch++) {
			const int is_ms = ch >= 2;
			if((is_ms && !do_ms) || (!is_ms && !do_indep)) continue;
			{
				const unsigned w = strip_wasted_bits(sig[ch], n); /* :2983-2997 */
				const unsigned bps = p->bits_per_sample - w + (ch == 3 ? 1 : 0);
				choose_subframe(c, sig[ch], n, bps, w, min_po, max_po, &plans[ch], info ? info->autoc[ch] : 0, info ? info->fixed_err[ch] : 0);
			}
		}
		if(p->loose_mid_side && ms->loose_count > 0) assignment = (ms->last_assignment == 0) ? 0 : 3; /* :3059-3061 */
		else { /* :3063-3086 */
			const uint32_t b[4] = { plans[0].bits + plans[1].bits, plans[0].bits + plans[3].bits, plans[1].bits + plans[3].bits, plans[2].bits + plans[3].bits };
			uint32_t m = b[0]; unsigned a;
			for(a = 1; a < 4; a++) if(b[a] < m) { m = b[a]; assignment = a; }
		}
		switch(assignment) { /* :3096-3115 */
		case 0: emit[0] = &plans[0]; emit[1] = &plans[1]; break;
		case 1: emit[0] = &plans[0]; emit[1] = &plans[3]; break;
		case 2: emit[0] = &plans[3]; emit[1] = &plans[1]; break;
		default: emit[0] = &plans[2]; emit[1] = &plans[3]; break;
		}
		nsub = 2;
	}
	else {
		/* independent channels only; plans are reused round-robin since each is emitted before the next is built */
		nsub = p->channels;
		for(ch = 0; ch < nsub; ch++) emit[ch] = 0;
	}

	s.p = out; s.cap = cap; s.nbits = 0; s.overflow = 0;
	write_frame_header(&s, n, p->sample_rate, p->channels, assignment, p->bits_per_sample, frame_number);
	if(p->channels == 2 && p->do_mid_side) {
		for(ch = 0; ch < 2; ch++) write_subframe(&s, emit[ch], n);
	}
	else {
		for(ch = 0; ch < nsub; ch++) {
			unsigned w;
			memcpy(sig[0], chan[ch], sizeof(int32_t) * n);
			w = strip_wasted_bits(sig[0], n);
			choose_subframe(c, sig[0], n, p->bits_per_sample - w, w, min_po, max_po, &plans[0], (info && ch < 4) ? info->autoc[ch] : 0, (info && ch < 4) ? info->fixed_err[ch] : 0);
			write_subframe(&s, &plans[0], n);
			if(info) {
				fo_subframe_info *si = &info->sub[ch]; unsigned k;
				si->type = plans[0].type; si->order = plans[0].order; si->wasted_bits = plans[0].wasted_bits; si->bps = plans[0].bps;
				si->qlp_precision = plans[0].qlp_precision; si->qlp_shift = plans[0].qlp_shift; memcpy(si->qlp, plans[0].qlp, sizeof si->qlp);
				si->partition_order = plans[0].rice.order; si->rice2 = plans[0].rice.rice2; si->estimate_bits = plans[0].bits;
				if(plans[0].type >= FO_SUB_FIXED) for(k = 0; k < (1u << plans[0].rice.order) && k < 64; k++) si->rice_param[k] = plans[0].rice.param[k];
			}
		}
	}
	if(info && p->channels == 2 && p->do_mid_side) {
		for(ch = 0; ch < 2; ch++) {
			fo_subframe_info *si = &info->sub[ch]; const subframe_plan *pl = emit[ch]; unsigned k;
			si->type = pl->type; si->order = pl->order; si->wasted_bits = pl->wasted_bits; si->bps = pl->bps;
			si->qlp_precision = pl->qlp_precision; si->qlp_shift = pl->qlp_shift; memcpy(si->qlp, pl->qlp, sizeof si->qlp);
			si->partition_order = pl->rice.order; si->rice2 = pl->rice.rice2; si->estimate_bits = pl->bits;
			if(pl->type >= FO_SUB_FIXED) for(k = 0; k < (1u << pl->rice.order) && k < 64; k++) si->rice_param[k] = pl->rice.param[k];
		}
	}
	/* :2890-2905 pad to a byte, CRC-16 of everything so far */
	if(s.nbits & 7) put_bits(&s, 0, 8 - (unsigned)(s.nbits & 7));
	if(!s.overflow) put_bits(&s, fo_crc16(out, (size_t)(s.nbits >> 3)), 16);
	if(s.overflow) return 0;

	if(p->loose_mid_side) { if(++ms->loose_count >= ms->loose_period) ms->loose_count = 0; } /* :3158-3162 */
	ms->last_assignment = assignment;
	if(info) { info->channel_assignment = assignment; info->bytes = (uint32_t)(s.nbits >> 3); }
	return (size_t)(s.nbits >> 3);
}

long fo_encode_stream(const fo_params *pin, const int32_t *pcm, uint64_t nsamples,
	uint8_t *out, size_t cap, size_t *outlen,
	uint32_t *frame_bytes, uint32_t *frame_samples, uint32_t max_frames,
	fo_frame_info *infos, uint8_t md5[16])
{
	return fo_encode_stream_ex(pin, pcm, nsamples, 0, 1, out, cap, outlen, frame_bytes, frame_samples, max_frames, infos, md5);
}

long fo_encode_stream_ex(const fo_params *pin, const int32_t *pcm, uint64_t nsamples, uint32_t first_frame_number, int final_block,
	uint8_t *out, size_t cap, size_t *outlen,
	uint32_t *frame_bytes, uint32_t *frame_samples, uint32_t max_frames,
	fo_frame_info *infos, uint8_t md5[16])
{
	fo_params p = *pin; coder_ctx c; ms_state ms; subframe_plan plans[4];
	int32_t *sig[4], *chanbuf[FO_MAX_CHANNELS]; const int32_t *chan[FO_MAX_CHANNELS];
	float *windows; unsigned N, ch, i; uint64_t pos = 0; long nframes = 0; size_t used = 0; long rc = 0;
	unsigned maxparams;

	if(fo_params_resolve(&p) != 0) return -1;
	N = p.blocksize;
	maxparams = 1u << p.max_residual_partition_order;
	memset(&c, 0, sizeof c); memset(plans, 0, sizeof plans);
	c.p = &p; c.stream_blocksize = N;
	c.wide_fixed = (p.bits_per_sample + ilog2u(N) + 1 > 30);
	windows = (float*)malloc(sizeof(float) * (size_t)N * p.num_apodizations);
	for(i = 0; i < p.num_apodizations; i++) fo_window(p.apod_type[i], p.apod_param[i], (int32_t)N, windows + (size_t)i * N);
	c.windows = windows;
	c.fscratch = (float*)malloc(sizeof(float) * N);
	c.res_tmp = (int32_t*)malloc(sizeof(int32_t) * N);
	c.sums = (uint64_t*)malloc(sizeof(uint64_t) * 2 * maxparams);
	c.param_scratch = (uint32_t*)malloc(sizeof(uint32_t) * 2 * maxparams);
	c.param_tmp = (uint32_t*)malloc(sizeof(uint32_t) * maxparams);
	for(i = 0; i < 4; i++) { sig[i] = (int32_t*)malloc(sizeof(int32_t) * N); plans[i].residual = (int32_t*)malloc(sizeof(int32_t) * N); plans[i].rice.param = (uint32_t*)malloc(sizeof(uint32_t) * maxparams); }
	for(ch = 0; ch < p.channels; ch++) { chanbuf[ch] = (int32_t*)malloc(sizeof(int32_t) * N); chan[ch] = chanbuf[ch]; }
	/* stream_encoder.c:878-890 */
	ms.loose_period = (unsigned)((double)p.sample_rate * 0.4 / (double)N + 0.5); if(ms.loose_period == 0) ms.loose_period = 1;
	ms.loose_count = first_frame_number % ms.loose_period; ms.last_assignment = 0;

	/* Blocking as FLAC__stream_encoder_process_interleaved + _finish (stream_encoder.c:2025-2105, :1322-1327):
	 * a block is coded once blocksize+1 samples are known, so every full block except a
	 * trailing exactly-full one goes through the non-final path; whatever remains (1..N
	 * samples) is coded by finish() with fractional = (remaining != N). */
	while(pos < nsamples) {
		const uint64_t left = nsamples - pos; const unsigned n = left > N ? N : (unsigned)left;
		const int fractional = (n != N) && final_block;
		size_t fb;
		if((uint32_t)nframes >= max_frames && frame_bytes) { rc = -3; break; }
		for(i = 0; i < n; i++) for(ch = 0; ch < p.channels; ch++) chanbuf[ch][i] = pcm[(pos + i) * p.channels + ch];
		fb = encode_frame(&c, chan, n, fractional, first_frame_number + (uint32_t)nframes, &ms, plans, sig, out + used, cap - used, infos ? &infos[nframes] : 0);
		if(fb == 0) { rc = -2; break; }
		if(frame_bytes) frame_bytes[nframes] = (uint32_t)fb;
		if(frame_samples) frame_samples[nframes] = n;
		used += fb; pos += n; nframes++;
	}
	if(md5) fo_md5_pcm(pcm, nsamples, p.channels, p.bits_per_sample, md5);
	if(outlen) *outlen = used;

	for(i = 0; i < 4; i++) { free(sig[i]); free(plans[i].residual); free(plans[i].rice.param); }
	for(ch = 0; ch < p.channels; ch++) free(chanbuf[ch]);
	free(windows); free(c.fscratch); free(c.res_tmp); free(c.sums); free(c.param_scratch); free(c.param_tmp);
	return rc < 0 ? rc : nframes;
}

/* ------------------------------------------------------------------ stream head */

void fo_utf8(uint32_t v, uint8_t out[8], uint32_t *len)
{
	bitsink s; s.p = out; s.cap = 8; s.nbits = 0; s.overflow = 0;
	put_utf8(&s, v);
	*len = (uint32_t)(s.nbits >> 3);
}

void fo_write_stream_head(uint8_t out[42], uint32_t min_blocksize, uint32_t max_blocksize,
	uint32_t min_framesize, uint32_t max_framesize, uint32_t sample_rate, uint32_t channels,
	uint32_t bps, uint64_t total_samples, const uint8_t md5[16], int is_last)
{
	bitsink s; s.p = out; s.cap = 42; s.nbits = 0; s.overflow = 0;
	put_bits(&s, 0x664C6143u, 32);               /* "fLaC" */
	put_bits(&s, is_last ? 1 : 0, 1); put_bits(&s, 0, 7); put_bits(&s, 34, 24);
	put_bits(&s, min_blocksize, 16); put_bits(&s, max_blocksize, 16);
	put_bits(&s, min_framesize, 24); put_bits(&s, max_framesize, 24);
	put_bits(&s, sample_rate, 20); put_bits(&s, channels - 1, 3); put_bits(&s, bps - 1, 5);
	put_bits(&s, (uint32_t)(total_samples >> 32) & 0xf, 4); put_bits(&s, (uint32_t)total_samples, 32);
	memcpy(out + 26, md5, 16);
}

/* ------------------------------------------------------------------ decoder */

typedef struct { const uint8_t *p; size_t len; uint64_t pos; int eof; } bitsrc;

static uint32_t get_bits(bitsrc *s, unsigned n)
{
	uint32_t v = 0;
	while(n) {
		const size_t byte = (size_t)(s->pos >> 3); const unsigned avail = 8 - (unsigned)(s->pos & 7), take = n < avail ? n : avail;
		if(byte >= s->len) { s->eof = 1; return 0; }
		v <<= take;
		v |= (s->p[byte] >> (avail - take)) & ((1u << take) - 1u);
		s->pos += take; n -= take;
	}
	return v;
}
static int32_t get_sbits(bitsrc *s, unsigned n)
{
	uint32_t v;
	if(n == 0) return 0;
	v = get_bits(s, n);
	if(n < 32 && (v & (1u << (n - 1)))) v |= ~((1u << n) - 1u);
	return (int32_t)v;
}
static uint32_t get_unary(bitsrc *s) { uint32_t z = 0; while(!s->eof && get_bits(s, 1) == 0) z++; return z; }

/* stream_decoder.c:2721-2777 */
static int read_residual(bitsrc *s, unsigned n, unsigned order, unsigned rice2, int32_t *res)
{
	const unsigned po = get_bits(s, 4), plen = rice2 ? 5 : 4, esc = rice2 ? 31 : 15, parts = 1u << po;
	unsigned part, done = 0;
	if(po == 0) { if(n < order) return -1; } else if((n >> po) < order) return -1;
	for(part = 0; part < parts; part++) {
		unsigned cnt = (po == 0) ? n - order : (n >> po) - (part == 0 ? order : 0), k = get_bits(s, plen), i;
		if(k < esc) {
			for(i = 0; i < cnt; i++) {
				const uint32_t msb = get_unary(s), lsb = k ? get_bits(s, k) : 0, u = (msb << k) | lsb;
				res[done + i] = (int32_t)(u >> 1) ^ -(int32_t)(u & 1); /* bitreader.c rice unfold */
				if(s->eof) return -1;
			}
		}
		else {
			const unsigned raw = get_bits(s, 5);
			for(i = 0; i < cnt; i++) res[done + i] = get_sbits(s, raw);
		}
		done += cnt;
	}
	return s->eof ? -1 : 0;
}

/* stream_decoder.c:2456-2524 + restore (fixed.c:395-435, lpc.c:795-1321) */
static int read_subframe(bitsrc *s, unsigned n, unsigned bps, int32_t *out, int32_t *res)
{
	uint32_t head = get_bits(s, 8); unsigned wasted = 0, i, j;
	if(head & 0x80) return -1;
	if(head & 1) { wasted = get_unary(s) + 1; if(wasted >= bps) return -1; bps -= wasted; }
	head &= 0xfe;
	if(head == 0) { const int32_t v = get_sbits(s, bps); for(i = 0; i < n; i++) out[i] = v; }
	else if(head == 2) { for(i = 0; i < n; i++) out[i] = get_sbits(s, bps); }
	else if(head < 16) return -5;
	else if(head <= 24) {
		const unsigned order = (head >> 1) & 7; unsigned method;
		if(order > n) return -1;
		for(i = 0; i < order; i++) out[i] = get_sbits(s, bps);
		method = get_bits(s, 2); if(method > 1) return -5;
		if(read_residual(s, n, order, method, res) != 0) return -1;
		for(i = order; i < n; i++) {
			const uint32_t r = (uint32_t)res[i - order]; uint32_t v;
			switch(order) {
			case 0: v = r; break;
			case 1: v = r + (uint32_t)out[i-1]; break;
			case 2: v = r + 2u*(uint32_t)out[i-1] - (uint32_t)out[i-2]; break;
			case 3: v = r + 3u*(uint32_t)out[i-1] - 3u*(uint32_t)out[i-2] + (uint32_t)out[i-3]; break;
			default: v = r + 4u*(uint32_t)out[i-1] - 6u*(uint32_t)out[i-2] + 4u*(uint32_t)out[i-3] - (uint32_t)out[i-4]; break;
			}
			out[i] = (int32_t)v;
		}
	}
	else if(head < 64) return -5;
	else {
		const unsigned order = ((head >> 1) & 31) + 1; unsigned prec, method; int shift; int32_t q[FO_MAX_LPC_ORDER]; int wide;
		if(order > n) return -1;
		for(i = 0; i < order; i++) out[i] = get_sbits(s, bps);
		prec = get_bits(s, 4); if(prec == 15) return -1; prec++;
		shift = get_sbits(s, 5);
		for(i = 0; i < order; i++) q[i] = get_sbits(s, prec);
		method = get_bits(s, 2); if(method > 1) return -5;
		if(read_residual(s, n, order, method, res) != 0) return -1;
		if(shift < 0) return -5; /* the reference encoder never emits a negative shift (lpc.c:225-260) */
		wide = !(bps + prec + ilog2u(order) <= 32); /* stream_decoder.c:2682 */
		for(i = order; i < n; i++) {
			if(wide) { int64_t sum = 0; for(j = 0; j < order; j++) sum += (int64_t)q[j] * (int64_t)out[i-1-j]; out[i] = (int32_t)((uint32_t)res[i-order] + (uint32_t)(int32_t)(sum >> shift)); }
			else { uint32_t sum = 0; for(j = 0; j < order; j++) sum += (uint32_t)q[j] * (uint32_t)out[i-1-j]; out[i] = (int32_t)((uint32_t)res[i-order] + (uint32_t)((int32_t)sum >> shift)); }
		}
	}
	if(wasted) for(i = 0; i < n; i++) out[i] = (int32_t)((uint32_t)out[i] << wasted);
	return s->eof ? -1 : 0;
}

long fo_decode_frames(const uint8_t *in, size_t len, uint32_t channels, uint32_t bps,
	int32_t *pcm, uint64_t cap_samples, uint64_t *nsamples_out, uint32_t *frame_offsets, uint32_t max_frames)
{
	size_t off = 0; long nframes = 0; uint64_t total = 0; int32_t *buf[FO_MAX_CHANNELS], *res; unsigned ch; long rc = 0;
	const unsigned maxn = 65536;
	for(ch = 0; ch < channels; ch++) buf[ch] = (int32_t*)malloc(sizeof(int32_t) * maxn);
	res = (int32_t*)malloc(sizeof(int32_t) * maxn);
	while(off < len) {
		bitsrc s; unsigned bscode, srcode, chcode, bpscode, n, fbps, assignment, nch, i; uint32_t first; size_t hdr_end;
		s.p = in + off; s.len = len - off; s.pos = 0; s.eof = 0;
		/* stream_decoder.c:2147-2454 header parse */
		if(get_bits(&s, 14) != 0x3ffe) { rc = -1; break; }
		if(get_bits(&s, 1) != 0) { rc = -5; break; }
		(void)get_bits(&s, 1);
		bscode = get_bits(&s, 4); srcode = get_bits(&s, 4); chcode = get_bits(&s, 4); bpscode = get_bits(&s, 3);
		if(get_bits(&s, 1) != 0) { rc = -5; break; }
		first = get_bits(&s, 8); /* UTF-8 coded number: skip continuation bytes */
		{ unsigned extra = 0; if(first & 0x80) { uint32_t m = 0x40; while(first & m) { extra++; m >>= 1; } if(extra == 0 || extra > 6) { rc = -1; break; } } for(i = 0; i < extra; i++) (void)get_bits(&s, 8); }
		switch(bscode) { case 0: n = 0; break; case 1: n = 192; break; case 2: case 3: case 4: case 5: n = 576u << (bscode - 2); break; case 6: n = get_bits(&s, 8) + 1; break; case 7: n = get_bits(&s, 16) + 1; break; default: n = 256u << (bscode - 8); break; }
		if(n == 0) { rc = -5; break; }
		if(srcode == 12) (void)get_bits(&s, 8); else if(srcode == 13 || srcode == 14) (void)get_bits(&s, 16); else if(srcode == 15) { rc = -1; break; }
		hdr_end = (size_t)(s.pos >> 3);
		{ const uint8_t want = (uint8_t)get_bits(&s, 8); if(s.eof) { rc = -1; break; } if(fo_crc8(in + off, hdr_end) != want) { rc = -2; break; } }
		switch(bpscode) { case 0: fbps = bps; break; case 1: fbps = 8; break; case 2: fbps = 12; break; case 4: fbps = 16; break; case 5: fbps = 20; break; case 6: fbps = 24; break; default: fbps = 0; break; }
		if(fbps == 0) { rc = -5; break; }
		if(chcode < 8) { nch = chcode + 1; assignment = 0; } else if(chcode <= 10) { nch = 2; assignment = chcode - 7; } else { rc = -5; break; }
		if(nch != channels) { rc = -5; break; }
		for(ch = 0; ch < nch; ch++) {
			unsigned sb = fbps; int r;
			if((assignment == 1 && ch == 1) || (assignment == 2 && ch == 0) || (assignment == 3 && ch == 1)) sb++; /* stream_decoder.c:2037-2058 side channel has one more bit */
			r = read_subframe(&s, n, sb, buf[ch], res);
			if(r != 0) { rc = r; break; }
		}
		if(rc != 0) break;
		if(s.pos & 7) { if(get_bits(&s, 8 - (unsigned)(s.pos & 7)) != 0) { rc = -1; break; } }
		{ const size_t body = (size_t)(s.pos >> 3); const uint32_t want = get_bits(&s, 16); if(s.eof) { rc = -1; break; } if(fo_crc16(in + off, body) != want) { rc = -3; break; } }
		/* stream_decoder.c:2073-2108 undo inter-channel decorrelation */
		if(assignment == 1) for(i = 0; i < n; i++) buf[1][i] = (int32_t)((uint32_t)buf[0][i] - (uint32_t)buf[1][i]);
		else if(assignment == 2) for(i = 0; i < n; i++) buf[0][i] = (int32_t)((uint32_t)buf[0][i] + (uint32_t)buf[1][i]);
		else if(assignment == 3) for(i = 0; i < n; i++) {
			const int32_t side = buf[1][i]; int32_t mid = (int32_t)((uint32_t)buf[0][i] << 1) | (side & 1);
			buf[0][i] = (int32_t)((uint32_t)mid + (uint32_t)side) >> 1; buf[1][i] = (int32_t)((uint32_t)mid - (uint32_t)side) >> 1;
		}
		if(total + n > cap_samples) { rc = -4; break; }
		for(i = 0; i < n; i++) for(ch = 0; ch < nch; ch++) pcm[(total + i) * nch + ch] = buf[ch][i];
		if(frame_offsets && (uint32_t)nframes < max_frames) frame_offsets[nframes] = (uint32_t)off;
		total += n; off += (size_t)(s.pos >> 3); nframes++;
	}
	for(ch = 0; ch < channels; ch++) free(buf[ch]);
	free(res);
	if(nsamples_out) *nsamples_out = total;
	return rc < 0 ? rc : nframes;
}
