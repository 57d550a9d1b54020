/* MD5 of PCM as the reference computes it for STREAMINFO (src/libFLAC/md5.c:216-420): the
 * standard digest over the samples laid out little-endian, (bps + 7) / 8 bytes each, channels
 * interleaved.  Host code shared by the stream encoder and decoder objects. */
#ifndef FLAC_AMD_HOST_MD5_H
#define FLAC_AMD_HOST_MD5_H
#include <stdint.h>
#include <stddef.h>
#include <string.h>

typedef struct { uint32_t h[4]; uint64_t nbytes; uint8_t buf[64]; } md5_state;
static const uint32_t md5_k[64] = {
	0xd76aa478,0xe8c7b756,0x242070db,0xc1bdceee,0xf57c0faf,0x4787c62a,0xa8304613,0xfd469501,0x698098d8,0x8b44f7af,0xffff5bb1,0x895cd7be,0x6b901122,0xfd987193,0xa679438e,0x49b40821,
	0xf61e2562,0xc040b340,0x265e5a51,0xe9b6c7aa,0xd62f105d,0x02441453,0xd8a1e681,0xe7d3fbc8,0x21e1cde6,0xc33707d6,0xf4d50d87,0x455a14ed,0xa9e3e905,0xfcefa3f8,0x676f02d9,0x8d2a4c8a,
	0xfffa3942,0x8771f681,0x6d9d6122,0xfde5380c,0xa4beea44,0x4bdecfa9,0xf6bb4b60,0xbebfbc70,0x289b7ec6,0xeaa127fa,0xd4ef3085,0x04881d05,0xd9d4d039,0xe6db99e5,0x1fa27cf8,0xc4ac5665,
	0xf4292244,0x432aff97,0xab9423a7,0xfc93a039,0x655b59c3,0x8f0ccc92,0xffeff47d,0x85845dd1,0x6fa87e4f,0xfe2ce6e0,0xa3014314,0x4e0811a1,0xf7537e82,0xbd3af235,0x2ad7d2bb,0xeb86d391 };
/* One 64-byte block (RFC 1321): 64 steps in four rounds of 16; step i mixes word md5_word(i) of the block with
 * md5_k[i] into the running state through the round's boolean function and a left rotation.  The state rotates
 * through a, b, c, d by renaming, which leaves the chain of 64 dependent steps that bounds a single stream. */
static inline uint32_t md5_rotl(uint32_t v, unsigned s) { return v << s | v >> (32u - s); }
static inline uint32_t md5_mix(unsigned round, uint32_t b, uint32_t c, uint32_t d)
{
	switch(round) {
	case 0: return d ^ (b & (c ^ d));      /* b ? c : d */
	case 1: return c ^ (d & (b ^ c));      /* d ? b : c */
	case 2: return b ^ c ^ d;              /* parity */
	default: return c ^ (b | ~d);
	}
}
static void md5_block(md5_state *m, const uint8_t *p)
{
	static const uint8_t rot[4][4] = { { 7, 12, 17, 22 }, { 5, 9, 14, 20 }, { 4, 11, 16, 23 }, { 6, 10, 15, 21 } };
	static const uint8_t first[4] = { 0, 1, 5, 0 }, stride[4] = { 1, 5, 3, 7 };   /* word of step j of round r: (first + j * stride) mod 16 */
	uint32_t w[16], a = m->h[0], b = m->h[1], c = m->h[2], d = m->h[3];
#if defined(__BYTE_ORDER__) && __BYTE_ORDER__ == __ORDER_LITTLE_ENDIAN__
	memcpy(w, p, 64);
#else
	{ unsigned j; for(j = 0; j < 16; j++) w[j] = (uint32_t)p[4*j] | (uint32_t)p[4*j+1] << 8 | (uint32_t)p[4*j+2] << 16 | (uint32_t)p[4*j+3] << 24; }
#endif
	/* 4 x 16 steps written out through the preprocessor (no reliance on an unrolling pragma: this header is compiled by
	 * gcc and by clang), every table index a constant */
#define MD5_STEP_AT(r, j) { \
		const uint32_t t = a + md5_mix((r), b, c, d) + w[(first[(r)] + (j) * stride[(r)]) & 15u] + md5_k[16u * (r) + (j)]; \
		a = d; d = c; c = b; \
		b += md5_rotl(t, rot[(r)][(j) & 3u]); }
#define MD5_STEPS_4(r, j) MD5_STEP_AT(r, j) MD5_STEP_AT(r, (j) + 1) MD5_STEP_AT(r, (j) + 2) MD5_STEP_AT(r, (j) + 3)
#define MD5_ROUND_16(r) MD5_STEPS_4(r, 0) MD5_STEPS_4(r, 4) MD5_STEPS_4(r, 8) MD5_STEPS_4(r, 12)
	MD5_ROUND_16(0) MD5_ROUND_16(1) MD5_ROUND_16(2) MD5_ROUND_16(3)
#undef MD5_ROUND_16
#undef MD5_STEPS_4
#undef MD5_STEP_AT
	m->h[0] += a; m->h[1] += b; m->h[2] += c; m->h[3] += d;
}
static void md5_init(md5_state *m) { m->h[0] = 0x67452301; m->h[1] = 0xefcdab89; m->h[2] = 0x98badcfe; m->h[3] = 0x10325476; m->nbytes = 0; }
static void md5_update(md5_state *m, const uint8_t *p, size_t n)
{
	size_t fill = (size_t)(m->nbytes & 63);
	m->nbytes += n;
	if(fill) { size_t t = 64 - fill; if(t > n) t = n; memcpy(m->buf + fill, p, t); p += t; n -= t; if(fill + t < 64) return; md5_block(m, m->buf); }
	while(n >= 64) { md5_block(m, p); p += 64; n -= 64; }
	if(n) memcpy(m->buf, p, n);
}
static void md5_final(md5_state *m, uint8_t out[16])
{
	const uint64_t bits = m->nbytes * 8; uint8_t pad[72]; const size_t fill = (size_t)(m->nbytes & 63), n = (fill < 56 ? 56 - fill : 120 - fill); int i;
	memset(pad, 0, sizeof pad); pad[0] = 0x80;
	for(i = 0; i < 8; i++) pad[n + i] = (uint8_t)(bits >> (8*i));
	md5_update(m, pad, n + 8);
	for(i = 0; i < 16; i++) out[i] = (uint8_t)(m->h[i >> 2] >> (8 * (i & 3)));
}


/* feed interleaved wide samples; scratch must hold at least 4096 bytes */
static void md5_feed_pcm(md5_state *m, const int32_t *interleaved, size_t nvalues, unsigned bits_per_sample, uint8_t *scratch, size_t scratch_bytes)
{
	const unsigned nb = (bits_per_sample + 7) / 8; size_t i, fill = 0; unsigned b;
	for(i = 0; i < nvalues; i++) {
		const uint32_t v = (uint32_t)interleaved[i];
		for(b = 0; b < nb; b++) scratch[fill++] = (uint8_t)(v >> (8 * b));
		if(fill + 4 > scratch_bytes) { md5_update(m, scratch, fill); fill = 0; }
	}
	if(fill) md5_update(m, scratch, fill);
}

/* the same for one frame held channel by channel (chan[c][i]): the samples go into the checksum interleaved (md5.c:394-420) */
static void md5_feed_planar(md5_state *m, const int32_t *const chan[], unsigned channels, size_t nsamples, unsigned bits_per_sample, uint8_t *scratch, size_t scratch_bytes)
{
	const unsigned nb = (bits_per_sample + 7) / 8; size_t i, fill = 0; unsigned b, c;
	if(channels == 2 && nb == 2) { /* the common case, two bytes of each channel in turn */
		const int32_t *l = chan[0], *r = chan[1];
		for(i = 0; i < nsamples; i++) {
			const uint32_t v = ((uint32_t)l[i] & 0xffffu) | ((uint32_t)r[i] << 16);
			scratch[fill] = (uint8_t)v; scratch[fill + 1] = (uint8_t)(v >> 8); scratch[fill + 2] = (uint8_t)(v >> 16); scratch[fill + 3] = (uint8_t)(v >> 24);
			fill += 4;
			if(fill + 4 > scratch_bytes) { md5_update(m, scratch, fill); fill = 0; }
		}
	}
	else for(i = 0; i < nsamples; i++) for(c = 0; c < channels; c++) {
		const uint32_t v = (uint32_t)chan[c][i];
		for(b = 0; b < nb; b++) scratch[fill++] = (uint8_t)(v >> (8 * b));
		if(fill + 4 > scratch_bytes) { md5_update(m, scratch, fill); fill = 0; }
	}
	if(fill) md5_update(m, scratch, fill);
}

/* the same for samples held as int16 (streams of at most 16 bits): two-byte samples are already the byte stream */
static void md5_feed_pcm16(md5_state *m, const int16_t *interleaved, size_t nvalues, unsigned bits_per_sample, uint8_t *scratch, size_t scratch_bytes)
{
	size_t i, fill = 0;
	if((bits_per_sample + 7) / 8 == 2) {
#if defined(__BYTE_ORDER__) && __BYTE_ORDER__ == __ORDER_LITTLE_ENDIAN__
		md5_update(m, (const uint8_t*)interleaved, nvalues * 2);
		return;
#else
		for(i = 0; i < nvalues; i++) { scratch[fill++] = (uint8_t)interleaved[i]; scratch[fill++] = (uint8_t)((uint16_t)interleaved[i] >> 8); if(fill + 4 > scratch_bytes) { md5_update(m, scratch, fill); fill = 0; } }
#endif
	}
	else for(i = 0; i < nvalues; i++) { scratch[fill++] = (uint8_t)interleaved[i]; if(fill + 4 > scratch_bytes) { md5_update(m, scratch, fill); fill = 0; } }
	if(fill) md5_update(m, scratch, fill);
}
#endif
