// MD5 of the unencoded PCM, one lane per independent stream.
//
// Replaces FLAC__MD5Accumulate / format_input_ / FLAC__MD5Final (md5.c:216-420): the
// channels are interleaved, each sample is written little-endian in (bps+7)/8 bytes, and
// the standard MD5 is run over that byte stream.  MD5 is a strictly serial chain over a
// stream (each 64-byte block depends on the previous state), so a stream maps to ONE lane
// and the batch supplies the parallelism: this entry point is for batches of many clips.
// A single long stream is hashed on a host thread by the stream encoder shim instead.
#include "engine_types.h"
#include <hip/hip_runtime.h>
#include <mutex>
#include <string.h>

__constant__ uint32_t md5_k[64] = {
	0xd76aa478,0xe8c7b756,0x242070db,0xc1bdceee,0xf57c0faf,0x4787c62a,0xa8304613,0xfd469501,
	0x698098d8,0x8b44f7af,0xffff5bb1,0x895cd7be,0x6b901122,0xfd987193,0xa679438e,0x49b40821,
	0xf61e2562,0xc040b340,0x265e5a51,0xe9b6c7aa,0xd62f105d,0x02441453,0xd8a1e681,0xe7d3fbc8,
	0x21e1cde6,0xc33707d6,0xf4d50d87,0x455a14ed,0xa9e3e905,0xfcefa3f8,0x676f02d9,0x8d2a4c8a,
	0xfffa3942,0x8771f681,0x6d9d6122,0xfde5380c,0xa4beea44,0x4bdecfa9,0xf6bb4b60,0xbebfbc70,
	0x289b7ec6,0xeaa127fa,0xd4ef3085,0x04881d05,0xd9d4d039,0xe6db99e5,0x1fa27cf8,0xc4ac5665,
	0xf4292244,0x432aff97,0xab9423a7,0xfc93a039,0x655b59c3,0x8f0ccc92,0xffeff47d,0x85845dd1,
	0x6fa87e4f,0xfe2ce6e0,0xa3014314,0x4e0811a1,0xf7537e82,0xbd3af235,0x2ad7d2bb,0xeb86d391 };

__device__ __forceinline__ uint32_t rotl(uint32_t v, uint32_t s) { return (v << s) | (v >> (32u - s)); }

__device__ void md5_block(uint32_t (&h)[4], const uint32_t (&m)[16])
{
	uint32_t a = h[0], b = h[1], c = h[2], d = h[3];
#define STEP(F, G, S, I) { const uint32_t f_ = (F) + a + md5_k[I] + m[G]; a = d; d = c; c = b; b = b + rotl(f_, S); }
#pragma unroll
	for(int i = 0; i < 16; i++) { const uint32_t s[4] = {7, 12, 17, 22}; STEP((b & c) | (~b & d), i, s[i & 3], i) }
#pragma unroll
	for(int i = 16; i < 32; i++) { const uint32_t s[4] = {5, 9, 14, 20}; STEP((d & b) | (~d & c), (5 * i + 1) & 15, s[i & 3], i) }
#pragma unroll
	for(int i = 32; i < 48; i++) { const uint32_t s[4] = {4, 11, 16, 23}; STEP(b ^ c ^ d, (3 * i + 5) & 15, s[i & 3], i) }
#pragma unroll
	for(int i = 48; i < 64; i++) { const uint32_t s[4] = {6, 10, 15, 21}; STEP(c ^ (b | ~d), (7 * i) & 15, s[i & 3], i) }
#undef STEP
	h[0] += a; h[1] += b; h[2] += c; h[3] += d;
}

// value t of the flattened interleaved stream
__device__ __forceinline__ uint32_t md5_val(const void *pcm, uint32_t fmt, uint64_t t)
{
	return fmt == FLACAMD_PCM_S16 ? (uint32_t)(int32_t)((const int16_t*)pcm)[t] : (uint32_t)((const int32_t*)pcm)[t];
}

// 64 consecutive values = NB whole 64-byte blocks, packed with compile-time byte positions
template<int NB>
__device__ __forceinline__ void md5_super_block(uint32_t (&h)[4], const void *pcm, uint32_t fmt, uint64_t t0)
{
	uint32_t mm[NB * 16];
	if(NB == 1) {
#pragma unroll
		for(int j = 0; j < 16; j++) mm[j] = (md5_val(pcm, fmt, t0 + 4 * j) & 0xffu) | ((md5_val(pcm, fmt, t0 + 4 * j + 1) & 0xffu) << 8) | ((md5_val(pcm, fmt, t0 + 4 * j + 2) & 0xffu) << 16) | (md5_val(pcm, fmt, t0 + 4 * j + 3) << 24);
	}
	else if(NB == 2) {
#pragma unroll
		for(int j = 0; j < 32; j++) mm[j] = (md5_val(pcm, fmt, t0 + 2 * j) & 0xffffu) | (md5_val(pcm, fmt, t0 + 2 * j + 1) << 16);
	}
	else {
#pragma unroll
		for(int j = 0; j < 16; j++) { // four 3-byte samples -> three words
			const uint32_t a = md5_val(pcm, fmt, t0 + 4 * j) & 0xffffffu, b = md5_val(pcm, fmt, t0 + 4 * j + 1) & 0xffffffu,
			               c = md5_val(pcm, fmt, t0 + 4 * j + 2) & 0xffffffu, d = md5_val(pcm, fmt, t0 + 4 * j + 3) & 0xffffffu;
			mm[3 * j] = a | (b << 24); mm[3 * j + 1] = (b >> 8) | (c << 16); mm[3 * j + 2] = (c >> 16) | (d << 8);
		}
	}
#pragma unroll
	for(int blk = 0; blk < NB; blk++) {
		uint32_t m[16];
#pragma unroll
		for(int i = 0; i < 16; i++) m[i] = mm[blk * 16 + i];
		md5_block(h, m);
	}
}

// 16 bytes from an address that is only as aligned as its samples (the hardware takes such loads; the type says so)
struct __attribute__((packed, aligned(2))) Md5Raw16 { uint32_t w[4]; };
__device__ __forceinline__ uint4 md5_load16(const uint8_t *p) { const Md5Raw16 v = *(const Md5Raw16*)p; return make_uint4(v.w[0], v.w[1], v.w[2], v.w[3]); }

// The bulk of a stream, 64 values at a time, for two-byte samples held as int16 (the memory IS the byte stream: 128 bytes,
// two blocks) and three-byte samples held as int32 (256 bytes of memory, 192 of stream, three blocks).  A lane fetches its
// bytes 16 at a time - every lane reads a stream of its own, so a load instruction touches 64 cache lines whatever its
// width - and the next 64 values are on their way while the blocks of these are worked through: the chain of a block's
// 64 steps is all a lane has to do, and nothing else would hide the latency of its loads.
template<int RAWVEC>
struct Md5Chunk { uint4 v[RAWVEC]; };
template<int RAWVEC>
__device__ __forceinline__ Md5Chunk<RAWVEC> md5_fetch(const uint8_t *p)
{
	Md5Chunk<RAWVEC> c;
#pragma unroll
	for(int j = 0; j < RAWVEC; j++) c.v[j] = md5_load16(p + 16 * j);
	return c;
}
__device__ __forceinline__ void md5_two_blocks(uint32_t (&h)[4], const Md5Chunk<8> &c)
{
#pragma unroll
	for(int blk = 0; blk < 2; blk++) {
		uint32_t m[16];
#pragma unroll
		for(int i = 0; i < 4; i++) { const uint4 q = c.v[blk * 4 + i]; m[4 * i] = q.x; m[4 * i + 1] = q.y; m[4 * i + 2] = q.z; m[4 * i + 3] = q.w; }
		md5_block(h, m);
	}
}
__device__ __forceinline__ void md5_three_blocks(uint32_t (&h)[4], const Md5Chunk<16> &c)
{
	uint32_t mm[48];
#pragma unroll
	for(int j = 0; j < 16; j++) { // four 3-byte samples -> three words
		const uint4 q = c.v[j];
		const uint32_t a = q.x & 0xffffffu, b = q.y & 0xffffffu, cc = q.z & 0xffffffu, d = q.w & 0xffffffu;
		mm[3 * j] = a | (b << 24); mm[3 * j + 1] = (b >> 8) | (cc << 16); mm[3 * j + 2] = (cc >> 16) | (d << 8);
	}
#pragma unroll
	for(int blk = 0; blk < 3; blk++) {
		uint32_t m[16];
#pragma unroll
		for(int i = 0; i < 16; i++) m[i] = mm[blk * 16 + i];
		md5_block(h, m);
	}
}

// one stream: [s0, s1) of pcm in wide samples -> 16 bytes at digest.  Every lane of a workgroup's first wave calls it (a
// lane without a stream with valid = false and an empty range); the second wave runs md5_fetcher beside it.
struct Md5Shared;
__device__ __forceinline__ void md5_one_stream(const void *__restrict__ pcm, uint32_t fmt, uint32_t channels, uint32_t bps, uint64_t s0, uint64_t s1, uint8_t *__restrict__ digest, bool valid, Md5Shared &sh);

// The bulk of a workgroup's 64 streams comes in through its second wave, 256 bytes of memory per stream and round: sixteen
// lanes take the sixteen 16-byte pieces of one stream's next 256 bytes, so a load instruction touches four streams (four
// pages, eight or twelve cache lines) instead of sixty-four, and the pieces go through LDS rows - two sets, filled and
// read in turn, one barrier per round - to the lane of the first wave that hashes the stream.  (With every hashing lane
// fetching its own stream 16 bytes at a time the kernel asked for an address translation per lane and load, and the coding
// kernels running beside it took three to five times as long as alone; with the hashing lanes doing the shared fetch
// themselves the hash chain, which is all that bounds this kernel, grew by a quarter.)
#define MD5_ROW 17   // uint4 per LDS row: 256 bytes + 16 of padding (rows a power of two apart would share banks)
struct Md5Shared {
	uint4 stage[2][64][MD5_ROW];
	unsigned long long ptr[64];   // where each stream's bulk starts
	uint32_t rounds[64];          // ... and how many rounds of 256 bytes it has
	uint32_t mode, maxr;          // 0: no shared fetch; 4: two-byte samples held as int16 (four blocks a round); 3: three-byte samples held as int32 (three)
};

__device__ __forceinline__ void md5_fetcher(Md5Shared &sh)
{
	const uint32_t lane = threadIdx.x & 63u, sub = lane >> 4, piece = lane & 15u;
	__syncthreads();
	const uint32_t maxr = sh.maxr;
	if(!sh.mode) return;
	// where this lane fetches for the sixteen streams it serves (streams 4 i + sub) and how many rounds each has left
	const uint8_t *pp[16];
	uint32_t cnt[16];
#pragma unroll
	for(int i = 0; i < 16; i++) { pp[i] = (const uint8_t*)(uintptr_t)sh.ptr[4 * i + (int)sub] + piece * 16u; cnt[i] = sh.rounds[4 * i + (int)sub]; }
	uint4 pre[16];
	auto fetch_round = [&]() {
#pragma unroll
		for(int i = 0; i < 16; i++) {
			pre[i] = cnt[i] ? md5_load16(pp[i]) : make_uint4(0u, 0u, 0u, 0u);
			if(cnt[i]) { pp[i] += 256; cnt[i]--; }
		}
	};
	fetch_round();
	for(uint32_t r = 0; r < maxr; r++) {
#pragma unroll
		for(int i = 0; i < 16; i++) sh.stage[r & 1u][4 * i + (int)sub][piece] = pre[i];
		fetch_round();      // the next round's loads are in flight while the first wave hashes this one
		__syncthreads();
	}
}

// the first wave's side: BLOCKS as Md5Shared::mode; `rounds` of this lane's stream, maxr the most of any lane
template<int BLOCKS>
__device__ __forceinline__ void md5_bulk_hash(uint32_t (&h)[4], uint32_t rounds, uint32_t maxr, Md5Shared &sh)
{
	const uint32_t lane = threadIdx.x & 63u;
	for(uint32_t r = 0; r < maxr; r++) {
		__syncthreads();
		if(r < rounds) {
			Md5Chunk<16> cur;
#pragma unroll
			for(int j = 0; j < 16; j++) cur.v[j] = sh.stage[r & 1u][lane][j];
			if(BLOCKS == 4) {
				Md5Chunk<8> a, b;
#pragma unroll
				for(int j = 0; j < 8; j++) { a.v[j] = cur.v[j]; b.v[j] = cur.v[8 + j]; }
				md5_two_blocks(h, a); md5_two_blocks(h, b);
			}
			else md5_three_blocks(h, cur);
		}
	}
}

__global__ void __launch_bounds__(128) fa_md5_kernel(const void *__restrict__ pcm, uint32_t fmt, uint32_t channels, uint32_t bps,
	const uint64_t *__restrict__ starts, const uint64_t *__restrict__ ends, uint32_t nstreams, uint8_t *__restrict__ digests)
{
	__shared__ Md5Shared sh;
	if(threadIdx.x >= 64u) { md5_fetcher(sh); return; }
	const uint32_t sidx = blockIdx.x * 64u + threadIdx.x;
	const bool valid = sidx < nstreams;
	md5_one_stream(pcm, fmt, channels, bps, valid ? starts[sidx] : 0ull, valid ? ends[sidx] : 0ull, digests + (size_t)(valid ? sidx : 0u) * 16u, valid, sh);
}

// the same for streams of several buffers and formats: what a lane hashes comes from its own record (a group's lanes
// are next to one another, so a wave is nearly always of one format)
struct FaMd5Rec { const void *pcm; uint8_t *digest; uint64_t s0, s1; uint32_t fmt, channels, bps, pad; };
__global__ void __launch_bounds__(128) fa_md5_multi_kernel(const FaMd5Rec *__restrict__ recs, uint32_t nstreams)
{
	__shared__ Md5Shared sh;
	if(threadIdx.x >= 64u) { md5_fetcher(sh); return; }
	const uint32_t sidx = blockIdx.x * 64u + threadIdx.x;
	FaMd5Rec r;
	r.pcm = 0; r.digest = 0; r.s0 = 0; r.s1 = 0; r.fmt = FLACAMD_PCM_S16; r.channels = 1; r.bps = 16; r.pad = 0;
	if(sidx < nstreams) r = recs[sidx];
	const bool valid = r.pcm != 0;   // (else filler up to a whole wave)
	if(!valid) { r.s0 = 0; r.s1 = 0; }
	md5_one_stream(r.pcm, r.fmt, r.channels, r.bps, r.s0, r.s1, r.digest, valid, sh);
}

__device__ __forceinline__ void md5_one_stream(const void *__restrict__ pcm, uint32_t fmt, uint32_t channels, uint32_t bps, uint64_t s0, uint64_t s1, uint8_t *__restrict__ digest, bool valid, Md5Shared &sh)
{
	const uint32_t nb = (bps + 7u) / 8u;
	uint32_t h[4] = { 0x67452301u, 0xefcdab89u, 0x98badcfeu, 0x10325476u };
	// bulk: 64 values at a time are a whole number of blocks
	uint64_t t = s0 * channels;
	const uint64_t tend = s1 * channels;
	uint64_t total = 0;
	// ... through the second wave where the streams are all of one of the two common layouts (lanes without a stream go along)
	{
		const bool is16 = nb == 2u && fmt == FLACAMD_PCM_S16, is24 = nb == 3u && fmt == FLACAMD_PCM_S32;
		const uint32_t mode = __all((is16 || !valid) ? 1 : 0) ? 4u : __all((is24 || !valid) ? 1 : 0) ? 3u : 0u;
		const uint32_t per = mode == 4u ? 128u : 64u;   // values per round of 256 bytes
		const uint64_t nr = (mode && valid) ? (tend - t) / per : 0ull;
		const uint32_t rounds = nr < 0xffffffffull ? (uint32_t)nr : 0xffffffffu;
		uint32_t maxr = rounds;
#pragma unroll
		for(int o = 32; o > 0; o >>= 1) { const uint32_t v = (uint32_t)__shfl_xor((int)maxr, o, 64); maxr = v > maxr ? v : maxr; }
		const uint32_t lane = threadIdx.x & 63u;
		sh.ptr[lane] = (unsigned long long)(uintptr_t)((const uint8_t*)pcm + t * (mode == 4u ? 2u : 4u));
		sh.rounds[lane] = rounds;
		if(lane == 0u) { sh.mode = maxr ? mode : 0u; sh.maxr = maxr; }
		__syncthreads();
		if(mode == 4u && maxr) md5_bulk_hash<4>(h, rounds, maxr, sh);
		else if(mode == 3u && maxr) md5_bulk_hash<3>(h, rounds, maxr, sh);
		t += (uint64_t)rounds * per; total += (uint64_t)rounds * (mode == 4u ? 256u : 192u);
	}
	if(nb == 2u && fmt == FLACAMD_PCM_S16 && t + 64u <= tend) {
		const uint8_t *p = (const uint8_t*)pcm + t * 2u;
		Md5Chunk<8> nxt = md5_fetch<8>(p);
		for(; t + 64u <= tend; t += 64u) {
			const Md5Chunk<8> cur = nxt;
			p += 128;
			if(t + 128u <= tend) nxt = md5_fetch<8>(p);
			md5_two_blocks(h, cur);
			total += 128u;
		}
	}
	else if(nb == 3u && fmt == FLACAMD_PCM_S32 && t + 64u <= tend) {
		const uint8_t *p = (const uint8_t*)pcm + t * 4u;
		Md5Chunk<16> nxt = md5_fetch<16>(p);
		for(; t + 64u <= tend; t += 64u) {
			const Md5Chunk<16> cur = nxt;
			p += 256;
			if(t + 128u <= tend) nxt = md5_fetch<16>(p);
			md5_three_blocks(h, cur);
			total += 192u;
		}
	}
	for(; t + 64u <= tend; t += 64u) {
		if(nb == 1u) md5_super_block<1>(h, pcm, fmt, t);
		else if(nb == 2u) md5_super_block<2>(h, pcm, fmt, t);
		else md5_super_block<3>(h, pcm, fmt, t);
		total += 64u * nb;
	}
	// tail and padding: byte by byte
	uint32_t m[16];
#pragma unroll
	for(int i = 0; i < 16; i++) m[i] = 0;
	uint32_t fill = 0;            // bytes in the current block
	auto push = [&](uint32_t byte) {
		// static register indexing: the word index is resolved by selects
		const uint32_t wi = fill >> 2, sh = (fill & 3u) * 8u;
#pragma unroll
		for(int i = 0; i < 16; i++) if((uint32_t)i == wi) m[i] |= byte << sh;
		fill++; total++;
		if(fill == 64u) {
			md5_block(h, m);
#pragma unroll
			for(int i = 0; i < 16; i++) m[i] = 0;
			fill = 0;
		}
	};
	for(; t < tend; t++) {
		const uint32_t v = md5_val(pcm, fmt, t);
		for(uint32_t k = 0; k < nb; k++) push((v >> (8u * k)) & 0xffu);
	}
	const uint64_t bits = total * 8u;
	push(0x80u);
	while(fill != 56u) push(0u);
	for(uint32_t k = 0; k < 8; k++) push((uint32_t)(bits >> (8u * k)) & 0xffu);
	if(valid) for(uint32_t i = 0; i < 16; i++) digest[i] = (uint8_t)(h[i >> 2] >> (8u * (i & 3u)));
}

extern "C" int flacamd_md5_device(int device, const void *d_pcm, int pcm_format, uint32_t channels, uint32_t bits_per_sample,
	const uint64_t *h_offsets, uint32_t nstreams, uint8_t *h_digests, void *stream)
{
	if(!d_pcm || !h_offsets || !h_digests || nstreams == 0 || channels == 0 || channels > FLACAMD_MAX_CHANNELS) return FLACAMD_E_INVALID_ARGUMENT;
	if(pcm_format != FLACAMD_PCM_S32 && pcm_format != FLACAMD_PCM_S16) return FLACAMD_E_INVALID_ARGUMENT;
	if(bits_per_sample < 1 || bits_per_sample > 32) return FLACAMD_E_INVALID_ARGUMENT;   // a sample is hashed as 1..4 bytes
	int ndev = 0;
	if(hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return FLACAMD_E_NO_DEVICE;
	hipStream_t st = (hipStream_t)stream;
	if(hipSetDevice(device) != hipSuccess) return FLACAMD_E_DEVICE;
	uint64_t *d_off = 0; uint8_t *d_dig = 0;
	if(hipMalloc((void**)&d_off, (size_t)(nstreams + 1) * 8) != hipSuccess) return FLACAMD_E_OUT_OF_MEMORY;
	if(hipMalloc((void**)&d_dig, (size_t)nstreams * 16) != hipSuccess) { (void)hipFree(d_off); return FLACAMD_E_OUT_OF_MEMORY; }
	int rc = FLACAMD_OK;
	if(hipMemcpyAsync(d_off, h_offsets, (size_t)(nstreams + 1) * 8, hipMemcpyHostToDevice, st) != hipSuccess) rc = FLACAMD_E_DEVICE;
	if(rc == FLACAMD_OK) {
		hipLaunchKernelGGL(fa_md5_kernel, dim3((nstreams + 63u) / 64u), dim3(128), 0, st, d_pcm, (uint32_t)pcm_format, channels, bits_per_sample, d_off, d_off + 1, nstreams, d_dig);
		if(hipMemcpyAsync(h_digests, d_dig, (size_t)nstreams * 16, hipMemcpyDeviceToHost, st) != hipSuccess) rc = FLACAMD_E_DEVICE;
		if(hipStreamSynchronize(st) != hipSuccess) rc = FLACAMD_E_DEVICE;
	}
	(void)hipFree(d_off); (void)hipFree(d_dig);
	return rc;
}

// ---- the same without any allocation --------------------------------------------------
// For callers that cannot call hipMalloc (tight serving loops, harnesses that forbid allocation after start-up):
// the stream offsets and the digests are staged in a fixed array that is part of the code object, one copy
// per GPU at its own address.  The array's name is only a host-side handle: the device address is looked up
// once per device with hipGetSymbolAddress and that pointer is what kernels and copies receive.  The call
// still copies from and to the caller's host buffers and waits for its stream, so it is not for use inside a
// stream capture.  Calls on one GPU share the array: they are serialised here (one lock per device, held from
// the upload of the offsets to the end of the wait), so overlapping callers wait for one another instead of
// overwriting each other's offsets and digests.
#define FA_MD5_STATIC_STREAMS 65536u
#define FA_MD5_STATIC_OFF_BYTES (((size_t)FA_MD5_STATIC_STREAMS + 1u) * 8u + 248u)   // the digests start 256-byte aligned
static __device__ __align__(256) unsigned char scratch[FA_MD5_STATIC_OFF_BYTES + (size_t)FA_MD5_STATIC_STREAMS * 16u];

#define FA_MD5_MAX_DEVICES 64
static std::mutex g_scratch_lock;
static std::mutex g_scratch_busy[FA_MD5_MAX_DEVICES];   // held by the call that is using the device's array
static unsigned char *g_scratch_ptr[FA_MD5_MAX_DEVICES];

// device address of this GPU's copy of `scratch` (the device must be current); NULL on failure
static unsigned char *md5_scratch_of(int device)
{
	std::lock_guard<std::mutex> hold(g_scratch_lock);
	if(!g_scratch_ptr[device]) {
		void *p = 0;
		if(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)) != hipSuccess || !p) { (void)hipGetLastError(); return 0; }
		g_scratch_ptr[device] = (unsigned char*)p;
	}
	return g_scratch_ptr[device];
}

extern "C" uint32_t flacamd_md5_device_static_max_streams(void) { return FA_MD5_STATIC_STREAMS; }

extern "C" int flacamd_md5_device_static(int device, const void *d_pcm, int pcm_format, uint32_t channels, uint32_t bits_per_sample,
	const uint64_t *h_offsets, uint32_t nstreams, uint8_t *h_digests, void *stream)
{
	if(!d_pcm || !h_offsets || !h_digests || nstreams == 0 || channels == 0 || channels > FLACAMD_MAX_CHANNELS) return FLACAMD_E_INVALID_ARGUMENT;
	if(pcm_format != FLACAMD_PCM_S32 && pcm_format != FLACAMD_PCM_S16) return FLACAMD_E_INVALID_ARGUMENT;
	if(bits_per_sample < 1 || bits_per_sample > 32) return FLACAMD_E_INVALID_ARGUMENT;
	if(nstreams > FA_MD5_STATIC_STREAMS) return FLACAMD_E_CAPACITY;   // more streams than the fixed array stages: split the batch, or use flacamd_md5_device
	int ndev = 0;
	if(hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev || device >= FA_MD5_MAX_DEVICES) return FLACAMD_E_NO_DEVICE;
	hipStream_t st = (hipStream_t)stream;
	if(hipSetDevice(device) != hipSuccess) return FLACAMD_E_DEVICE;
	unsigned char *base = md5_scratch_of(device);
	if(!base) return FLACAMD_E_DEVICE;
	uint64_t *d_off = (uint64_t*)base;
	uint8_t *d_dig = base + FA_MD5_STATIC_OFF_BYTES;
	std::lock_guard<std::mutex> busy(g_scratch_busy[device]);
	// nothing in the array has to start at zero: the kernel reads offsets[0..nstreams] and writes every byte of
	// digests[0..nstreams), both overwritten here, and nothing beyond them is copied out
	if(hipMemcpyAsync(d_off, h_offsets, (size_t)(nstreams + 1) * 8, hipMemcpyHostToDevice, st) != hipSuccess) return FLACAMD_E_DEVICE;
	hipLaunchKernelGGL(fa_md5_kernel, dim3((nstreams + 63u) / 64u), dim3(128), 0, st, d_pcm, (uint32_t)pcm_format, channels, bits_per_sample, d_off, d_off + 1, nstreams, d_dig);
	int rc = hipGetLastError() == hipSuccess ? FLACAMD_OK : FLACAMD_E_DEVICE;
	if(rc == FLACAMD_OK && hipMemcpyAsync(h_digests, d_dig, (size_t)nstreams * 16, hipMemcpyDeviceToHost, st) != hipSuccess) rc = FLACAMD_E_DEVICE;
	if(hipStreamSynchronize(st) != hipSuccess) rc = FLACAMD_E_DEVICE;
	return rc;
}

// for the clip batch entry point of engine.hip: streams [starts[i], ends[i]) of d_pcm, digests to device memory, no synchronisation
extern "C" void fa_launch_md5(const void *d_pcm, uint32_t pcm_format, uint32_t channels, uint32_t bits_per_sample, const uint64_t *d_starts, const uint64_t *d_ends,
	uint32_t nstreams, uint8_t *d_digests, hipStream_t st)
{
	hipLaunchKernelGGL(fa_md5_kernel, dim3((nstreams + 63u) / 64u), dim3(128), 0, st, d_pcm, pcm_format, channels, bits_per_sample, d_starts, d_ends, nstreams, d_digests);
}

// ---- the digests of several clip batches in one launch
namespace {
std::mutex g_md5_groups_lock;
struct Md5GroupsBuf { void *h; void *d; size_t cap; hipEvent_t done; bool pending; };   // done: recorded behind the last launch that reads h and d
Md5GroupsBuf g_md5_groups_buf[64];
}

extern "C" int flacamd_md5_groups_device(int device, uint32_t ngroups, const flacamd_md5_group *groups, void *stream)
{
	if(!groups || ngroups == 0) return FLACAMD_E_INVALID_ARGUMENT;
	int ndev = 0;
	if(hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev || device >= 64) return FLACAMD_E_NO_DEVICE;
	size_t total = 0;
	for(uint32_t g = 0; g < ngroups; g++) {
		const flacamd_md5_group &q = groups[g];
		if(!q.d_pcm || !q.h_start || !q.h_len || !q.d_digests || q.nclips == 0 || q.channels == 0 || q.channels > FLACAMD_MAX_CHANNELS) return FLACAMD_E_INVALID_ARGUMENT;
		if(q.pcm_format != FLACAMD_PCM_S32 && q.pcm_format != FLACAMD_PCM_S16) return FLACAMD_E_INVALID_ARGUMENT;
		total += ((size_t)q.nclips + 63u) & ~(size_t)63u;   // every group starts a wave of its own
	}
	if(total > 0xffffffffull) return FLACAMD_E_CAPACITY;
	if(hipSetDevice(device) != hipSuccess) return FLACAMD_E_DEVICE;
	std::lock_guard<std::mutex> hold(g_md5_groups_lock);
	Md5GroupsBuf &b = g_md5_groups_buf[device];
	const size_t bytes = total * sizeof(FaMd5Rec);
	// the previous call's copy reads b.h and its kernel reads b.d when the stream gets to them, not when the call
	// returned: wait for both before the staging buffers are written again or freed
	if(b.pending) { if(hipEventSynchronize(b.done) != hipSuccess) return FLACAMD_E_DEVICE; b.pending = false; }
	if(!b.done && hipEventCreateWithFlags(&b.done, hipEventDisableTiming) != hipSuccess) { b.done = 0; return FLACAMD_E_DEVICE; }
	if(b.cap < bytes) {   // grow-only staging: none of this on the steady path
		if(b.h) (void)hipHostFree(b.h);
		if(b.d) (void)hipFree(b.d);
		b.h = 0; b.d = 0; b.cap = 0;
		if(hipHostMalloc(&b.h, bytes, hipHostMallocDefault) != hipSuccess || hipMalloc(&b.d, bytes) != hipSuccess) {
			if(b.h) (void)hipHostFree(b.h);
			b.h = 0; (void)hipGetLastError();
			return FLACAMD_E_OUT_OF_MEMORY;
		}
		b.cap = bytes;
	}
	FaMd5Rec *r = (FaMd5Rec*)b.h;
	size_t at = 0;
	for(uint32_t g = 0; g < ngroups; g++) {
		const flacamd_md5_group &q = groups[g];
		for(uint32_t i = 0; i < q.nclips; i++, at++) {
			r[at].pcm = q.d_pcm; r[at].digest = q.d_digests + (size_t)i * 16u; r[at].s0 = q.h_start[i]; r[at].s1 = q.h_start[i] + q.h_len[i];
			r[at].fmt = (uint32_t)q.pcm_format; r[at].channels = q.channels; r[at].bps = q.bits_per_sample; r[at].pad = 0;
		}
		for(; at & 63u; at++) memset(&r[at], 0, sizeof r[at]);
	}
	hipStream_t st = (hipStream_t)stream;
	if(hipMemcpyAsync(b.d, b.h, bytes, hipMemcpyHostToDevice, st) != hipSuccess) return FLACAMD_E_DEVICE;
	hipLaunchKernelGGL(fa_md5_multi_kernel, dim3((uint32_t)(total / 64u)), dim3(128), 0, st, (const FaMd5Rec*)b.d, (uint32_t)total);
	if(hipGetLastError() != hipSuccess) return FLACAMD_E_DEVICE;
	if(hipEventRecord(b.done, st) == hipSuccess) b.pending = true;
	else if(hipStreamSynchronize(st) != hipSuccess) return FLACAMD_E_DEVICE;
	return FLACAMD_OK;
}
