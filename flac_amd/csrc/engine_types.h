// Shared host/device definitions of the gfx950 FLAC frame engine.
#pragma once
#include <stdint.h>
#include "../../include/flac_amd_engine.h"

#define FA_MAX_FIXED_ORDER 4u
#define FA_MIN_QLP_PRECISION 5u
#define FA_MAX_QLP_PRECISION 15u
#define FA_MAX_PARTS (1u << FLACAMD_MAX_PARTITION_ORDER)
#define FA_DBG_WORDS 131072u
#define FA_PACK_LARGE_WORDS 38000u   // bit buffer of the pack stage's second pass (frames far beyond the verbatim size)
#define FA_LDS_BLOCKSIZE 16384u   // largest block whose padded 32-bit signal shares LDS with the other per-workgroup tables

// subframe kinds (bitstream order of the type codes)
enum { FA_SUB_CONSTANT = 0, FA_SUB_VERBATIM = 1, FA_SUB_FIXED = 2, FA_SUB_LPC = 3 };

// Everything a kernel needs to know about the stream and the batch; passed by value
// (lands in SGPRs).
struct FaCfg {
	uint32_t channels;        // 1..8
	uint32_t bps;             // stream bits per sample
	uint32_t rate;
	uint32_t N;               // stream blocksize
	uint32_t nsig;            // signals analysed per frame: 4 (L,R,M,S) for mid/side stereo, else channels
	uint32_t midside;         // 1 when nsig == 4 comes from mid/side
	uint32_t loose;           // loose mid/side
	uint32_t loose_period;    // frames per decision period (stream_encoder.c:878-890)
	uint32_t pcm_format;      // FLACAMD_PCM_*
	uint32_t F;               // frames in this batch
	uint32_t last_n;          // samples in the last frame of the batch
	uint32_t last_fractional; // last frame is the short final block of the stream
	uint32_t multi;           // the batch holds several complete streams: frame lengths and numbers come from FaWork::frame_len / frame_num; 2 = each stream's frames are also moved FaWork::frame_shift bytes further in the output (room for the stream heads)
	uint32_t skip_full;       // general search kernel: full-length frames are left to the register-resident kernel; its workgroups then cover
	                          // only the nshort frames listed in FaWork::short_list (the short final blocks of a batch of several streams)
	uint32_t nshort;
	uint32_t spill;           // blocks above FA_LDS_BLOCKSIZE samples: the general kernels keep their copy of the signal in FaWork::xs_spill instead of LDS
	uint32_t first_frame;     // frame number of frame 0 of the batch
	uint32_t prev_assignment;
	uint32_t max_lpc_order, qlp_precision, prec_search, exhaustive;
	uint32_t min_po, max_po;  // as configured (before per-frame / per-order limiting)
	uint32_t disable_constant, disable_fixed, disable_verbatim;
	uint32_t num_apod;
	uint32_t lag;             // max_lpc_order + 1
	uint32_t max_cand;        // candidate slots per signal
	uint32_t cand_stride;     // dwords per candidate record: header + max(max_lpc_order, 1) coefficients + their sum, rounded up to a multiple of 4
	uint32_t k_limit;         // 15, or 31 when stream bps > 16 (stream_encoder.c:3200)
	uint32_t wide_fixed;      // 64-bit fixed-predictor error sums (stream_encoder.c:894)
	uint32_t slot_bytes;      // bytes reserved per frame in the slot buffer (multiple of 16)
	uint32_t G;               // F * nsig
	uint32_t debug;           // development ablation switches (FLACAMD_DEBUG), 0 in production
	uint32_t direct_out;      // the register-resident pack kernel writes every frame at its final place in the caller's buffer (FaWork::out), found by a look-back over the frame sizes: no copy out of per-frame slots
	uint32_t fixed_psum;      // the 16-bit stereo preparation kernel leaves the fixed predictors' |residual| sums per finest partition (FaWork::fixed_psum) and the register-resident search kernel takes them from there
};

// Candidate header word written by the model kernel, read by the search kernel.
//  [0]    kind: 0 fixed, 1 lpc
//  [6:1]  order (0..32)
//  [11:7] coefficient precision (lpc)
//  [16:12] shift (lpc, 0..15)
//  [21:17] suggested Rice parameter for partition order 0
//  [22]   64-bit prediction sum (stream_encoder.c:3597-3603)
//  [23]   bounded: (sum of |coefficient| << (subframe bps - 1)) >> shift < 2^27, so no prediction of this candidate can
//         reach 2^27 in magnitude on samples of this width: x - prediction does not wrap, and a thread's sixteen
//         |residual|s add up within 32 bits.  The search kernels take |x - prediction| with the unsigned
//         absolute-difference instruction for such candidates; the others go the reference's way step by step.
#define FA_CAND_KIND(h)  ((h) & 1u)
#define FA_CAND_ORDER(h) (((h) >> 1) & 63u)
#define FA_CAND_PREC(h)  (((h) >> 7) & 31u)
#define FA_CAND_SHIFT(h) (((h) >> 12) & 31u)
#define FA_CAND_K(h)     (((h) >> 17) & 31u)
#define FA_CAND_WIDE(h)  (((h) >> 22) & 1u)
#define FA_CAND_BOUNDED(h) (((h) >> 23) & 1u)
#define FA_CAND_PACK(kind, order, prec, shift, k, wide) \
	((uint32_t)(kind) | ((uint32_t)(order) << 1) | ((uint32_t)(prec) << 7) | ((uint32_t)(shift) << 12) | ((uint32_t)(k) << 17) | ((uint32_t)(wide) << 22))

// Winner record of one signal (search kernel -> pack kernel), array of structs.
//  hdr[0]: [1:0] type  [7:2] order  [12:8] precision  [17:13] shift  [21:18] partition order
//          [22] rice2  [27:23] wasted bits  [28] 64-bit prediction sum
//  hdr[1]: estimated bits of the subframe (the value the reference compares)
//  hdr[2]: subframe bits per sample
//  hdr[3]: index of the winning candidate slot (0xffffffff for verbatim / constant)
struct FaWinner {
	uint32_t hdr[4];
	int32_t  q[FLACAMD_MAX_LPC_ORDER];
	uint8_t  k[FA_MAX_PARTS];
};
#define FA_WIN_TYPE(h)   ((h) & 3u)
#define FA_WIN_ORDER(h)  (((h) >> 2) & 63u)
#define FA_WIN_PREC(h)   (((h) >> 8) & 31u)
#define FA_WIN_SHIFT(h)  (((h) >> 13) & 31u)
#define FA_WIN_PO(h)     (((h) >> 18) & 15u)
#define FA_WIN_RICE2(h)  (((h) >> 22) & 1u)
#define FA_WIN_WASTED(h) (((h) >> 23) & 31u)
#define FA_WIN_WIDE(h)   (((h) >> 28) & 1u)

// What the pack stage needs to know about one frame, gathered in one place by the assignment
// kernel so that the pack workgroup fetches it in a single round of loads: per coded subframe the
// winner's header word, bits per sample and signal index, its coefficients and its Rice parameters
// (spread to the finest configured partition order).
struct FaPackRec {
	uint32_t hdr0[2], sbps[2], sig[2], assignment, pad;
	int32_t  q[2][12];
	uint8_t  k[2][FA_MAX_PARTS];
};

// Device workspace pointers of one encoder.
struct FaWork {
	// prep
	uint8_t  *wasted;      // [G]
	uint8_t  *is_const;    // [G] bit 0: all samples equal; bit 1: every sample fits 16 bits (set by the 16-bit stereo preparation kernel only)
	uint64_t *fixed_err;   // [5][G]
	uint32_t *fixed_psum;  // [G][5][64] sum of |k-th difference| over the residual samples of each finest partition, before the wasted-bit shift (FaCfg::fixed_psum)
	// autocorrelation
	float    *autoc;       // [num_apod][lag][G]
	// model
	uint32_t *ncand;       // [G]
	// candidate records, one signal's records contiguous: [G][max_cand][cand_stride] dwords, each record
	// = header word, the coefficients (max(max_lpc_order, 1) rows), then the sum of the coefficients of a linear
	// predictor; cand_q points one dword past cand_hdr so that both index alike
	uint32_t *cand_hdr;
	int32_t  *cand_q;
	// search
	FaWinner *winner;      // [G]
	// pack
	uint8_t  *slots;       // [F][slot_bytes]
	uint32_t *frame_bytes; // [F]
	uint32_t *assignment;  // [F]
	FaPackRec *packrec;    // [F] (mono / stereo fast path)
	uint32_t *status;      // [1] nonzero = some frame overflowed its slot
	uint64_t *offsets;     // [F+1] exclusive scan of frame_bytes
	uint32_t *frame_len;   // [F] samples per frame, multi-stream batches only
	uint32_t *frame_num;   // [F] frame number within its stream, multi-stream batches only
	uint32_t *frame_shift; // [F] bytes added to the frame's output offset (FaCfg::multi == 2)
	int32_t  *xs_spill;    // [G][xs_words_for(N)] signal copies of the general kernels when FaCfg::spill (else unused)
	uint32_t *short_list;  // [nshort] frames of a batch of several streams that are shorter than the blocksize (FaCfg::skip_full)
	const float *window;   // [num_apod][N]
	const uint16_t *crc16_table4; // [4][256] CRC-16 of one byte followed by 0..3 zero bytes (crc.c:77-110 polynomial)
	const uint16_t *crc16_xpw;    // [bitbuf words + 1] x^(32 w) mod the CRC-16 polynomial
	unsigned long long *dbg;      // [FA_DBG_WORDS] development timestamps (FLACAMD_DEBUG & 0x1000), unused in production
	// FaCfg::direct_out (set per call)
	unsigned long long *lb;       // [F] look-back state of every frame, zeroed before the pack kernel: bits 63..62 = 0 nothing yet, 1 the frame's own size, 2 the size of all frames up to and including it; the rest the value.
	                              // Behind them u32 [F]: nonzero for a frame that went to its slot because its place was not known in time (fa_place_late_kernel)
	uint8_t  *out;                // the caller's output buffer
	unsigned long long out_cap;   // its size
	uint32_t *frame_bytes_out;    // the caller's array of frame sizes (may be null)
};

#ifdef __HIPCC__
#include <hip/hip_runtime.h>

// number of samples in frame f of the batch
__device__ __forceinline__ uint32_t fa_frame_len(const FaCfg &c, const FaWork &w, uint32_t f) { return c.multi ? w.frame_len[f] : ((f + 1 == c.F) ? c.last_n : c.N); }
// frame number coded in its header (streams of a multi-stream batch each start at 0)
__device__ __forceinline__ uint32_t fa_frame_num(const FaCfg &c, const FaWork &w, uint32_t f) { return c.multi ? w.frame_num[f] : c.first_frame + f; }
// the short final block of a stream: coded with partition order 0 (stream_encoder.c:2934)
__device__ __forceinline__ bool fa_frame_fractional(const FaCfg &c, const FaWork &w, uint32_t f) { return c.multi ? w.frame_len[f] != c.N : (f + 1 == c.F && c.last_fractional); }

// Sample `i` of channel `ch` of the wide-sample stream starting at batch sample 0.
__device__ __forceinline__ int32_t fa_pcm(const void *pcm, uint32_t fmt, uint32_t channels, uint64_t wide, uint32_t ch)
{
	const uint64_t idx = wide * channels + ch;
	return fmt == FLACAMD_PCM_S16 ? (int32_t)((const int16_t*)pcm)[idx] : ((const int32_t*)pcm)[idx];
}

// Signal `s` of a frame at wide-sample index `wide`, before wasted-bit removal.
// Mid/side as stream_encoder.c:2051-2058: mid = (l + r) >> 1 (arithmetic), side = l - r.
__device__ __forceinline__ int32_t fa_signal(const void *pcm, const FaCfg &c, uint32_t s, uint64_t wide)
{
	if(c.midside) {
		if(c.pcm_format == FLACAMD_PCM_S16) {
			const uint32_t lr = ((const uint32_t*)pcm)[wide];
			const int32_t l = (int16_t)(lr & 0xffffu), r = (int32_t)lr >> 16;
			return s == 0 ? l : s == 1 ? r : s == 2 ? ((l + r) >> 1) : (l - r);
		}
		else {
			const int2 lr = ((const int2*)pcm)[wide];
			const int32_t l = lr.x, r = lr.y;
			return s == 0 ? l : s == 1 ? r : s == 2 ? (int32_t)((uint32_t)l + (uint32_t)r) >> 1 : (int32_t)((uint32_t)l - (uint32_t)r);
		}
	}
	return fa_pcm(pcm, c.pcm_format, c.channels, wide, s);
}

// dword index of the record of candidate `slot` of signal g (header at cand_hdr[idx], coefficient j at cand_q[idx + j])
__device__ __forceinline__ size_t fa_cand_idx(const FaCfg &c, uint32_t g, uint32_t slot) { return ((size_t)g * c.max_cand + slot) * c.cand_stride; }

// Workgroups are handed to the eight XCDs round robin (id mod 8) and each XCD has its own L2.  The signals of one
// frame are formed from the same PCM bytes, so a launch with one workgroup per (frame, signal) gives the ids
// b, b + 8, b + 16, ... to the signals of ONE frame: the frame is then fetched from memory once, by one XCD,
// instead of once per signal.  Returns the signal index g for workgroup b of `total` (a multiple of nsig).
__device__ __forceinline__ uint32_t fa_xcd_signal(uint32_t b, uint32_t total, uint32_t nsig)
{
	const uint32_t span = 8u * nsig, whole = (total / span) * span;
	if(b >= whole) return b;
	const uint32_t blk = b / span, r = b - blk * span;
	return (blk * 8u + (r & 7u)) * nsig + (r >> 3);
}

__device__ __forceinline__ uint32_t fa_ilog2(uint32_t v) { return 31u - (uint32_t)__clz((int)v); } // v > 0
#endif
