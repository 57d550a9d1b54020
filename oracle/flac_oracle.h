/* TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C99) of the reference encoder/decoder frame path of libFLAC 1.2.1
 * (/root/reference/src/libFLAC).  It exists to CHECK the HIP engine: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product
 * (flac_amd/csrc) never links, loads or calls anything in this directory.
 *
 * Parity status: PINNED.  tests/test_oracle_vs_reference.py compares every frame this
 * code produces with the frames of the real reference (oracle/_ref/libflacref.so, built
 * from the reference's own sources) and tests/golden/ holds reference-generated vectors.
 */
#ifndef FLAC_ORACLE_H
#define FLAC_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FO_MAX_CHANNELS 8
#define FO_MAX_LPC_ORDER 32
#define FO_MAX_APODIZATIONS 32

/* window kinds, in the order of the reference's set_apodization() parser
 * (stream_encoder.c:1546-1585) */
enum {
	FO_WIN_BARTLETT = 0, FO_WIN_BARTLETT_HANN, FO_WIN_BLACKMAN, FO_WIN_BLACKMAN_HARRIS_4TERM,
	FO_WIN_CONNES, FO_WIN_FLATTOP, FO_WIN_GAUSS, FO_WIN_HAMMING, FO_WIN_HANN, FO_WIN_KAISER_BESSEL,
	FO_WIN_NUTTALL, FO_WIN_RECTANGLE, FO_WIN_TRIANGLE, FO_WIN_TUKEY, FO_WIN_WELCH
};

typedef struct {
	uint32_t channels, bits_per_sample, sample_rate, blocksize; /* blocksize 0 = default */
	uint32_t do_mid_side, loose_mid_side;
	uint32_t max_lpc_order, qlp_coeff_precision /* 0 = auto */, do_qlp_coeff_prec_search;
	uint32_t do_exhaustive_model_search;
	uint32_t min_residual_partition_order, max_residual_partition_order;
	uint32_t disable_constant, disable_fixed, disable_verbatim;
	uint32_t num_apodizations;
	uint32_t apod_type[FO_MAX_APODIZATIONS];
	float    apod_param[FO_MAX_APODIZATIONS]; /* tukey p / gauss stddev */
} fo_params;

/* subframe kinds as laid out in the bitstream */
enum { FO_SUB_CONSTANT = 0, FO_SUB_VERBATIM = 1, FO_SUB_FIXED = 2, FO_SUB_LPC = 3 };

/* per-subframe decision as written into the frame (for white-box comparisons) */
typedef struct {
	uint32_t type, order, wasted_bits, bps;
	uint32_t qlp_precision; int32_t qlp_shift; int32_t qlp[FO_MAX_LPC_ORDER];
	uint32_t partition_order, rice2, estimate_bits;
	uint32_t rice_param[64]; /* first 64 only; enough for partition order <= 6 */
} fo_subframe_info;

typedef struct {
	uint32_t blocksize, channel_assignment, frame_number, bytes;
	fo_subframe_info sub[FO_MAX_CHANNELS];
	float autoc[4][FO_MAX_LPC_ORDER + 1];   /* first window, signals L,R,M,S (or channels 0..3) */
	uint64_t fixed_err[4][5];
} fo_frame_info;

/* preset table (stream_encoder.c:127-148): fills the analysis fields for level 0..8 */
void fo_params_preset(fo_params *p, unsigned level);
/* defaults + validation as init_stream_internal_ (stream_encoder.c:667-766); 0 = ok */
int fo_params_resolve(fo_params *p);
/* one apodization window of length L (window.c:49-223) */
void fo_window(uint32_t type, float param, int32_t L, float *w);

/* Whole-stream encode of interleaved int32 PCM: frames only (no "fLaC"/metadata).
 * frame_bytes[i] / frame_samples[i] per frame; infos may be NULL (else >= nframes entries).
 * md5 (may be NULL) receives the MD5 of the PCM as the reference computes it.
 * Returns number of frames, or <0 on error. */
long fo_encode_stream(const fo_params *p, const int32_t *pcm, uint64_t nsamples,
	uint8_t *out, size_t cap, size_t *outlen,
	uint32_t *frame_bytes, uint32_t *frame_samples, uint32_t max_frames,
	fo_frame_info *infos, uint8_t md5[16]);

/* Same, for a run of blocks in the middle of a stream: frames are numbered from
 * first_frame_number; a short last block is coded as the stream's final block only when
 * final_block is nonzero (loose mid/side streams: assumes the run starts a decision period
 * or that the previous assignment was independent). */
long fo_encode_stream_ex(const fo_params *p, const int32_t *pcm, uint64_t nsamples, uint32_t first_frame_number, int final_block,
	uint8_t *out, size_t cap, size_t *outlen,
	uint32_t *frame_bytes, uint32_t *frame_samples, uint32_t max_frames,
	fo_frame_info *infos, uint8_t md5[16]);

/* "UTF-8" coding of a frame number as written into frame headers (bitwriter.c:784-832) */
void fo_utf8(uint32_t v, uint8_t out[8], uint32_t *len);

/* Decode a run of raw frames (no metadata) whose stream parameters are channels/bps;
 * pcm receives interleaved int32.  Returns number of frames decoded or <0:
 * -1 bad sync/header, -2 header CRC-8, -3 frame CRC-16, -4 capacity, -5 reserved/unsupported. */
long fo_decode_frames(const uint8_t *in, size_t len, uint32_t channels, uint32_t bps,
	int32_t *pcm, uint64_t cap_samples, uint64_t *nsamples_out, uint32_t *frame_offsets, uint32_t max_frames);

void fo_md5_pcm(const int32_t *pcm, uint64_t nsamples, uint32_t channels, uint32_t bps, uint8_t digest[16]);
uint8_t fo_crc8(const uint8_t *d, size_t n);
uint16_t fo_crc16(const uint8_t *d, size_t n);

/* STREAMINFO + "fLaC" marker: writes the 42-byte stream head a decoder needs
 * (format.h STREAMINFO layout; stream_encoder_framing.c:50-110). is_last sets the
 * last-metadata-block flag. */
void fo_write_stream_head(uint8_t out[42], uint32_t min_blocksize, uint32_t max_blocksize,
	uint32_t min_framesize, uint32_t max_framesize, uint32_t sample_rate, uint32_t channels,
	uint32_t bps, uint64_t total_samples, const uint8_t md5[16], int is_last);

#ifdef __cplusplus
}
#endif
#endif
