/* C ABI of the MI355X (gfx950) FLAC frame engine.
 *
 * This is the drop-in boundary: a batch of whole, independent frames goes in, packed
 * FLAC frames and their sizes come out.  It replaces, one level above the reference's
 * per-call function pointers (stream_encoder.c:378-386), the body
 *     process_frame_ -> process_subframes_ -> process_subframe_ -> evaluate_*_subframe_
 *     -> find_best_partition_order_ -> FLAC__frame_add_header / add_subframe_ -> CRC-16
 * (stream_encoder.c:2866-3167, :3169-3402, :3475-3766; stream_encoder_framing.c:219-553)
 * for encode, and read_frame_ -> read_subframe_* -> read_residual_partitioned_rice_ ->
 * restore_signal -> channel decorrelation (stream_decoder.c:2002-2145, :2456-2777) for
 * decode.  The host-side libFLAC-style object (flac_amd/csrc/stream_encoder_shim.c)
 * queues blocks, calls these entry points and replays write_frame_ in stream order.
 *
 * Plain C: pointers and sizes only, no HIP or torch types.  Device pointers are passed
 * as void* / const void*; streams as void* (a hipStream_t, or NULL for the default stream).
 * Every function returns 0 on success or a negative FLACAMD_E_* code; nothing falls back
 * to a CPU path.
 */
#ifndef FLAC_AMD_ENGINE_H
#define FLAC_AMD_ENGINE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FLACAMD_MAX_CHANNELS 8
#define FLACAMD_MAX_LPC_ORDER 32
#define FLACAMD_MAX_APODIZATIONS 32
#define FLACAMD_MAX_BLOCKSIZE 65535      /* the format's limit (format.h:108); blocks above 16384 samples take a slower path that keeps the signal in device memory instead of LDS */
#define FLACAMD_MAX_PARTITION_ORDER 8    /* engine limit; 8 is also the subset limit */

/* error codes */
#define FLACAMD_OK 0
#define FLACAMD_E_INVALID_ARGUMENT (-1)
#define FLACAMD_E_UNSUPPORTED (-2)       /* legal for the reference, outside the engine limits above */
#define FLACAMD_E_NO_DEVICE (-3)
#define FLACAMD_E_OUT_OF_MEMORY (-4)
#define FLACAMD_E_DEVICE (-5)            /* a HIP call failed; see flacamd_last_error() */
#define FLACAMD_E_CAPACITY (-6)          /* output buffer / batch capacity too small */
#define FLACAMD_E_FRAME_OVERFLOW (-7)    /* a frame exceeded its slot (pathological residuals) */
#define FLACAMD_E_BAD_STREAM (-8)        /* decoder: sync / CRC / reserved-field failure */

/* Window kinds in the order of the reference's set_apodization parser
 * (stream_encoder.c:1546-1585). */
enum {
	FLACAMD_WIN_BARTLETT = 0, FLACAMD_WIN_BARTLETT_HANN, FLACAMD_WIN_BLACKMAN, FLACAMD_WIN_BLACKMAN_HARRIS_4TERM,
	FLACAMD_WIN_CONNES, FLACAMD_WIN_FLATTOP, FLACAMD_WIN_GAUSS, FLACAMD_WIN_HAMMING, FLACAMD_WIN_HANN,
	FLACAMD_WIN_KAISER_BESSEL, FLACAMD_WIN_NUTTALL, FLACAMD_WIN_RECTANGLE, FLACAMD_WIN_TRIANGLE, FLACAMD_WIN_TUKEY,
	FLACAMD_WIN_WELCH
};

/* PCM layouts accepted by the encoder / produced by the decoder (always interleaved by
 * channel, one "wide sample" = one sample of every channel). */
enum {
	FLACAMD_PCM_S32 = 0,  /* int32 per sample, right-justified: what process_interleaved takes (stream_encoder.h:1760) */
	FLACAMD_PCM_S16 = 1,  /* int16 per sample (streams with bits_per_sample <= 16) */
	FLACAMD_PCM_S32_FRAMES = 2  /* decoder output only: int32 per sample, frame after frame, each frame as channel 0's blocksize samples,
	                             * then channel 1's, ...: the buffers a FLAC__StreamDecoderWriteCallback is handed
	                             * (stream_decoder.h:509-511, output[channel][sample] of stream_decoder.c:2963) with no host-side copy */
};

/* Page-locked host memory for the buffers of the *_host entry points: copies to and from it run at the speed of the
 * bus (pageable memory goes through a staging copy at a fraction of it).  Freed buffers are kept for the next caller
 * (flacamd_pipe_trim gives them back).  No counterpart in the reference: its data never leaves the host. */
void *flacamd_host_alloc(size_t bytes);
void flacamd_host_free(void *p, size_t bytes);

/* The analysis settings of one stream: the fields of FLAC__StreamEncoderProtected that the
 * frame path reads (include/protected/stream_encoder.h:78-108). Zero means "default" for
 * blocksize and qlp_coeff_precision exactly as in init_stream_internal_
 * (stream_encoder.c:686-735). */
typedef struct {
	uint32_t channels, bits_per_sample, sample_rate, blocksize;
	uint32_t do_mid_side, loose_mid_side;
	uint32_t max_lpc_order, qlp_coeff_precision, do_qlp_coeff_prec_search;
	uint32_t do_exhaustive_model_search;
	uint32_t min_residual_partition_order, max_residual_partition_order;
	uint32_t disable_constant, disable_fixed, disable_verbatim; /* stream_encoder.c:1747-1778 */
	uint32_t num_apodizations;
	uint32_t apod_type[FLACAMD_MAX_APODIZATIONS];
	float    apod_param[FLACAMD_MAX_APODIZATIONS]; /* tukey p / gauss stddev */
} flacamd_params;

typedef struct flacamd_encoder flacamd_encoder;
typedef struct flacamd_decoder flacamd_decoder;

/* Fill the analysis fields for compression level 0..8 (compression_levels_[],
 * stream_encoder.c:127-148; FLAC__stream_encoder_set_compression_level :1455-1495). */
void flacamd_params_preset(flacamd_params *p, unsigned level);
/* Resolve defaults and validate like init_stream_internal_ (stream_encoder.c:667-766).
 * Returns 0, FLACAMD_E_INVALID_ARGUMENT, or FLACAMD_E_UNSUPPORTED. */
int flacamd_params_resolve(flacamd_params *p);
/* One apodization window of length L into w[L] (window.c:49-223), computed on the host. */
void flacamd_window(uint32_t type, float param, int32_t L, float *w);

/* Create an encoder bound to HIP device `device` able to take up to max_frames frames per
 * batch.  Allocates all device workspace once (replaces resize_buffers_, stream_encoder.c:2250). */
int flacamd_encoder_create(flacamd_encoder **out, int device, const flacamd_params *params, uint32_t max_frames);
void flacamd_encoder_destroy(flacamd_encoder *e);
/* Resolved parameters (blocksize, precision filled in). */
const flacamd_params *flacamd_encoder_params(const flacamd_encoder *e);
/* Upper bound of the bytes one batch of `nframes` frames can produce. */
size_t flacamd_encoder_max_output_bytes(const flacamd_encoder *e, uint32_t nframes);

/* Encode a run of consecutive blocks of ONE stream, PCM already in device memory.
 *   d_pcm          nsamples wide samples, interleaved, layout pcm_format
 *   nsamples       all blocks are full (blocksize samples) except that, when `final_block`
 *                  is nonzero, the last one may be short: it is then coded as the
 *                  reference's finish() codes it (stream_encoder.c:1322-1327: partition
 *                  order 0, head of the full-length window).  Without final_block,
 *                  nsamples must be a multiple of blocksize.
 *   first_frame_number  number of the first frame of the batch (frame header, :2952)
 *   prev_assignment     channel assignment of the frame before the batch, only read for
 *                  loose mid/side streams whose batch does not start a decision period
 *                  (stream_encoder.c:2958-2967); pass 0 otherwise
 *   d_out / out_cap     device buffer receiving the frames back to back
 *   d_frame_bytes  device array [nframes]: size of each frame
 *   h_total_bytes  host: receives the total (this call synchronises the stream once at
 *                  its end to read it and the status word); may be NULL
 * Returns 0 or a negative error; on FLACAMD_E_FRAME_OVERFLOW no output is valid.
 * flacamd_encoder_max_output_bytes() covers every frame whose size stays within the verbatim bound.  The reference
 * can write larger ones (a subframe is kept on its estimate, and its narrow partition sums wrap on full-scale noise of
 * wide samples, stream_encoder.c:3808); such a batch is coded in a second pass and, when out_cap is too small for it,
 * the call returns FLACAMD_E_CAPACITY with the bytes it takes in *h_total_bytes: call again with a buffer of that size. */
int flacamd_encode_device(flacamd_encoder *e, const void *d_pcm, int pcm_format, uint64_t nsamples,
	int final_block, uint32_t first_frame_number, uint32_t prev_assignment,
	void *d_out, size_t out_cap, uint32_t *d_frame_bytes, uint64_t *h_total_bytes, void *stream);

/* flacamd_encode_device without its synchronisation: the batch is queued on `stream` and the call returns while the
 * device works, so a caller that codes batch after batch (process_frame_ in a loop, stream_encoder.c:2866-2923) keeps
 * the device busy across batch borders.  Up to FLACAMD_QUEUE_DEPTH batches of one encoder may be outstanding, all on
 * the same stream (they share the encoder's work arrays and run in order); each needs its own d_out / d_frame_bytes
 * until it has been waited for.  A loose mid/side stream's batches must be whole decision periods (prev_assignment
 * of a queued batch cannot come from one still running).
 * flacamd_encoder_wait waits for every outstanding batch, stores their total sizes in h_total_bytes[0..] (in queue
 * order, at most max_totals) and their number in *n_batches (either may be NULL), and returns the first failure among
 * them: FLACAMD_E_FRAME_OVERFLOW (a frame beyond the verbatim bound: code that batch again with flacamd_encode_device,
 * which runs the second pass) or FLACAMD_E_CAPACITY (that total is the size its output buffer needs). */
#define FLACAMD_QUEUE_DEPTH 16u
int flacamd_encode_device_queue(flacamd_encoder *e, const void *d_pcm, int pcm_format, uint64_t nsamples,
	int final_block, uint32_t first_frame_number, uint32_t prev_assignment,
	void *d_out, size_t out_cap, uint32_t *d_frame_bytes, void *stream);
int flacamd_encoder_wait(flacamd_encoder *e, uint64_t *h_total_bytes, uint32_t max_totals, uint32_t *n_batches);

/* Several complete, independent streams of the encoder's format in ONE pass over the device (many short
 * clips: one call instead of one per clip).  Stream i is h_len[i] wide samples starting at wide sample
 * h_start[i] of d_pcm; the streams tile the buffer in whole blocks: h_start[0] = 0 and h_start[i + 1] =
 * h_start[i] + ceil(h_len[i] / blocksize) * blocksize (what lies in the padding is never read as audio).
 * Every stream is coded as the reference codes a whole stream: frame numbers from 0, loose mid/side
 * periods from its first frame, a short last block as a final block (stream_encoder.c:1322-1327).  The
 * frames of all streams come out back to back in stream order, d_frame_bytes gets one size per frame
 * (stream i owns ceil(h_len[i] / blocksize) consecutive entries).  No counterpart in the reference, which
 * codes one stream per encoder object. */
int flacamd_encode_streams_device(flacamd_encoder *e, const void *d_pcm, int pcm_format, uint32_t nstreams,
	const uint64_t *h_start, const uint64_t *h_len, void *d_out, size_t out_cap, uint32_t *d_frame_bytes,
	uint64_t *h_total_bytes, void *stream);

/* A batch of clips as COMPLETE streams in one pass: like flacamd_encode_streams_device, but every clip comes out as the
 * file the reference's encoder leaves behind for it when given a seekable sink and no client metadata - "fLaC",
 * STREAMINFO with frame size extremes, sample count and the MD5 of its samples (stream_encoder.c:1034-1106, :2521-2620;
 * md5.c:394-420, one GPU lane per clip beside the coding stages), the VORBIS_COMMENT carrying the vendor string, then
 * its frames.  The streams lie back to back in d_out; h_stream_off[nclips + 1] receives their byte offsets.  With
 * FLACAMD_E_CAPACITY, *h_total_bytes is the output size the call needs.  No counterpart in the reference (one encoder
 * object per stream there); replaces the per-file loop of the flac tool (src/flac/main.c:327, encode.c:966) for clips
 * already in device memory. */
int flacamd_encode_clips_device(flacamd_encoder *e, const void *d_pcm, int pcm_format, uint32_t nclips,
	const uint64_t *h_start, const uint64_t *h_len, void *d_out, size_t out_cap, uint64_t *h_stream_off, uint64_t *h_total_bytes, void *stream);

/* The digests of several clip batches in ONE pass: every clip of every group is a lane of the same kernel launch, so the
 * whole hash takes as long as the longest clip (a digest is a serial chain, md5.c:216-420) instead of that once per
 * group.  d_digests of group g receives nclips x 16 bytes (device memory).  Asynchronous on `stream`; the tables are
 * staged in buffers that belong to the device, and a call first waits until the previous call's work on that device
 * has passed its launch (an event recorded behind it), so calls may follow one another on any streams.  Hand the result to the next
 * flacamd_encode_clips_device call of a group's encoder with flacamd_encoder_use_digests. */
typedef struct {
	const void *d_pcm; int pcm_format; uint32_t channels, bits_per_sample, nclips;
	const uint64_t *h_start, *h_len;   /* as for flacamd_encode_clips_device */
	uint8_t *d_digests;
} flacamd_md5_group;
int flacamd_md5_groups_device(int device, uint32_t ngroups, const flacamd_md5_group *groups, void *stream);
/* The next flacamd_encode_clips_device call of this encoder takes its clips' digests from d_digests instead of computing
 * them; ready_event (a hipEvent_t, may be null) is waited for on the call's stream before the stream heads are written. */
int flacamd_encoder_use_digests(flacamd_encoder *e, const uint8_t *d_digests, void *ready_event);

/* Same with host buffers: copies PCM in, frames and sizes out (PCIe inclusive). */
int flacamd_encode_host(flacamd_encoder *e, const void *h_pcm, int pcm_format, uint64_t nsamples,
	int final_block, uint32_t first_frame_number, uint32_t prev_assignment,
	uint8_t *h_out, size_t out_cap, uint32_t *h_frame_bytes, uint64_t *h_total_bytes, uint32_t *h_last_assignment);

/* ---- pipelined encoding from host memory, over one or several GPUs ------------------------------
 * Replaces, for a client whose PCM arrives in host memory, the loop "process_frame_ per block as soon as
 * blocksize+1 samples are buffered" (stream_encoder.c:1973-2105, :2866-2923) and the hand-over of each frame
 * (write_bitbuffer_ :2384-2429): batches of `batch_frames` blocks are kept in flight so that the upload of one
 * batch, the kernels of the one before and the download of the one before that overlap, and successive batches
 * go to the listed devices in turn (frame ranges sharded over the GPUs of a node; the host puts them back in
 * order).  For loose mid/side streams batch_frames is rounded up to whole decision periods (:2958-2967) so that
 * no batch depends on the one before.  `depth` = batches in flight per device (2..8).
 *
 * Use: p = fill_buffer() (pinned host memory for one batch, layout pcm_format(): int16 for streams of at most
 * 16 bits, else int32; NULL while every slot is in flight) -> write up to batch_frames * blocksize wide samples
 * -> submit(nsamples, final_block, first_frame_number) -> ... -> collect() (waits for the oldest batch; the
 * pointers stay valid until release()) -> release().  Batches come back in submission order. */
typedef struct flacamd_pipe flacamd_pipe;
int flacamd_pipe_create(flacamd_pipe **out, const int *devices, uint32_t ndevices, const flacamd_params *params, uint32_t batch_frames, uint32_t depth);
void flacamd_pipe_destroy(flacamd_pipe *p);
/* The pinned buffers of destroyed pipes are kept (at most 1 GiB) for the next pipe of the same geometry, because
 * fresh pinned memory is slow to touch for the first time; this releases them. */
void flacamd_pipe_trim(void);
int flacamd_pipe_pcm_format(const flacamd_pipe *p);
uint32_t flacamd_pipe_batch_frames(const flacamd_pipe *p);
uint32_t flacamd_pipe_slots(const flacamd_pipe *p);          /* ndevices * depth */
uint32_t flacamd_pipe_in_flight(const flacamd_pipe *p);      /* submitted and not yet released */
size_t flacamd_pipe_max_batch_bytes(const flacamd_pipe *p);  /* size bound of one batch's frames (see flacamd_encoder_max_output_bytes) */
const flacamd_params *flacamd_pipe_params(const flacamd_pipe *p);
void *flacamd_pipe_fill_buffer(flacamd_pipe *p);
void *flacamd_pipe_slot_buffer(flacamd_pipe *p, uint32_t slot); /* batch q (counted from 0) is filled in slot q % slots */
int flacamd_pipe_submit(flacamd_pipe *p, uint64_t nsamples, int final_block, uint32_t first_frame_number);
int flacamd_pipe_ready(flacamd_pipe *p);                     /* 1: collect() would not wait, 0: it would */
int flacamd_pipe_collect(flacamd_pipe *p, const uint8_t **h_frames, const uint32_t **h_frame_bytes, uint32_t *nframes,
	uint64_t *total_bytes, const void **h_pcm, uint64_t *nsamples, uint32_t *first_frame_number);
int flacamd_pipe_release(flacamd_pipe *p);
/* A whole run of blocks from ordinary host memory through the pipe in one call (copies into the pinned buffers,
 * narrowing int32 to int16 where the stream allows): frames back to back in h_out, one size per frame. */
int flacamd_pipe_encode_host(flacamd_pipe *p, const void *h_pcm, int pcm_format, uint64_t nsamples, int final_block, uint32_t first_frame_number,
	uint8_t *h_out, size_t out_cap, uint32_t *h_frame_bytes, size_t frame_cap, uint64_t *h_total_bytes, uint32_t *h_nframes);
/* int32 samples of a stream of at most 16 bits -> int16 (what format_input does for the CLI, src/flac/encode.c:2338, the other way round) */
void flacamd_pcm_narrow(int16_t *dst, const int32_t *src, size_t nvalues);

/* White-box views of the last batch for parity tests (device -> host copies):
 *   what = 0: float autoc[num_apod][max_lpc_order+1][nsignals]   (lpc.c:63-110 outputs)
 *   what = 1: uint64 fixed_err[5][nsignals]                       (fixed.c:234-243 sums)
 *   what = 2: uint32 winner header words [nsignals][4]            (see engine_types.h)
 *   what = 3: uint32 estimated bits of the winning subframe [nsignals]
 *   what = 4: uint8 wasted bits [nsignals]
 * nsignals = frames x (4 for mid/side stereo, else channels). Returns bytes written or <0. */
long flacamd_encoder_debug_read(flacamd_encoder *e, int what, void *h_dst, size_t cap);

/* The reference's SPOTCHECK_ESTIMATE build (stream_encoder.c:3444-3473: every candidate's estimated size against the
 * bits it really takes) as a switch: an encoder made while the environment has FLACAMD_DEBUG bit 22 (0x400000) set
 * compares, for every subframe it writes (mono / stereo streams with blocks up to 4096 samples), the winner's estimate
 * with the bits written and adds up: out[0] subframes, out[1] subframes whose estimate is not exact, out[2] sum of
 * |estimate - bits|, out[3] the largest difference, out[4] bits written.  reset != 0 clears the totals afterwards.
 * FLACAMD_E_UNSUPPORTED when the encoder was made without the switch. */
int flacamd_encoder_spotcheck(flacamd_encoder *e, uint64_t out[5], int reset);

/* White-box view for parity tests: the two logarithm expressions of the model stage (fixed.c:256-268: bits expected of a
 * fixed predictor from its |residual| sum e over len samples; lpc.c:1336-1351: bits expected of an LPC order from its
 * prediction error and scale), evaluated on the device for n inputs of the caller's choosing. */
int flacamd_debug_model_logs(int device, uint32_t n, const uint64_t *h_e, const double *h_len, const double *h_err, const double *h_scale, float *h_fb, double *h_eb);

/* Kernel time of the stages of the last batch in milliseconds, measured with HIP events on
 * the stream the kernels ran on: ms[0..5] = prep (the autocorrelation runs beside it on a second stream), what was left
 * of the autocorrelation to wait for when prep had ended, model, search, pack, gather.
 * Valid only when the batch was submitted with timing enabled. */
void flacamd_encoder_set_timing(flacamd_encoder *e, int enabled);
int flacamd_encoder_last_timing(flacamd_encoder *e, float ms[6]);

/* MD5 of the unencoded PCM as FLAC__MD5Accumulate sees it (md5.c:273-420): one digest per
 * independent stream of a batch, computed on the device, one lane per stream.
 * d_pcm holds the streams back to back; h_offsets[nstreams+1] are wide-sample offsets. */
int flacamd_md5_device(int device, const void *d_pcm, int pcm_format, uint32_t channels, uint32_t bits_per_sample,
	const uint64_t *h_offsets, uint32_t nstreams, uint8_t *h_digests /* [nstreams][16] */, void *stream);

/* flacamd_md5_device without any allocation.  No counterpart in the reference: it is for callers that
 * cannot allocate device memory - tight serving loops, harnesses that forbid hipMalloc after start-up.  Same
 * arguments and results as flacamd_md5_device; the stream offsets and the digests are staged in a fixed array
 * inside the library (one copy per GPU, about 1.5 MiB, held until the process ends), so nstreams is limited to
 * flacamd_md5_device_static_max_streams() = 65536 (FLACAMD_E_CAPACITY beyond that) and the call makes no
 * hipMalloc / hipFree.  Like flacamd_md5_device it copies from and to the caller's host buffers and waits for
 * `stream` before it returns, so it cannot be recorded into a stream capture.  All calls on one GPU share that
 * array; the library serialises them (a call waits while another one on the same GPU is between its upload and the
 * end of its wait).  bits_per_sample outside 1..32: FLACAMD_E_INVALID_ARGUMENT (both entry points). */
int flacamd_md5_device_static(int device, const void *d_pcm, int pcm_format, uint32_t channels, uint32_t bits_per_sample,
	const uint64_t *h_offsets, uint32_t nstreams, uint8_t *h_digests /* [nstreams][16] */, void *stream);
uint32_t flacamd_md5_device_static_max_streams(void);

/* ---- decoder ---------------------------------------------------------------------- */

/* Decoder for streams of known channels / bits_per_sample / max blocksize (from STREAMINFO). */
int flacamd_decoder_create(flacamd_decoder **out, int device, uint32_t channels, uint32_t bits_per_sample,
	uint32_t max_blocksize, uint32_t max_frames, size_t max_bytes);
void flacamd_decoder_destroy(flacamd_decoder *d);

/* Decode a byte range of frames held in device memory (no metadata; d_bytes 16-byte aligned): finds the frame
 * boundaries (sync code + header CRC-8 + frame CRC-16, stream_decoder.c:1947-2145),
 * decodes every frame in parallel and writes interleaved PCM (pcm_format) to d_pcm in
 * stream order.  h_info[0]=frames, [1]=wide samples, [2]=first frame number,
 * [3]=bytes consumed. */
int flacamd_decode_device(flacamd_decoder *d, const void *d_bytes, size_t nbytes,
	void *d_pcm, int pcm_format, uint64_t pcm_cap_samples, uint64_t h_info[4], void *stream);
int flacamd_decode_host(flacamd_decoder *d, const uint8_t *h_bytes, size_t nbytes,
	void *h_pcm, int pcm_format, uint64_t pcm_cap_samples, uint64_t h_info[4]);

/* Per frame of the last successful decode, in stream order (what read_frame_header_ leaves in
 * FLAC__FrameHeader, stream_decoder.c:2147-2454): first wide sample within the batch, blocksize, the
 * frame/sample number coded in the header, channel assignment (0 independent, 1 left/side,
 * 2 right/side, 3 mid/side) and the byte range of the frame within the input.  Any pointer may be NULL. */
int flacamd_decoder_frame_table(flacamd_decoder *d, uint32_t nframes, uint64_t *h_sample_off, uint32_t *h_blocksize,
	uint64_t *h_number, uint32_t *h_assignment, uint64_t *h_byte_pos, uint64_t *h_byte_end);

/* Damaged input.  By default a byte range whose CRC-valid frames do not follow one another exactly is
 * refused (FLACAMD_E_BAD_STREAM).  With tolerant mode on, the decode calls return FLACAMD_OK for such a
 * range too: every CRC-valid frame is written to the PCM output in order of position, h_info counts
 * those, and flacamd_decoder_damage() is nonzero (bit 0: frames do not tile the range, bit 1: no frame
 * at all).  The caller then replays the reference's serial walk (frame_sync_ stream_decoder.c:1947-2000,
 * read_frame_ :2002-2145) over the candidate table to decide which frames to keep, where zeroed frames
 * and error callbacks go. */
int flacamd_decoder_set_tolerant(flacamd_decoder *d, int on);
uint32_t flacamd_decoder_damage(const flacamd_decoder *d);

/* Every position of the last decode that passed the frame header checks (sync code, legal fields, CRC-8),
 * in order of position: byte offset, the offset where parsing of the frame stopped (just past its CRC-16
 * when it parsed completely, else where the reference resumes its search for a sync code), blocksize,
 * coded frame/sample number, first wide sample in the PCM output (all ones if the frame was not written)
 * and info: bit 0 parsed to the end, bit 1 CRC-16 correct, bits 4..7 channel assignment, bits 8..11 why
 * parsing stopped early (1 = the cases the reference reports as LOST_SYNC: stream_decoder.c:2481, :2628,
 * :2732-2745, :2785; 4 = UNPARSEABLE_STREAM: :2494-2508, :2580, :2659; 5 = a field the reference does not
 * define; 6 = the frame runs past the end of the buffer).  *h_count receives the number of entries;
 * FLACAMD_E_CAPACITY if it exceeds max_entries. */
int flacamd_decoder_candidate_table(flacamd_decoder *d, uint32_t max_entries, uint32_t *h_count, uint64_t *h_byte_pos, uint64_t *h_byte_end,
	uint32_t *h_info, uint32_t *h_blocksize, uint64_t *h_number, uint64_t *h_sample_off);

/* What the decoder read of one subframe: the fields the reference leaves in FLAC__Subframe for the client's write
 * callback (include/FLAC/format.h:283-352, filled by read_subframe_* stream_decoder.c:2456-2719), for analysis clients
 * (src/flac/analyze.c:61-125).  Recorded per candidate frame and channel when flacamd_decoder_set_subframe_info is on.
 *   head: [1:0] type (0 constant, 1 verbatim, 2 fixed, 3 lpc)  [7:2] order  [12:8] wasted bits
 *         [17:13] coefficient precision (lpc)  [18] residual coding method (0 RICE, 1 RICE2)  [22:19] partition order
 *         [27:23] quantisation level as coded (5-bit two's complement)
 *         [28] the prediction was summed in 64 bits (stream_decoder.c:2682-2692)
 *   value: the sample of a constant subframe (before the wasted-bits shift)
 *   rice[i]: parameter of partition i, or 0x80 | raw_bits for an escaped partition; the first FLACAMD_SUBINFO_PARTS
 *         partitions are recorded (partition orders up to 8, the streamable subset's limit) */
#define FLACAMD_SUBINFO_PARTS 256u
typedef struct {
	uint32_t head; int32_t value; uint32_t reserved[2];
	int32_t warmup[FLACAMD_MAX_LPC_ORDER];
	int16_t q[FLACAMD_MAX_LPC_ORDER];
	uint8_t rice[FLACAMD_SUBINFO_PARTS];
} flacamd_subframe_info;
int flacamd_decoder_set_subframe_info(flacamd_decoder *d, int on);
/* The records of the last decode, by candidate (as in flacamd_decoder_candidate_table): h_records[candidate * channels
 * + channel].  h_frame_candidate (may be NULL) receives, per accepted frame in stream order, its candidate index. */
int flacamd_decoder_subframe_info(flacamd_decoder *d, uint32_t max_candidates, flacamd_subframe_info *h_records, uint32_t *h_frame_candidate, uint32_t max_frames);

/* The residuals the stream carried, for the frames of the last decode (subframe info must be on): what the reference
 * hands a client as FLAC__Subframe_Fixed::residual / FLAC__Subframe_LPC::residual and, for a verbatim subframe, as
 * FLAC__Subframe_Verbatim::data (include/FLAC/format.h:283-352; read_residual_partitioned_rice_ stream_decoder.c:2721-2777).
 * Layout of FLACAMD_PCM_S32_FRAMES: frame after frame, within a frame channel after channel, `blocksize` entries per
 * channel of which the first blocksize - order are the subframe's residuals (rows of constant subframes are not written).
 * One pass over the restored signal, parallel in the sample index.  The device variant is asynchronous on `stream`
 * (the one the decode ran on, or one ordered after it). */
int flacamd_decoder_residuals_device(flacamd_decoder *d, int32_t *d_out, uint64_t cap_samples, void *stream);
int flacamd_decoder_residuals_host(flacamd_decoder *d, int32_t *h_out, uint64_t cap_samples);

/* "fLaC" + STREAMINFO + the empty VORBIS_COMMENT with the reference's vendor string, as the reference
 * leaves them in a finished file (stream_encoder.c:1034-1106 written at init, :2521-2620 patched at
 * finish): for callers that assemble streams from engine output themselves.  Returns the byte count. */
#define FLACAMD_STREAM_HEAD_BYTES 86
size_t flacamd_write_stream_head(uint8_t out[FLACAMD_STREAM_HEAD_BYTES], uint32_t blocksize, uint32_t min_framesize, uint32_t max_framesize,
	uint32_t sample_rate, uint32_t channels, uint32_t bits_per_sample, uint64_t total_samples, const uint8_t md5[16]);

const char *flacamd_strerror(int code);
/* Text of the last failing HIP call on this thread ("" if none). */
const char *flacamd_last_error(void);
/* Number of HIP devices visible, or 0. Does not initialise a context. */
int flacamd_device_count(void);

#ifdef __cplusplus
}
#endif
#endif
