/* Host side of the libFLAC-style encoder object (include/flac_amd_stream.h) in plain C.
 *
 * Keeps the reference's layer-3 semantics - settings only before init, the init checks
 * and their status codes, the "fLaC" / STREAMINFO / VORBIS_COMMENT stream head, callbacks
 * in stream order, min/max frame size, MD5, STREAMINFO back-patching at finish - and hands
 * the layer-2/1 work (process_frame_ and below) to the HIP engine in batches of frames.
 * Citations are to /root/reference/src/libFLAC/stream_encoder.c unless noted.
 */
#define _FILE_OFFSET_BITS 64
#include "../../include/flac_amd_stream.h"
#include "../../include/flac_amd_engine.h"
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <sys/types.h>

#define VENDOR_STRING FLAC__VENDOR_STRING /* format_tables.c */

const char * const FLAC__StreamEncoderStateString[] = { /* :458-468 */
	"FLAC__STREAM_ENCODER_OK", "FLAC__STREAM_ENCODER_UNINITIALIZED", "FLAC__STREAM_ENCODER_OGG_ERROR",
	"FLAC__STREAM_ENCODER_VERIFY_DECODER_ERROR", "FLAC__STREAM_ENCODER_VERIFY_MISMATCH_IN_AUDIO_DATA",
	"FLAC__STREAM_ENCODER_CLIENT_ERROR", "FLAC__STREAM_ENCODER_IO_ERROR", "FLAC__STREAM_ENCODER_FRAMING_ERROR",
	"FLAC__STREAM_ENCODER_MEMORY_ALLOCATION_ERROR" };
const char * const FLAC__StreamEncoderInitStatusString[] = { /* :470-485 */
	"FLAC__STREAM_ENCODER_INIT_STATUS_OK", "FLAC__STREAM_ENCODER_INIT_STATUS_ENCODER_ERROR",
	"FLAC__STREAM_ENCODER_INIT_STATUS_UNSUPPORTED_CONTAINER", "FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_CALLBACKS",
	"FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_NUMBER_OF_CHANNELS", "FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_BITS_PER_SAMPLE",
	"FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_SAMPLE_RATE", "FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_BLOCK_SIZE",
	"FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_MAX_LPC_ORDER", "FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_QLP_COEFF_PRECISION",
	"FLAC__STREAM_ENCODER_INIT_STATUS_BLOCK_SIZE_TOO_SMALL_FOR_LPC_ORDER", "FLAC__STREAM_ENCODER_INIT_STATUS_NOT_STREAMABLE",
	"FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_METADATA", "FLAC__STREAM_ENCODER_INIT_STATUS_ALREADY_INITIALIZED" };

#include "host_md5.h"
#include "metadata_blocks.h"

/* a hint to the core that this is a spin-wait; a plain compiler barrier where the architecture has no such hint */
#if defined(__x86_64__) || defined(__i386__)
#define cpu_relax() __builtin_ia32_pause()
#else
#define cpu_relax() __asm__ __volatile__("" ::: "memory")
#endif

/* ---- the object */
struct FLAC__StreamEncoder {
	FLAC__StreamEncoderState state;
	flacamd_params p;                 /* the protected_ settings (include/protected/stream_encoder.h:78-108) */
	FLAC__bool verify, streamable_subset, do_md5, do_escape_coding;
	unsigned rice_parameter_search_dist;
	FLAC__uint64 total_samples_estimate;
	int devices[64]; unsigned ndevices, batch_frames, depth;   /* GPUs the batches go to in turn, frames per batch, batches in flight per GPU */
	/* callbacks */
	FLAC__StreamEncoderWriteCallback write_cb; FLAC__StreamEncoderSeekCallback seek_cb; FLAC__StreamEncoderTellCallback tell_cb;
	FLAC__StreamEncoderMetadataCallback metadata_cb; FLAC__StreamEncoderProgressCallback progress_cb; void *client_data;
	FILE *file;
	/* stream bookkeeping (:2423-2426, :2505-2511) */
	FLAC__StreamMetadata streaminfo;
	FLAC__uint64 streaminfo_offset, audio_offset, bytes_written, samples_written;
	unsigned frames_written, total_frames_estimate, current_frame_number;
	md5_state md5;
	/* The batches.  Samples go straight from the client's buffer into the pinned upload buffer of the batch being
	 * filled (as int16 for streams of at most 16 bits); full batches are in flight on the GPUs while the next one
	 * fills, and come back in order (flacamd_pipe_*).  There is no second host copy and no CPU coding path. */
	flacamd_pipe *pipe;
	int fmt16;                        /* the pipe's PCM layout is int16 */
	uint8_t *fill; size_t filled;     /* batch being filled: its buffer, wide samples in it */
	size_t batch_samples;             /* wide samples of a full batch */
	unsigned next_frame_number;       /* number of the first frame of the batch being filled */
	unsigned nslots; uint8_t **slot_buf; /* the pipe's upload buffers in the order they are filled */
	uint64_t batches_started;         /* batches whose filling has begun: batch q fills slot q % nslots */
	size_t vdec_cap;
	/* MD5 of the samples (md5.c:394-420) on a thread of its own: one stream is one serial chain, about 0.6 GB/s on a
	 * host core, far below what the GPU codes; it reads the samples where the client thread put them */
	pthread_t md5_thread; int md5_running, md5_quit;
	pthread_mutex_t md5_lock; pthread_cond_t md5_data, md5_space;
	uint64_t md5_produced, md5_hashed; /* wide samples written into upload buffers / hashed so far */
	uint8_t *md5_scratch;
	/* helpers for large process_interleaved calls: the copy out of the client's buffer (narrowing to int16) is what
	 * bounds a host-fed stream once everything else overlaps; one thread moves about 4 G samples/s, the upload 14 */
	unsigned fill_threads;            /* threads a large call is spread over, the caller included (default 4) */
	struct fill_pool *pool;
	/* client metadata blocks (:1711-1741: the array is ours, the blocks are the client's) and seek table tracking (:2458-2485) */
	FLAC__StreamMetadata **metadata; unsigned num_metadata;
	FLAC__StreamMetadata_SeekTable *seek_table; unsigned first_seekpoint_to_check;
	FLAC__uint64 seektable_offset;
	/* verify mode (:984-1017): device decoder, its output, what the reference keeps in verify.error_stats */
	flacamd_decoder *vdec; void *vpcm;
	FLAC__StreamDecoderState vstate;
	struct verify_stats { FLAC__uint64 absolute_sample; unsigned frame_number, channel, sample; FLAC__int32 expected, got; } vstats;
	struct { int armed; FLAC__uint64 absolute_sample; unsigned channel; FLAC__int32 mask; } sample_fault;
	struct { int armed; unsigned frame_number, byte_index, mask; } frame_fault;
};

static void set_defaults(FLAC__StreamEncoder *e) /* set_defaults_, :2107-2160 */
{
	int devices[64]; const unsigned ndev = e->ndevices, batch = e->batch_frames, depth = e->depth, fillt = e->fill_threads;
	const struct verify_stats keep = e->vstats; /* the reference only clears these at init (:1012-1017) */
	memcpy(devices, e->devices, sizeof devices);
	free(e->metadata);
	memset(e, 0, sizeof *e);
	e->vstats = keep; e->vstate = FLAC__STREAM_DECODER_UNINITIALIZED; /* what the finished verify decoder reports (:1351) */
	memcpy(e->devices, devices, sizeof devices); e->ndevices = ndev ? ndev : 1u; e->batch_frames = batch ? batch : 2048u; e->depth = depth; e->fill_threads = fillt ? fillt : 4u;
	e->state = FLAC__STREAM_ENCODER_UNINITIALIZED;
	e->verify = 0; e->streamable_subset = 1; e->do_md5 = 1;
	e->p.channels = 2; e->p.bits_per_sample = 16; e->p.sample_rate = 44100; e->p.blocksize = 0;
	flacamd_params_preset(&e->p, 5);
}

FLAC__StreamEncoder *FLAC__stream_encoder_new(void)
{
	FLAC__StreamEncoder *e = (FLAC__StreamEncoder*)calloc(1, sizeof *e);
	if(e) set_defaults(e);
	return e;
}

static void md5_stop(FLAC__StreamEncoder *e);
static void pool_stop(FLAC__StreamEncoder *e);
static void *md5_worker(void *arg);

static void free_buffers(FLAC__StreamEncoder *e)
{
	md5_stop(e);
	pool_stop(e);
	if(e->pipe) { flacamd_pipe_destroy(e->pipe); e->pipe = 0; }
	if(e->vdec) { flacamd_decoder_destroy(e->vdec); e->vdec = 0; }
	free(e->vpcm); e->vpcm = 0;
	free(e->slot_buf); e->slot_buf = 0; free(e->md5_scratch); e->md5_scratch = 0;
	e->fill = 0; e->filled = 0;
}

void FLAC__stream_encoder_delete(FLAC__StreamEncoder *e)
{
	if(!e) return;
	/* :601-638: delete finishes without emitting the pending block */
	if(e->file && e->file != stdout) fclose(e->file);
	free_buffers(e);
	free(e->metadata);
	free(e);
}

#define SETTER(name, type, stmt) FLAC__bool FLAC__stream_encoder_##name(FLAC__StreamEncoder *e, type value) { if(!e || e->state != FLAC__STREAM_ENCODER_UNINITIALIZED) return 0; stmt; return 1; }
SETTER(set_streamable_subset, FLAC__bool, e->streamable_subset = value)
SETTER(set_channels, unsigned, e->p.channels = value)
SETTER(set_bits_per_sample, unsigned, e->p.bits_per_sample = value)
SETTER(set_sample_rate, unsigned, e->p.sample_rate = value)
SETTER(set_blocksize, unsigned, e->p.blocksize = value)
SETTER(set_do_mid_side_stereo, FLAC__bool, e->p.do_mid_side = value ? 1u : 0u)
SETTER(set_loose_mid_side_stereo, FLAC__bool, e->p.loose_mid_side = value ? 1u : 0u)
SETTER(set_max_lpc_order, unsigned, e->p.max_lpc_order = value)
SETTER(set_qlp_coeff_precision, unsigned, e->p.qlp_coeff_precision = value)
SETTER(set_do_qlp_coeff_prec_search, FLAC__bool, e->p.do_qlp_coeff_prec_search = value ? 1u : 0u)
SETTER(set_do_exhaustive_model_search, FLAC__bool, e->p.do_exhaustive_model_search = value ? 1u : 0u)
SETTER(set_min_residual_partition_order, unsigned, e->p.min_residual_partition_order = value)
SETTER(set_max_residual_partition_order, unsigned, e->p.max_residual_partition_order = value)
SETTER(set_total_samples_estimate, FLAC__uint64, e->total_samples_estimate = value)
SETTER(disable_constant_subframes, FLAC__bool, e->p.disable_constant = value ? 1u : 0u)
SETTER(disable_fixed_subframes, FLAC__bool, e->p.disable_fixed = value ? 1u : 0u)
SETTER(disable_verbatim_subframes, FLAC__bool, e->p.disable_verbatim = value ? 1u : 0u)
SETTER(set_do_md5, FLAC__bool, e->do_md5 = value)
/* :1635-1649 and :1684-1698: stored but never acted on ("not ready yet") */
SETTER(set_do_escape_coding, FLAC__bool, (void)value; e->do_escape_coding = 0)
SETTER(set_rice_parameter_search_dist, unsigned, (void)value; e->rice_parameter_search_dist = 0)

SETTER(set_verify, FLAC__bool, e->verify = value ? 1 : 0)

FLAC__bool FLAC__stream_encoder_set_compression_level(FLAC__StreamEncoder *e, unsigned value) /* :1455-1495 */
{
	if(!e || e->state != FLAC__STREAM_ENCODER_UNINITIALIZED) return 0;
	flacamd_params_preset(&e->p, value > 8 ? 8 : value);
	return 1;
}

FLAC__bool FLAC__stream_encoder_set_metadata(FLAC__StreamEncoder *e, FLAC__StreamMetadata **metadata, unsigned num_blocks)
{
	if(!e || e->state != FLAC__STREAM_ENCODER_UNINITIALIZED) return 0;
	if(!metadata) num_blocks = 0;
	free(e->metadata); e->metadata = 0; e->num_metadata = 0;
	if(num_blocks) {
		e->metadata = (FLAC__StreamMetadata**)malloc(sizeof(e->metadata[0]) * num_blocks);
		if(!e->metadata) return 0;
		memcpy(e->metadata, metadata, sizeof(e->metadata[0]) * num_blocks);
		e->num_metadata = num_blocks;
	}
	return 1;
}

FLAC__bool flacamd_stream_encoder_inject_verify_fault(FLAC__StreamEncoder *e, FLAC__uint64 absolute_sample, unsigned channel, FLAC__int32 mask)
{
	if(!e || e->state != FLAC__STREAM_ENCODER_UNINITIALIZED) return 0;
	e->sample_fault.armed = 1; e->sample_fault.absolute_sample = absolute_sample; e->sample_fault.channel = channel; e->sample_fault.mask = mask;
	return 1;
}
FLAC__bool flacamd_stream_encoder_inject_frame_fault(FLAC__StreamEncoder *e, unsigned frame_number, unsigned byte_index, unsigned mask)
{
	if(!e || e->state != FLAC__STREAM_ENCODER_UNINITIALIZED) return 0;
	e->frame_fault.armed = 1; e->frame_fault.frame_number = frame_number; e->frame_fault.byte_index = byte_index; e->frame_fault.mask = mask;
	return 1;
}

FLAC__bool flacamd_stream_encoder_set_device(FLAC__StreamEncoder *e, int device)
{
	if(!e || e->state != FLAC__STREAM_ENCODER_UNINITIALIZED || device < 0) return 0;
	e->devices[0] = device; e->ndevices = 1; return 1;
}
FLAC__bool flacamd_stream_encoder_set_devices(FLAC__StreamEncoder *e, unsigned num_devices, const int *device_ids)
{
	unsigned i;
	if(!e || e->state != FLAC__STREAM_ENCODER_UNINITIALIZED || num_devices == 0 || num_devices > 64 || !device_ids) return 0;
	for(i = 0; i < num_devices; i++) if(device_ids[i] < 0) return 0;
	memcpy(e->devices, device_ids, sizeof(int) * num_devices); e->ndevices = num_devices;
	return 1;
}
FLAC__bool flacamd_stream_encoder_set_batch_frames(FLAC__StreamEncoder *e, unsigned frames)
{
	if(!e || e->state != FLAC__STREAM_ENCODER_UNINITIALIZED || frames == 0) return 0;
	e->batch_frames = frames; return 1;
}
FLAC__bool flacamd_stream_encoder_set_fill_threads(FLAC__StreamEncoder *e, unsigned threads)
{
	if(!e || e->state != FLAC__STREAM_ENCODER_UNINITIALIZED || threads == 0 || threads > 8) return 0;
	e->fill_threads = threads; return 1;
}
FLAC__bool flacamd_stream_encoder_set_batches_in_flight(FLAC__StreamEncoder *e, unsigned per_device)
{
	if(!e || e->state != FLAC__STREAM_ENCODER_UNINITIALIZED || per_device < 2 || per_device > 8) return 0;
	e->depth = per_device; return 1;
}

/* :1531-1600: ';'-separated window names, tukey(p) / gauss(stddev) take a parameter,
 * unknown names are skipped, an empty result falls back to tukey(0.5) */
FLAC__bool FLAC__stream_encoder_set_apodization(FLAC__StreamEncoder *e, const char *spec)
{
	static const struct { const char *name; uint32_t type; } plain[] = {
		{"bartlett", FLACAMD_WIN_BARTLETT}, {"bartlett_hann", FLACAMD_WIN_BARTLETT_HANN}, {"blackman", FLACAMD_WIN_BLACKMAN},
		{"blackman_harris_4term_92db", FLACAMD_WIN_BLACKMAN_HARRIS_4TERM}, {"connes", FLACAMD_WIN_CONNES}, {"flattop", FLACAMD_WIN_FLATTOP},
		{"hamming", FLACAMD_WIN_HAMMING}, {"hann", FLACAMD_WIN_HANN}, {"kaiser_bessel", FLACAMD_WIN_KAISER_BESSEL}, {"nuttall", FLACAMD_WIN_NUTTALL},
		{"rectangle", FLACAMD_WIN_RECTANGLE}, {"triangle", FLACAMD_WIN_TRIANGLE}, {"welch", FLACAMD_WIN_WELCH} };
	if(!e || !spec || e->state != FLAC__STREAM_ENCODER_UNINITIALIZED) return 0;
	e->p.num_apodizations = 0;
	for(;;) {
		const char *sep = strchr(spec, ';');
		const size_t n = sep ? (size_t)(sep - spec) : strlen(spec);
		size_t i; int matched = 0;
		for(i = 0; i < sizeof plain / sizeof plain[0] && !matched; i++)
			if(n == strlen(plain[i].name) && 0 == strncmp(plain[i].name, spec, n)) { e->p.apod_type[e->p.num_apodizations++] = plain[i].type; matched = 1; }
		if(!matched && n > 7 && 0 == strncmp("gauss(", spec, 6)) {
			const float sd = (float)strtod(spec + 6, 0);
			if(sd > 0.0 && sd <= 0.5) { e->p.apod_param[e->p.num_apodizations] = sd; e->p.apod_type[e->p.num_apodizations++] = FLACAMD_WIN_GAUSS; }
		}
		else if(!matched && n > 7 && 0 == strncmp("tukey(", spec, 6)) {
			const float pp = (float)strtod(spec + 6, 0);
			if(pp >= 0.0 && pp <= 1.0) { e->p.apod_param[e->p.num_apodizations] = pp; e->p.apod_type[e->p.num_apodizations++] = FLACAMD_WIN_TUKEY; }
		}
		if(e->p.num_apodizations == 32) break;
		if(sep) spec = sep + 1; else break;
	}
	if(e->p.num_apodizations == 0) { e->p.num_apodizations = 1; e->p.apod_type[0] = FLACAMD_WIN_TUKEY; e->p.apod_param[0] = 0.5f; }
	return 1;
}

#define GETTER(name, type, expr) type FLAC__stream_encoder_##name(const FLAC__StreamEncoder *e) { return expr; }
GETTER(get_state, FLAC__StreamEncoderState, e->state)
GETTER(get_verify, FLAC__bool, e->verify)
GETTER(get_streamable_subset, FLAC__bool, e->streamable_subset)
GETTER(get_do_md5, FLAC__bool, e->do_md5)
GETTER(get_channels, unsigned, e->p.channels)
GETTER(get_bits_per_sample, unsigned, e->p.bits_per_sample)
GETTER(get_sample_rate, unsigned, e->p.sample_rate)
GETTER(get_blocksize, unsigned, e->p.blocksize)
GETTER(get_do_mid_side_stereo, FLAC__bool, (FLAC__bool)e->p.do_mid_side)
GETTER(get_loose_mid_side_stereo, FLAC__bool, (FLAC__bool)e->p.loose_mid_side)
GETTER(get_max_lpc_order, unsigned, e->p.max_lpc_order)
GETTER(get_qlp_coeff_precision, unsigned, e->p.qlp_coeff_precision)
GETTER(get_do_qlp_coeff_prec_search, FLAC__bool, (FLAC__bool)e->p.do_qlp_coeff_prec_search)
GETTER(get_do_escape_coding, FLAC__bool, e->do_escape_coding)
GETTER(get_do_exhaustive_model_search, FLAC__bool, (FLAC__bool)e->p.do_exhaustive_model_search)
GETTER(get_min_residual_partition_order, unsigned, e->p.min_residual_partition_order)
GETTER(get_max_residual_partition_order, unsigned, e->p.max_residual_partition_order)
GETTER(get_rice_parameter_search_dist, unsigned, e->rice_parameter_search_dist)
GETTER(get_total_samples_estimate, FLAC__uint64, e->total_samples_estimate)
const char *FLAC__stream_encoder_get_resolved_state_string(const FLAC__StreamEncoder *e) /* :1799-1808 */
{
	return e->state != FLAC__STREAM_ENCODER_VERIFY_DECODER_ERROR ? FLAC__StreamEncoderStateString[e->state] : FLAC__StreamDecoderStateString[e->vstate];
}
FLAC__StreamDecoderState FLAC__stream_encoder_get_verify_decoder_state(const FLAC__StreamEncoder *e) { return e->verify ? e->vstate : FLAC__STREAM_DECODER_UNINITIALIZED; }
void FLAC__stream_encoder_get_verify_decoder_error_stats(const FLAC__StreamEncoder *e, FLAC__uint64 *absolute_sample, unsigned *frame_number, unsigned *channel, unsigned *sample, FLAC__int32 *expected, FLAC__int32 *got)
{
	if(absolute_sample) *absolute_sample = e->vstats.absolute_sample;
	if(frame_number) *frame_number = e->vstats.frame_number;
	if(channel) *channel = e->vstats.channel;
	if(sample) *sample = e->vstats.sample;
	if(expected) *expected = e->vstats.expected;
	if(got) *got = e->vstats.got;
}

/* ---- writing: write_frame_ (:2431-2518) and write_bitbuffer_'s frame size tracking (:2423-2426) */
static FLAC__bool emit(FLAC__StreamEncoder *e, const uint8_t *buf, size_t bytes, unsigned samples)
{
	FLAC__uint64 pos = 0;
	if(e->tell_cb && e->tell_cb(e, &pos, e->client_data) == FLAC__STREAM_ENCODER_TELL_STATUS_ERROR) { e->state = FLAC__STREAM_ENCODER_CLIENT_ERROR; return 0; }
	if(samples == 0) { /* :2445-2451: the STREAMINFO and the first SEEKTABLE block going by */
		const unsigned type = buf[0] & 0x7f;
		if(type == FLAC__METADATA_TYPE_STREAMINFO) e->streaminfo_offset = pos;
		else if(type == FLAC__METADATA_TYPE_SEEKTABLE && e->seektable_offset == 0) e->seektable_offset = pos;
	}
	if(samples > 0 && e->seek_table && e->audio_offset > 0 && e->seek_table->num_points > 0) { /* :2458-2485: template points that fall into this frame */
		/* the frame's own length: finish() shrinks the blocksize before coding the last block (:1323-1324) */
		const FLAC__uint64 first = e->samples_written, last = first + samples - 1;
		unsigned i;
		for(i = e->first_seekpoint_to_check; i < e->seek_table->num_points; i++) {
			FLAC__StreamMetadata_SeekPoint *pt = &e->seek_table->points[i];
			if(pt->sample_number > last) break;
			if(pt->sample_number >= first) { pt->sample_number = first; pt->stream_offset = pos - e->audio_offset; pt->frame_samples = samples; }
			e->first_seekpoint_to_check++; /* several targets may share a frame: duplicates are removed by the sort at finish */
		}
	}
	if(e->write_cb(e, buf, bytes, samples, e->current_frame_number, e->client_data) != FLAC__STREAM_ENCODER_WRITE_STATUS_OK) { e->state = FLAC__STREAM_ENCODER_CLIENT_ERROR; return 0; }
	e->bytes_written += bytes; e->samples_written += samples;
	if(e->frames_written < e->current_frame_number + 1) e->frames_written = e->current_frame_number + 1;
	if(samples > 0) {
		FLAC__StreamMetadata_StreamInfo *si = &e->streaminfo.data.stream_info;
		if(bytes < si->min_framesize) si->min_framesize = (unsigned)bytes;
		if(bytes > si->max_framesize) si->max_framesize = (unsigned)bytes;
	}
	return 1;
}

/* file callbacks (:4262-4337) */
static FLAC__StreamEncoderSeekStatus file_seek(const FLAC__StreamEncoder *e, FLAC__uint64 off, void *cd) { (void)cd; return fseeko(e->file, (off_t)off, SEEK_SET) < 0 ? FLAC__STREAM_ENCODER_SEEK_STATUS_ERROR : FLAC__STREAM_ENCODER_SEEK_STATUS_OK; }
static FLAC__StreamEncoderTellStatus file_tell(const FLAC__StreamEncoder *e, FLAC__uint64 *off, void *cd)
{
	const off_t o = ftello(e->file); (void)cd;
	if(o < 0) return FLAC__STREAM_ENCODER_TELL_STATUS_ERROR;
	*off = (FLAC__uint64)o; return FLAC__STREAM_ENCODER_TELL_STATUS_OK;
}
static FLAC__StreamEncoderWriteStatus file_write(const FLAC__StreamEncoder *e, const FLAC__byte buf[], size_t bytes, unsigned samples, unsigned frame, void *cd)
{
	(void)cd; (void)frame;
	if(fwrite(buf, 1, bytes, e->file) != bytes) return FLAC__STREAM_ENCODER_WRITE_STATUS_FATAL_ERROR;
	if(e->progress_cb && samples > 0) /* the counters are updated after this callback returns, hence the +bytes etc. (:4324-4331) */
		e->progress_cb(e, e->bytes_written + bytes, e->samples_written + samples, e->frames_written + 1, e->total_frames_estimate, e->client_data);
	return FLAC__STREAM_ENCODER_WRITE_STATUS_OK;
}

static void put_be(uint8_t *p, uint64_t v, unsigned nbytes) { unsigned i; for(i = 0; i < nbytes; i++) p[i] = (uint8_t)(v >> (8 * (nbytes - 1 - i))); }

/* The head of a finished stream in one piece: "fLaC", STREAMINFO with every field filled in, and the
 * empty VORBIS_COMMENT that carries the vendor string (what a seekable client ends up with after
 * update_metadata_, :2521-2620).  For callers of the batch engine that assemble files themselves. */
size_t flacamd_write_stream_head(uint8_t out[FLACAMD_STREAM_HEAD_BYTES], uint32_t blocksize, uint32_t min_framesize, uint32_t max_framesize,
	uint32_t sample_rate, uint32_t channels, uint32_t bits_per_sample, uint64_t total_samples, const uint8_t md5[16])
{
	const size_t vlen = strlen(VENDOR_STRING);
	uint8_t *b = out;
	memcpy(b, "fLaC", 4); b += 4;
	b[0] = 0; put_be(b + 1, 34, 3);
	put_be(b + 4, blocksize, 2); put_be(b + 6, blocksize, 2); put_be(b + 8, min_framesize, 3); put_be(b + 11, max_framesize, 3);
	put_be(b + 14, ((uint64_t)sample_rate << 44) | ((uint64_t)(channels - 1) << 41) | ((uint64_t)(bits_per_sample - 1) << 36) | (total_samples & 0xFFFFFFFFFull), 8);
	if(md5) memcpy(b + 22, md5, 16); else memset(b + 22, 0, 16);
	b += 38;
	b[0] = 0x80 | 4; put_be(b + 1, 8 + vlen, 3);
	b[4] = (uint8_t)vlen; b[5] = b[6] = b[7] = 0; memcpy(b + 8, VENDOR_STRING, vlen); memset(b + 8 + vlen, 0, 4);
	b += 4 + 8 + vlen;
	return (size_t)(b - out);
}

static FLAC__StreamEncoderInitStatus init_internal(FLAC__StreamEncoder *e, FLAC__StreamEncoderWriteCallback wcb, FLAC__StreamEncoderSeekCallback scb,
	FLAC__StreamEncoderTellCallback tcb, FLAC__StreamEncoderMetadataCallback mcb, void *client_data)
{
	flacamd_params *p = &e->p;
	int rc, has_vorbis_comment = 0;
	/* the checks and their order are init_stream_internal_'s (:656-766) */
	if(e->state != FLAC__STREAM_ENCODER_UNINITIALIZED) return FLAC__STREAM_ENCODER_INIT_STATUS_ALREADY_INITIALIZED;
	if(!wcb || (scb && !tcb)) return FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_CALLBACKS;
	if(p->channels == 0 || p->channels > 8) return FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_NUMBER_OF_CHANNELS;
	if(p->channels != 2) { p->do_mid_side = 0; p->loose_mid_side = 0; } else if(!p->do_mid_side) p->loose_mid_side = 0;
	if(p->bits_per_sample < 4 || p->bits_per_sample > 24) return FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_BITS_PER_SAMPLE;
	if(p->sample_rate == 0 || p->sample_rate > 655350) return FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_SAMPLE_RATE;
	if(p->blocksize == 0) p->blocksize = p->max_lpc_order == 0 ? 1152 : 4096;
	if(p->blocksize < 16 || p->blocksize > 65535) return FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_BLOCK_SIZE;
	if(p->max_lpc_order > 32) return FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_MAX_LPC_ORDER;
	if(p->blocksize < p->max_lpc_order) return FLAC__STREAM_ENCODER_INIT_STATUS_BLOCK_SIZE_TOO_SMALL_FOR_LPC_ORDER;
	if(p->qlp_coeff_precision != 0 && (p->qlp_coeff_precision < 5 || p->qlp_coeff_precision > 15)) return FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_QLP_COEFF_PRECISION;
	if(e->streamable_subset) { /* :737-761 with format.c's subset predicates */
		const unsigned bs = p->blocksize, sr = p->sample_rate, b = p->bits_per_sample;
		if(bs > 16384 || (sr <= 48000 && bs > 4608)) return FLAC__STREAM_ENCODER_INIT_STATUS_NOT_STREAMABLE;
		if(sr >= (1u << 16) && !(sr % 1000 == 0 || sr % 10 == 0)) return FLAC__STREAM_ENCODER_INIT_STATUS_NOT_STREAMABLE;
		if(b != 8 && b != 12 && b != 16 && b != 20 && b != 24) return FLAC__STREAM_ENCODER_INIT_STATUS_NOT_STREAMABLE;
		if(p->max_residual_partition_order > 8) return FLAC__STREAM_ENCODER_INIT_STATUS_NOT_STREAMABLE;
		if(sr <= 48000 && (bs > 4608 || p->max_lpc_order > 12)) return FLAC__STREAM_ENCODER_INIT_STATUS_NOT_STREAMABLE;
	}
	rc = flacamd_params_resolve(p); /* fills the automatic precision, clips the partition orders */
	if(rc != FLACAMD_OK) { e->state = FLAC__STREAM_ENCODER_FRAMING_ERROR; return FLAC__STREAM_ENCODER_INIT_STATUS_ENCODER_ERROR; } /* legal for the reference, outside the engine's limits */

	/* client metadata (:783-845): first SEEKTABLE is the one that gets filled in; one SEEKTABLE and one
	 * VORBIS_COMMENT at most, no STREAMINFO, legal seek table / cue sheet / picture, one icon of each kind */
	e->seek_table = 0; e->first_seekpoint_to_check = 0; e->seektable_offset = 0;
	{
		unsigned i; int has_seektable = 0, has_icon1 = 0, has_icon2 = 0;
		has_vorbis_comment = 0;
		for(i = 0; i < e->num_metadata; i++)
			if(e->metadata[i] && e->metadata[i]->type == FLAC__METADATA_TYPE_SEEKTABLE) { e->seek_table = &e->metadata[i]->data.seek_table; break; }
		for(i = 0; i < e->num_metadata; i++) {
			const FLAC__StreamMetadata *m = e->metadata[i];
			if(!m) return FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_METADATA; /* the reference would crash here */
			switch(m->type) {
			case FLAC__METADATA_TYPE_STREAMINFO: return FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_METADATA;
			case FLAC__METADATA_TYPE_SEEKTABLE:
				if(has_seektable || !FLAC__format_seektable_is_legal(&m->data.seek_table)) return FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_METADATA;
				has_seektable = 1; break;
			case FLAC__METADATA_TYPE_VORBIS_COMMENT:
				if(has_vorbis_comment) return FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_METADATA;
				has_vorbis_comment = 1; break;
			case FLAC__METADATA_TYPE_CUESHEET:
				if(!FLAC__format_cuesheet_is_legal(&m->data.cue_sheet, m->data.cue_sheet.is_cd, 0)) return FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_METADATA;
				break;
			case FLAC__METADATA_TYPE_PICTURE:
				if(!FLAC__format_picture_is_legal(&m->data.picture, 0)) return FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_METADATA;
				if(m->data.picture.type == FLAC__STREAM_METADATA_PICTURE_TYPE_FILE_ICON_STANDARD) { /* 32x32 PNG (or a link), once */
					if(has_icon1) return FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_METADATA;
					has_icon1 = 1;
					if((strcmp(m->data.picture.mime_type, "image/png") && strcmp(m->data.picture.mime_type, "-->")) || m->data.picture.width != 32 || m->data.picture.height != 32)
						return FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_METADATA;
				}
				else if(m->data.picture.type == FLAC__STREAM_METADATA_PICTURE_TYPE_FILE_ICON) {
					if(has_icon2) return FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_METADATA;
					has_icon2 = 1;
				}
				break;
			default: break;
			}
		}
	}

	e->state = FLAC__STREAM_ENCODER_OK;
	e->write_cb = wcb; e->seek_cb = scb; e->tell_cb = tcb; e->metadata_cb = mcb; e->client_data = client_data;
	e->current_frame_number = 0;
	e->streaminfo_offset = 0; e->audio_offset = 0;

	e->next_frame_number = 0; e->batches_started = 0; e->filled = 0; e->fill = 0;
	rc = flacamd_pipe_create(&e->pipe, e->devices, e->ndevices, p, e->batch_frames, e->depth ? e->depth : (e->ndevices > 1 ? 2u : 3u));
	if(rc != FLACAMD_OK) { e->state = FLAC__STREAM_ENCODER_MEMORY_ALLOCATION_ERROR; return FLAC__STREAM_ENCODER_INIT_STATUS_ENCODER_ERROR; }
	e->fmt16 = flacamd_pipe_pcm_format(e->pipe) == FLACAMD_PCM_S16;
	e->batch_samples = (size_t)flacamd_pipe_batch_frames(e->pipe) * p->blocksize; /* the pipe rounds batches up to whole loose mid/side periods */
	e->nslots = flacamd_pipe_slots(e->pipe);
	e->slot_buf = (uint8_t**)calloc(e->nslots, sizeof(uint8_t*));
	e->md5_scratch = (uint8_t*)malloc((size_t)65536 * 4);
	if(!e->slot_buf || !e->md5_scratch) { e->state = FLAC__STREAM_ENCODER_MEMORY_ALLOCATION_ERROR; return FLAC__STREAM_ENCODER_INIT_STATUS_ENCODER_ERROR; }
	{ unsigned i; for(i = 0; i < e->nslots; i++) e->slot_buf[i] = (uint8_t*)flacamd_pipe_slot_buffer(e->pipe, i); }
	memset(&e->vstats, 0, sizeof e->vstats); /* :1012-1017 */
	if(e->verify) { /* :984-1011: a decoder of our own; sized for a whole batch plus false sync candidates */
		const unsigned bf = flacamd_pipe_batch_frames(e->pipe);
		e->vdec_cap = flacamd_pipe_max_batch_bytes(e->pipe);
		rc = flacamd_decoder_create(&e->vdec, e->devices[0], p->channels, p->bits_per_sample, p->blocksize, 2 * bf + 256, e->vdec_cap);
		e->vpcm = malloc((e->fmt16 ? sizeof(int16_t) : sizeof(int32_t)) * (size_t)bf * p->blocksize * p->channels);
		if(rc != FLACAMD_OK || !e->vpcm) { e->state = FLAC__STREAM_ENCODER_MEMORY_ALLOCATION_ERROR; return FLAC__STREAM_ENCODER_INIT_STATUS_ENCODER_ERROR; }
		e->vstate = FLAC__STREAM_DECODER_SEARCH_FOR_FRAME_SYNC;
	}
	if(e->do_md5) {
		md5_init(&e->md5);
		e->md5_produced = e->md5_hashed = 0; e->md5_quit = 0;
		if(pthread_mutex_init(&e->md5_lock, 0) || pthread_cond_init(&e->md5_data, 0) || pthread_cond_init(&e->md5_space, 0) || pthread_create(&e->md5_thread, 0, md5_worker, e)) {
			e->state = FLAC__STREAM_ENCODER_MEMORY_ALLOCATION_ERROR; return FLAC__STREAM_ENCODER_INIT_STATUS_ENCODER_ERROR;
		}
		e->md5_running = 1;
	}

	/* stream head: "fLaC", STREAMINFO, empty VORBIS_COMMENT carrying the vendor string (:1034-1106,
	 * layouts stream_encoder_framing.c:50-110, :134-160) */
	{
		static const uint8_t magic[4] = { 'f', 'L', 'a', 'C' };
		uint8_t b[64]; FLAC__StreamMetadata_StreamInfo *si = &e->streaminfo.data.stream_info;
		const size_t vlen = strlen(VENDOR_STRING);
		if(!emit(e, magic, 4, 0)) return FLAC__STREAM_ENCODER_INIT_STATUS_ENCODER_ERROR;
		e->streaminfo.type = 0; e->streaminfo.is_last = 0; e->streaminfo.length = 34;
		si->min_blocksize = si->max_blocksize = p->blocksize; si->min_framesize = si->max_framesize = 0;
		si->sample_rate = p->sample_rate; si->channels = p->channels; si->bits_per_sample = p->bits_per_sample;
		si->total_samples = e->total_samples_estimate; memset(si->md5sum, 0, 16);
		if(e->do_md5) md5_init(&e->md5);
		b[0] = 0; put_be(b + 1, 34, 3);
		put_be(b + 4, si->min_blocksize, 2); put_be(b + 6, si->max_blocksize, 2); put_be(b + 8, 0, 3); put_be(b + 11, 0, 3);
		put_be(b + 14, ((uint64_t)si->sample_rate << 44) | ((uint64_t)(si->channels - 1) << 41) | ((uint64_t)(si->bits_per_sample - 1) << 36) | (si->total_samples & 0xFFFFFFFFFull), 8);
		memset(b + 22, 0, 16);
		if(!emit(e, b, 38, 0)) return FLAC__STREAM_ENCODER_INIT_STATUS_ENCODER_ERROR;
		si->min_framesize = (1u << 24) - 1; si->total_samples = 0; /* :1075-1077 */
		if(!has_vorbis_comment) { /* :1089-1106 */
			b[0] = (uint8_t)((e->num_metadata == 0 ? 0x80 : 0) | 4); put_be(b + 1, 8 + vlen, 3);
			b[4] = (uint8_t)vlen; b[5] = b[6] = b[7] = 0; memcpy(b + 8, VENDOR_STRING, vlen); memset(b + 8 + vlen, 0, 4);
			if(!emit(e, b, 4 + 8 + vlen, 0)) return FLAC__STREAM_ENCODER_INIT_STATUS_ENCODER_ERROR;
		}
	}
	{ /* the client's blocks in the order given (:1108-1121) */
		unsigned i;
		for(i = 0; i < e->num_metadata; i++) {
			uint8_t *img = 0; size_t len = 0; FLAC__bool ok;
			e->metadata[i]->is_last = (i == e->num_metadata - 1);
			if(flacamd_metadata_block_image(e->metadata[i], VENDOR_STRING, &img, &len) != 0) { e->state = FLAC__STREAM_ENCODER_FRAMING_ERROR; return FLAC__STREAM_ENCODER_INIT_STATUS_ENCODER_ERROR; }
			ok = emit(e, img, len, 0);
			free(img);
			if(!ok) return FLAC__STREAM_ENCODER_INIT_STATUS_ENCODER_ERROR;
		}
	}
	if(e->tell_cb && e->tell_cb(e, &e->audio_offset, e->client_data) == FLAC__STREAM_ENCODER_TELL_STATUS_ERROR) { e->state = FLAC__STREAM_ENCODER_CLIENT_ERROR; return FLAC__STREAM_ENCODER_INIT_STATUS_ENCODER_ERROR; }
	return FLAC__STREAM_ENCODER_INIT_STATUS_OK;
}

FLAC__StreamEncoderInitStatus FLAC__stream_encoder_init_stream(FLAC__StreamEncoder *e, FLAC__StreamEncoderWriteCallback wcb, FLAC__StreamEncoderSeekCallback scb,
	FLAC__StreamEncoderTellCallback tcb, FLAC__StreamEncoderMetadataCallback mcb, void *client_data)
{
	return init_internal(e, wcb, scb, tcb, mcb, client_data);
}

FLAC__StreamEncoderInitStatus FLAC__stream_encoder_init_FILE(FLAC__StreamEncoder *e, FILE *file, FLAC__StreamEncoderProgressCallback pcb, void *client_data)
{
	FLAC__StreamEncoderInitStatus st;
	if(e->state != FLAC__STREAM_ENCODER_UNINITIALIZED) return FLAC__STREAM_ENCODER_INIT_STATUS_ALREADY_INITIALIZED;
	if(!file) { e->state = FLAC__STREAM_ENCODER_IO_ERROR; return FLAC__STREAM_ENCODER_INIT_STATUS_ENCODER_ERROR; }
	e->file = file; e->progress_cb = pcb; e->bytes_written = 0; e->samples_written = 0; e->frames_written = 0;
	st = init_internal(e, file_write, file == stdout ? 0 : file_seek, file == stdout ? 0 : file_tell, 0, client_data);
	if(st != FLAC__STREAM_ENCODER_INIT_STATUS_OK) return st;
	e->total_frames_estimate = (unsigned)((e->total_samples_estimate + e->p.blocksize - 1) / e->p.blocksize);
	return st;
}

FLAC__StreamEncoderInitStatus FLAC__stream_encoder_init_file(FLAC__StreamEncoder *e, const char *filename, FLAC__StreamEncoderProgressCallback pcb, void *client_data)
{
	FILE *f;
	if(e->state != FLAC__STREAM_ENCODER_UNINITIALIZED) return FLAC__STREAM_ENCODER_INIT_STATUS_ALREADY_INITIALIZED;
	f = filename ? fopen(filename, "w+b") : stdout; /* :1290-1308 */
	if(!f) { e->state = FLAC__STREAM_ENCODER_IO_ERROR; return FLAC__STREAM_ENCODER_INIT_STATUS_ENCODER_ERROR; }
	return FLAC__stream_encoder_init_FILE(e, f, pcb, client_data);
}

/* The Ogg inits as a library built without Ogg transport has them (:1156-1308 over the `#if !FLAC__HAS_OGG`
 * exit of init_stream_internal_, :659-662): every check that comes before that exit runs - and init_ogg_file has
 * by then created or truncated the file, whose FILE stays with the object - and the answer is
 * UNSUPPORTED_CONTAINER with the object still uninitialized. */
FLAC__StreamEncoderInitStatus FLAC__stream_encoder_init_ogg_stream(FLAC__StreamEncoder *e, FLAC__StreamEncoderReadCallback rcb, FLAC__StreamEncoderWriteCallback wcb,
	FLAC__StreamEncoderSeekCallback scb, FLAC__StreamEncoderTellCallback tcb, FLAC__StreamEncoderMetadataCallback mcb, void *client_data)
{
	(void)rcb; (void)wcb; (void)scb; (void)tcb; (void)mcb; (void)client_data;
	if(e->state != FLAC__STREAM_ENCODER_UNINITIALIZED) return FLAC__STREAM_ENCODER_INIT_STATUS_ALREADY_INITIALIZED;
	return FLAC__STREAM_ENCODER_INIT_STATUS_UNSUPPORTED_CONTAINER;
}

FLAC__StreamEncoderInitStatus FLAC__stream_encoder_init_ogg_FILE(FLAC__StreamEncoder *e, FILE *file, FLAC__StreamEncoderProgressCallback pcb, void *client_data)
{
	(void)client_data;
	if(e->state != FLAC__STREAM_ENCODER_UNINITIALIZED) return FLAC__STREAM_ENCODER_INIT_STATUS_ALREADY_INITIALIZED;
	if(!file) { e->state = FLAC__STREAM_ENCODER_IO_ERROR; return FLAC__STREAM_ENCODER_INIT_STATUS_ENCODER_ERROR; }
	e->file = file; e->progress_cb = pcb; e->bytes_written = 0; e->samples_written = 0; e->frames_written = 0;
	return FLAC__STREAM_ENCODER_INIT_STATUS_UNSUPPORTED_CONTAINER;
}

FLAC__StreamEncoderInitStatus FLAC__stream_encoder_init_ogg_file(FLAC__StreamEncoder *e, const char *filename, FLAC__StreamEncoderProgressCallback pcb, void *client_data)
{
	FILE *f;
	if(e->state != FLAC__STREAM_ENCODER_UNINITIALIZED) return FLAC__STREAM_ENCODER_INIT_STATUS_ALREADY_INITIALIZED;
	f = filename ? fopen(filename, "w+b") : stdout;
	if(!f) { e->state = FLAC__STREAM_ENCODER_IO_ERROR; return FLAC__STREAM_ENCODER_INIT_STATUS_ENCODER_ERROR; }
	return FLAC__stream_encoder_init_ogg_FILE(e, f, pcb, client_data);
}

/* :1378-1393 without libogg */
FLAC__bool FLAC__stream_encoder_set_ogg_serial_number(FLAC__StreamEncoder *e, long value) { (void)e; (void)value; return 0; }

/* sample i (interleaved index) of a batch's PCM, which is int16 or int32 */
static int32_t pcm_at(const FLAC__StreamEncoder *e, const void *pcm, size_t i) { return e->fmt16 ? (int32_t)((const int16_t*)pcm)[i] : ((const int32_t*)pcm)[i]; }

/* first difference between decoded and submitted samples of one frame, searched channel by channel like
 * verify_write_callback_ (:4207-4231); `at` = interleaved index of the frame's first sample in both buffers */
static int frame_differs(FLAC__StreamEncoder *e, const void *got, size_t got_at, const void *expected, size_t exp_at, unsigned blocksize, FLAC__uint64 first_sample)
{
	const unsigned ch = e->p.channels; unsigned c, i; const size_t w = e->fmt16 ? 2u : 4u;
	if(0 == memcmp((const uint8_t*)got + got_at * w, (const uint8_t*)expected + exp_at * w, w * (size_t)blocksize * ch)) return 0;
	for(c = 0; c < ch; c++) for(i = 0; i < blocksize; i++) {
		const int32_t g = pcm_at(e, got, got_at + (size_t)i * ch + c), x = pcm_at(e, expected, exp_at + (size_t)i * ch + c);
		if(g != x) {
			e->vstats.absolute_sample = first_sample + i; e->vstats.frame_number = (unsigned)(first_sample / blocksize); /* the frame's own length, as the reference divides (:4214) */
			e->vstats.channel = c; e->vstats.sample = i; e->vstats.expected = x; e->vstats.got = g;
			return 1;
		}
	}
	return 0;
}

/* Decode the batch just coded on the device and compare with the samples it was made from.  Returns the number of
 * leading frames that verified; when that is less than nframes, *bad is the encoder state the failing frame asks for. */
static unsigned verify_batch(FLAC__StreamEncoder *e, uint8_t *frames, const uint32_t *frame_bytes, void *pcm, size_t nsamples, unsigned nframes, size_t total_bytes, unsigned first_frame, FLAC__StreamEncoderState *bad)
{
	const unsigned N = e->p.blocksize, ch = e->p.channels;
	const int fmt = e->fmt16 ? FLACAMD_PCM_S16 : FLACAMD_PCM_S32;
	const FLAC__uint64 batch_first = (FLAC__uint64)first_frame * N;
	uint64_t info[4] = { 0, 0, 0, 0 };
	unsigned f; size_t off = 0; int rc;
	if(total_bytes > e->vdec_cap) { /* frames beyond the usual bound: the verify decoder takes a batch of that size too */
		flacamd_decoder_destroy(e->vdec); e->vdec = 0; e->vdec_cap = total_bytes;
		if(flacamd_decoder_create(&e->vdec, e->devices[0], ch, e->p.bits_per_sample, N, 2 * flacamd_pipe_batch_frames(e->pipe) + 256, e->vdec_cap) != FLACAMD_OK) {
			*bad = FLAC__STREAM_ENCODER_MEMORY_ALLOCATION_ERROR; return 0;
		}
	}
	if(e->frame_fault.armed && e->frame_fault.frame_number >= first_frame && e->frame_fault.frame_number < first_frame + nframes) {
		size_t o = 0; const unsigned rel = e->frame_fault.frame_number - first_frame;
		for(f = 0; f < rel; f++) o += frame_bytes[f];
		if(e->frame_fault.byte_index < frame_bytes[rel]) frames[o + e->frame_fault.byte_index] ^= (uint8_t)e->frame_fault.mask;
		e->frame_fault.armed = 0;
	}
	if(e->sample_fault.armed && e->sample_fault.absolute_sample >= batch_first && e->sample_fault.absolute_sample < batch_first + nsamples && e->sample_fault.channel < ch) {
		const size_t i = (size_t)(e->sample_fault.absolute_sample - batch_first) * ch + e->sample_fault.channel;
		if(e->fmt16) ((int16_t*)pcm)[i] ^= (int16_t)e->sample_fault.mask; else ((int32_t*)pcm)[i] ^= e->sample_fault.mask;
		e->sample_fault.armed = 0;
	}
	rc = flacamd_decode_host(e->vdec, frames, total_bytes, e->vpcm, fmt, nsamples, info);
	if(rc == FLACAMD_OK && info[0] == nframes && info[1] == nsamples && info[3] == total_bytes) {
		if(0 == memcmp(e->vpcm, pcm, (e->fmt16 ? 2u : 4u) * nsamples * ch)) return nframes;
		for(f = 0; f < nframes; f++) {
			const unsigned bs = (f + 1 == nframes) ? (unsigned)(nsamples - (size_t)f * N) : N;
			if(frame_differs(e, e->vpcm, (size_t)f * N * ch, pcm, (size_t)f * N * ch, bs, batch_first + (FLAC__uint64)f * N)) { *bad = FLAC__STREAM_ENCODER_VERIFY_MISMATCH_IN_AUDIO_DATA; return f; }
		}
		return nframes;
	}
	/* the batch did not decode as a whole: find the frame, one at a time */
	for(f = 0; f < nframes; f++) {
		const unsigned bs = (f + 1 == nframes) ? (unsigned)(nsamples - (size_t)f * N) : N;
		rc = flacamd_decode_host(e->vdec, frames + off, frame_bytes[f], e->vpcm, fmt, bs, info);
		if(rc != FLACAMD_OK || info[0] != 1 || info[1] != bs || info[3] != frame_bytes[f]) {
			/* what verify_error_callback_ leaves behind (:4245-4250); the reference's decoder would be hunting for the next sync */
			*bad = FLAC__STREAM_ENCODER_VERIFY_DECODER_ERROR; e->vstate = (rc == FLACAMD_OK || rc == FLACAMD_E_BAD_STREAM) ? FLAC__STREAM_DECODER_SEARCH_FOR_FRAME_SYNC : FLAC__STREAM_DECODER_ABORTED;
			return f;
		}
		if(frame_differs(e, e->vpcm, 0, pcm, (size_t)f * N * ch, bs, batch_first + (FLAC__uint64)f * N)) { *bad = FLAC__STREAM_ENCODER_VERIFY_MISMATCH_IN_AUDIO_DATA; return f; }
		off += frame_bytes[f];
	}
	*bad = FLAC__STREAM_ENCODER_VERIFY_DECODER_ERROR; e->vstate = FLAC__STREAM_DECODER_ABORTED; /* every frame decodes alone but the batch does not */
	return 0;
}

/* Take the oldest batch in flight back from the GPU (waiting for it if need be) and replay the per-frame writes
 * of write_bitbuffer_ / write_frame_ (:2384-2518) in stream order. */
static FLAC__bool drain_one(FLAC__StreamEncoder *e)
{
	const uint8_t *frames = 0; const uint32_t *frame_bytes = 0; const void *pcm = 0;
	uint32_t nframes = 0, first = 0; uint64_t total = 0, nsamples = 0; size_t off = 0; unsigned f;
	const unsigned N = e->p.blocksize;
	const int rc = flacamd_pipe_collect(e->pipe, &frames, &frame_bytes, &nframes, &total, &pcm, &nsamples, &first);
	if(rc != FLACAMD_OK) { e->state = (rc == FLACAMD_E_OUT_OF_MEMORY) ? FLAC__STREAM_ENCODER_MEMORY_ALLOCATION_ERROR : FLAC__STREAM_ENCODER_FRAMING_ERROR; return 0; }
	if(e->verify) { /* write_bitbuffer_ verifies a frame before it is written (:2396-2411): stop ahead of the first bad one */
		FLAC__StreamEncoderState bad = FLAC__STREAM_ENCODER_OK;
		const unsigned good = verify_batch(e, (uint8_t*)frames, frame_bytes, (void*)pcm, (size_t)nsamples, nframes, (size_t)total, first, &bad);
		if(good < nframes) {
			for(f = 0; f < good; f++) {
				if(!emit(e, frames + off, frame_bytes[f], N)) return 0;
				off += frame_bytes[f]; e->current_frame_number++; e->streaminfo.data.stream_info.total_samples += N;
			}
			e->state = bad;
			return 0;
		}
	}
	for(f = 0; f < nframes; f++) {
		const unsigned samples = (f + 1 == nframes) ? (unsigned)(nsamples - (size_t)f * N) : N;
		if(!emit(e, frames + off, frame_bytes[f], samples)) return 0;
		off += frame_bytes[f];
		e->current_frame_number++;
		e->streaminfo.data.stream_info.total_samples += samples; /* :2920 */
	}
	(void)flacamd_pipe_release(e->pipe);
	return 1;
}

/* ---- MD5 worker: hashes the samples in the order they were handed over, out of the upload buffers */
static void *md5_worker(void *arg)
{
	FLAC__StreamEncoder *e = (FLAC__StreamEncoder*)arg;
	const size_t ch = e->p.channels, wbytes = e->fmt16 ? 2u : 4u;
	pthread_mutex_lock(&e->md5_lock);
	for(;;) {
		uint64_t from, to;
		while(e->md5_hashed == e->md5_produced && !e->md5_quit) pthread_cond_wait(&e->md5_data, &e->md5_lock);
		if(e->md5_hashed == e->md5_produced) break;
		from = e->md5_hashed; to = e->md5_produced;
		pthread_mutex_unlock(&e->md5_lock);
		while(from < to) {
			const uint64_t q = from / e->batch_samples; const size_t at = (size_t)(from % e->batch_samples);
			size_t n = (size_t)(to - from);
			const uint8_t *src = e->slot_buf[q % e->nslots] + at * ch * wbytes;
			if(n > e->batch_samples - at) n = e->batch_samples - at;
			if(n > (1u << 18)) n = 1u << 18; /* report progress often enough for the filling thread to get its buffer back */
			if(e->fmt16) md5_feed_pcm16(&e->md5, (const int16_t*)src, n * ch, e->p.bits_per_sample, e->md5_scratch, (size_t)65536 * 4);
			else md5_feed_pcm(&e->md5, (const int32_t*)src, n * ch, e->p.bits_per_sample, e->md5_scratch, (size_t)65536 * 4);
			from += n;
			pthread_mutex_lock(&e->md5_lock);
			e->md5_hashed = from;
			pthread_cond_broadcast(&e->md5_space);
			pthread_mutex_unlock(&e->md5_lock);
		}
		pthread_mutex_lock(&e->md5_lock);
	}
	pthread_mutex_unlock(&e->md5_lock);
	return 0;
}

/* let the worker finish what was handed over, then end it */
static void md5_stop(FLAC__StreamEncoder *e)
{
	if(!e->md5_running) return;
	pthread_mutex_lock(&e->md5_lock);
	e->md5_quit = 1;
	pthread_cond_broadcast(&e->md5_data);
	pthread_mutex_unlock(&e->md5_lock);
	pthread_join(e->md5_thread, 0);
	pthread_mutex_destroy(&e->md5_lock); pthread_cond_destroy(&e->md5_data); pthread_cond_destroy(&e->md5_space);
	e->md5_running = 0;
}

static void md5_publish(FLAC__StreamEncoder *e, size_t nsamples)
{
	if(!e->md5_running) return;
	pthread_mutex_lock(&e->md5_lock);
	e->md5_produced += nsamples;
	pthread_cond_signal(&e->md5_data);
	pthread_mutex_unlock(&e->md5_lock);
}

/* The buffer of the next batch to fill.  All batches may be in flight: then the oldest is taken back first (its
 * frames are written out); and the MD5 worker must be done with the batch that used this buffer last. */
static FLAC__bool begin_batch(FLAC__StreamEncoder *e)
{
	while((e->fill = (uint8_t*)flacamd_pipe_fill_buffer(e->pipe)) == 0) if(!drain_one(e)) return 0;
	e->filled = 0;
	if(e->md5_running && e->batches_started >= e->nslots) {
		const uint64_t need = (e->batches_started - e->nslots + 1) * (uint64_t)e->batch_samples;
		pthread_mutex_lock(&e->md5_lock);
		while(e->md5_hashed < need) pthread_cond_wait(&e->md5_space, &e->md5_lock);
		pthread_mutex_unlock(&e->md5_lock);
	}
	e->batches_started++;
	return 1;
}

/* hand the batch being filled to the GPUs; batches that have come back meanwhile are written out */
static FLAC__bool submit_batch(FLAC__StreamEncoder *e, int final_block)
{
	const unsigned N = e->p.blocksize;
	const int rc = flacamd_pipe_submit(e->pipe, e->filled, final_block, e->next_frame_number);
	if(rc != FLACAMD_OK) { e->state = (rc == FLACAMD_E_OUT_OF_MEMORY) ? FLAC__STREAM_ENCODER_MEMORY_ALLOCATION_ERROR : FLAC__STREAM_ENCODER_FRAMING_ERROR; return 0; }
	e->next_frame_number += (unsigned)((e->filled + N - 1) / N);
	e->fill = 0; e->filled = 0;
	/* with verify on, a bad frame is reported by the call that completes its block, as in the reference (:2396-2411): no batch stays in flight */
	while(flacamd_pipe_in_flight(e->pipe) > 0 && (e->verify || flacamd_pipe_ready(e->pipe) == 1)) if(!drain_one(e)) return 0;
	return 1;
}

/* ---- helper threads for the copy of a large call */
#define FILL_MAX_THREADS 8
#define FILL_MIN_VALUES ((size_t)1 << 19)   /* below this a call is copied by the caller alone */
struct fill_part { const int32_t *src; void *dst; size_t nvalues; };
struct fill_pool {
	pthread_t th[FILL_MAX_THREADS]; unsigned nth; int narrow;
	pthread_mutex_t lock; pthread_cond_t go, done;
	unsigned generation, pending; int quit;
	struct fill_part part[FILL_MAX_THREADS];
	struct fill_arg { struct fill_pool *pool; unsigned index; } arg[FILL_MAX_THREADS];
};

static void fill_copy(int narrow, const struct fill_part *p)
{
	if(narrow) flacamd_pcm_narrow((int16_t*)p->dst, p->src, p->nvalues);
	else memcpy(p->dst, p->src, sizeof(int32_t) * p->nvalues);
}

static void *fill_worker(void *argp)
{
	struct fill_arg *a = (struct fill_arg*)argp; struct fill_pool *pl = a->pool;
	unsigned seen = 0;
	pthread_mutex_lock(&pl->lock);
	for(;;) {
		while(pl->generation == seen && !pl->quit) pthread_cond_wait(&pl->go, &pl->lock);
		if(pl->quit) break;
		seen = pl->generation;
		pthread_mutex_unlock(&pl->lock);
		fill_copy(pl->narrow, &pl->part[a->index]);
		pthread_mutex_lock(&pl->lock);
		if(--pl->pending == 0) pthread_cond_signal(&pl->done);
		/* a client streaming large buffers calls again within microseconds: look out for the next job for a short
		 * while before going to sleep (a sleeping thread takes tens of microseconds to wake, as long as its share of a call) */
		{
			int spin;
			pthread_mutex_unlock(&pl->lock);
			for(spin = 0; spin < 4000 && __atomic_load_n(&pl->generation, __ATOMIC_ACQUIRE) == seen && !__atomic_load_n(&pl->quit, __ATOMIC_ACQUIRE); spin++) cpu_relax();
			pthread_mutex_lock(&pl->lock);
		}
	}
	pthread_mutex_unlock(&pl->lock);
	return 0;
}

static void pool_stop(FLAC__StreamEncoder *e)
{
	struct fill_pool *pl = e->pool; unsigned i;
	if(!pl) return;
	pthread_mutex_lock(&pl->lock); pl->quit = 1; pthread_cond_broadcast(&pl->go); pthread_mutex_unlock(&pl->lock);
	for(i = 0; i < pl->nth; i++) pthread_join(pl->th[i], 0);
	pthread_mutex_destroy(&pl->lock); pthread_cond_destroy(&pl->go); pthread_cond_destroy(&pl->done);
	free(pl); e->pool = 0;
}

/* copy nvalues interleaved samples into the batch being filled, spread over the helper threads when the call is large */
static void fill_values(FLAC__StreamEncoder *e, void *dst, const int32_t *src, size_t nvalues)
{
	struct fill_part whole; unsigned i, helpers;
	whole.src = src; whole.dst = dst; whole.nvalues = nvalues;
	if(nvalues < FILL_MIN_VALUES || e->fill_threads < 2) { fill_copy(e->fmt16, &whole); return; }
	if(!e->pool) { /* started by the first large call */
		struct fill_pool *pl = (struct fill_pool*)calloc(1, sizeof *pl);
		if(!pl || pthread_mutex_init(&pl->lock, 0) || pthread_cond_init(&pl->go, 0) || pthread_cond_init(&pl->done, 0)) { free(pl); e->fill_threads = 1; fill_copy(e->fmt16, &whole); return; }
		pl->narrow = e->fmt16;
		for(i = 0; i + 1 < e->fill_threads && i < FILL_MAX_THREADS; i++) {
			pl->arg[i].pool = pl; pl->arg[i].index = i;
			if(pthread_create(&pl->th[i], 0, fill_worker, &pl->arg[i])) break;
			pl->nth++;
		}
		e->pool = pl;
	}
	helpers = e->pool->nth;
	if(helpers == 0) { fill_copy(e->fmt16, &whole); return; }
	{
		struct fill_pool *pl = e->pool;
		const size_t per = (nvalues / (helpers + 1) + 63) & ~(size_t)63;
		const size_t w = e->fmt16 ? 2u : 4u;
		size_t at = 0;
		pthread_mutex_lock(&pl->lock);
		for(i = 0; i < helpers; i++) {
			size_t n = per; if(at + n > nvalues) n = nvalues - at;
			pl->part[i].src = src + at; pl->part[i].dst = (uint8_t*)dst + at * w; pl->part[i].nvalues = n; at += n;
		}
		pl->pending = helpers; pl->generation++;
		pthread_cond_broadcast(&pl->go);
		pthread_mutex_unlock(&pl->lock);
		whole.src = src + at; whole.dst = (uint8_t*)dst + at * w; whole.nvalues = nvalues - at;
		fill_copy(e->fmt16, &whole);
		pthread_mutex_lock(&pl->lock);
		while(pl->pending) pthread_cond_wait(&pl->done, &pl->lock);
		pthread_mutex_unlock(&pl->lock);
	}
}

/* Room for samples in the batch being filled.  The reference codes a block once blocksize+1 samples are known
 * (:2006, :2062): a full batch is only submitted when a later sample has arrived, so that finish() can still
 * code its last block as the final one.  Returns the number of wide samples that fit (at least 1), 0 on error. */
static size_t room(FLAC__StreamEncoder *e)
{
	if(e->fill && e->filled == e->batch_samples) { if(!submit_batch(e, 0)) return 0; }
	if(!e->fill && !begin_batch(e)) return 0;
	return e->batch_samples - e->filled;
}

FLAC__bool FLAC__stream_encoder_process_interleaved(FLAC__StreamEncoder *e, const FLAC__int32 buffer[], unsigned samples)
{
	size_t done = 0; const unsigned ch = e ? e->p.channels : 0;
	if(!e || e->state != FLAC__STREAM_ENCODER_OK) return 0;
	while(done < samples) {
		size_t n = room(e);
		if(n == 0) return 0;
		if(n > samples - done) n = samples - done;
		fill_values(e, e->fill + e->filled * ch * (e->fmt16 ? 2u : 4u), buffer + done * ch, n * ch);
		e->filled += n; done += n;
		md5_publish(e, n);
	}
	return 1;
}

FLAC__bool FLAC__stream_encoder_process(FLAC__StreamEncoder *e, const FLAC__int32 * const buffer[], unsigned samples)
{
	size_t done = 0; const unsigned ch = e ? e->p.channels : 0;
	if(!e || e->state != FLAC__STREAM_ENCODER_OK) return 0;
	while(done < samples) {
		size_t n = room(e), i; unsigned c;
		if(n == 0) return 0;
		if(n > samples - done) n = samples - done;
		if(e->fmt16) {
			int16_t *dst = (int16_t*)e->fill + e->filled * ch;
			if(ch == 2) { const FLAC__int32 *l = buffer[0] + done, *r = buffer[1] + done; for(i = 0; i < n; i++) { dst[2 * i] = (int16_t)l[i]; dst[2 * i + 1] = (int16_t)r[i]; } }
			else for(c = 0; c < ch; c++) { const FLAC__int32 *src = buffer[c] + done; for(i = 0; i < n; i++) dst[i * ch + c] = (int16_t)src[i]; }
		}
		else {
			int32_t *dst = (int32_t*)e->fill + e->filled * ch;
			for(c = 0; c < ch; c++) { const FLAC__int32 *src = buffer[c] + done; for(i = 0; i < n; i++) dst[i * ch + c] = src[i]; }
		}
		e->filled += n; done += n;
		md5_publish(e, n);
	}
	return 1;
}

FLAC__bool FLAC__stream_encoder_finish(FLAC__StreamEncoder *e) /* :1310-1377 */
{
	FLAC__bool error = 0;
	if(!e) return 0;
	if(e->state == FLAC__STREAM_ENCODER_UNINITIALIZED) return 1;
	if(e->state == FLAC__STREAM_ENCODER_OK && e->fill && e->filled > 0) { if(!submit_batch(e, 1)) error = 1; }
	while(e->state == FLAC__STREAM_ENCODER_OK && e->pipe && flacamd_pipe_in_flight(e->pipe) > 0) if(!drain_one(e)) error = 1;
	if(e->do_md5 && e->md5_running) { md5_stop(e); md5_final(&e->md5, e->streaminfo.data.stream_info.md5sum); }
	if(e->state == FLAC__STREAM_ENCODER_OK) {
		if(e->seek_cb) { /* update_metadata_ (:2521-2620): MD5, total samples, min/max frame size patched in place */
			const FLAC__StreamMetadata_StreamInfo *si = &e->streaminfo.data.stream_info;
			uint8_t b[6]; int ok = 1;
			FLAC__StreamEncoderSeekStatus ss;
			e->current_frame_number = 0;
			if((ss = e->seek_cb(e, e->streaminfo_offset + 22, e->client_data)) == FLAC__STREAM_ENCODER_SEEK_STATUS_OK) {
				if(e->write_cb(e, si->md5sum, 16, 0, 0, e->client_data) != FLAC__STREAM_ENCODER_WRITE_STATUS_OK) { e->state = FLAC__STREAM_ENCODER_CLIENT_ERROR; ok = 0; }
			} else { if(ss == FLAC__STREAM_ENCODER_SEEK_STATUS_ERROR) e->state = FLAC__STREAM_ENCODER_CLIENT_ERROR; ok = 0; }
			if(ok) {
				b[0] = (uint8_t)(((si->bits_per_sample - 1) << 4) | ((si->total_samples >> 32) & 0x0F)); put_be(b + 1, si->total_samples & 0xFFFFFFFFu, 4);
				if((ss = e->seek_cb(e, e->streaminfo_offset + 17, e->client_data)) == FLAC__STREAM_ENCODER_SEEK_STATUS_OK) {
					if(e->write_cb(e, b, 5, 0, 0, e->client_data) != FLAC__STREAM_ENCODER_WRITE_STATUS_OK) { e->state = FLAC__STREAM_ENCODER_CLIENT_ERROR; ok = 0; }
				} else { if(ss == FLAC__STREAM_ENCODER_SEEK_STATUS_ERROR) e->state = FLAC__STREAM_ENCODER_CLIENT_ERROR; ok = 0; }
			}
			if(ok) {
				put_be(b, si->min_framesize, 3); put_be(b + 3, si->max_framesize, 3);
				if((ss = e->seek_cb(e, e->streaminfo_offset + 8, e->client_data)) == FLAC__STREAM_ENCODER_SEEK_STATUS_OK) {
					if(e->write_cb(e, b, 6, 0, 0, e->client_data) != FLAC__STREAM_ENCODER_WRITE_STATUS_OK) { e->state = FLAC__STREAM_ENCODER_CLIENT_ERROR; ok = 0; }
				} else { if(ss == FLAC__STREAM_ENCODER_SEEK_STATUS_ERROR) e->state = FLAC__STREAM_ENCODER_CLIENT_ERROR; ok = 0; }
			}
			if(ok && e->state == FLAC__STREAM_ENCODER_OK && e->seek_table && e->seek_table->num_points > 0 && e->seektable_offset > 0) { /* :2627-2672 */
				unsigned i; uint8_t pt[18];
				FLAC__format_seektable_sort(e->seek_table);
				if((ss = e->seek_cb(e, e->seektable_offset + FLAC__STREAM_METADATA_HEADER_LENGTH, e->client_data)) == FLAC__STREAM_ENCODER_SEEK_STATUS_OK) {
					for(i = 0; i < e->seek_table->num_points; i++) {
						put_be(pt, e->seek_table->points[i].sample_number, 8); put_be(pt + 8, e->seek_table->points[i].stream_offset, 8); put_be(pt + 16, e->seek_table->points[i].frame_samples, 2);
						if(e->write_cb(e, pt, 18, 0, 0, e->client_data) != FLAC__STREAM_ENCODER_WRITE_STATUS_OK) { e->state = FLAC__STREAM_ENCODER_CLIENT_ERROR; break; }
					}
				} else if(ss == FLAC__STREAM_ENCODER_SEEK_STATUS_ERROR) e->state = FLAC__STREAM_ENCODER_CLIENT_ERROR;
			}
			if(e->state != FLAC__STREAM_ENCODER_OK) error = 1;
		}
		if(e->metadata_cb) e->metadata_cb(e, &e->streaminfo, e->client_data);
	}
	else error = 1;
	if(e->file) { if(e->file != stdout) fclose(e->file); e->file = 0; }
	free_buffers(e);
	{
		const FLAC__StreamEncoderState st = e->state;
		set_defaults(e);
		if(error) e->state = st == FLAC__STREAM_ENCODER_OK ? FLAC__STREAM_ENCODER_CLIENT_ERROR : st; /* :1372-1376: the state survives a failed finish */
	}
	return !error;
}
