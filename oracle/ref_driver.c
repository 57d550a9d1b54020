/* TEST INFRASTRUCTURE - never linked into or called by the product path.
 *
 * Thin driver over the REFERENCE libFLAC's public API (include/FLAC/stream_encoder.h,
 * include/FLAC/stream_decoder.h).  It is compiled together with the reference's own
 * objects into oracle/_ref/libflacref.so by oracle/Makefile; this file is this
 * repository's code, the reference sources stay where they lie.
 *
 * Purpose: run the real reference encoder/decoder on in-memory PCM so that
 *  - the CPU restatement in flac_oracle.c can be pinned frame-by-frame,
 *  - golden fixtures under tests/golden/ can be generated,
 *  - bench.py can time the real reference as the CPU baseline.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "FLAC/stream_encoder.h"
#include "FLAC/stream_decoder.h"

/* hidden test hooks exported by the reference (stream_encoder.c:1419,1747-1778) */
extern FLAC__bool FLAC__stream_encoder_disable_constant_subframes(FLAC__StreamEncoder *encoder, FLAC__bool value);
extern FLAC__bool FLAC__stream_encoder_disable_fixed_subframes(FLAC__StreamEncoder *encoder, FLAC__bool value);
extern FLAC__bool FLAC__stream_encoder_disable_verbatim_subframes(FLAC__StreamEncoder *encoder, FLAC__bool value);
extern FLAC__bool FLAC__stream_encoder_set_do_md5(FLAC__StreamEncoder *encoder, FLAC__bool value);

typedef struct {
	uint32_t channels, bits_per_sample, sample_rate, blocksize;
	uint32_t do_mid_side, loose_mid_side;
	uint32_t max_lpc_order, qlp_coeff_precision, do_qlp_coeff_prec_search;
	uint32_t do_exhaustive_model_search;
	uint32_t min_residual_partition_order, max_residual_partition_order;
	uint32_t disable_constant, disable_fixed, disable_verbatim;
	uint32_t streamable_subset;
	const char *apodization; /* NULL = leave the reference default */
} refdrv_params;

typedef struct {
	uint8_t *out; size_t cap, len; int overflow;
	uint32_t *frame_bytes, *frame_samples; uint32_t max_frames, nframes;
	uint64_t audio_offset; int seen_audio;
	uint8_t md5[16]; uint32_t min_framesize, max_framesize; uint64_t total_samples;
} sink_t;

static FLAC__StreamEncoderWriteStatus write_cb(const FLAC__StreamEncoder *e, const FLAC__byte buffer[], size_t bytes, unsigned samples, unsigned current_frame, void *cd)
{
	sink_t *s = (sink_t*)cd;
	(void)e; (void)current_frame;
	if(samples > 0) {
		if(!s->seen_audio) { s->seen_audio = 1; s->audio_offset = s->len; }
		if(s->nframes < s->max_frames) {
			if(s->frame_bytes) s->frame_bytes[s->nframes] = (uint32_t)bytes;
			if(s->frame_samples) s->frame_samples[s->nframes] = samples;
		}
		s->nframes++;
	}
	if(s->len + bytes > s->cap) { s->overflow = 1; return FLAC__STREAM_ENCODER_WRITE_STATUS_FATAL_ERROR; }
	memcpy(s->out + s->len, buffer, bytes);
	s->len += bytes;
	return FLAC__STREAM_ENCODER_WRITE_STATUS_OK;
}

static void meta_cb(const FLAC__StreamEncoder *e, const FLAC__StreamMetadata *m, void *cd)
{
	sink_t *s = (sink_t*)cd;
	(void)e;
	if(m->type == FLAC__METADATA_TYPE_STREAMINFO) {
		memcpy(s->md5, m->data.stream_info.md5sum, 16);
		s->min_framesize = m->data.stream_info.min_framesize;
		s->max_framesize = m->data.stream_info.max_framesize;
		s->total_samples = m->data.stream_info.total_samples;
	}
}

/* Encode `nsamples` wide samples of interleaved int32 PCM, feeding the encoder `chunk`
 * wide samples per process_interleaved() call (0 = all at once).
 * out receives the whole stream ("fLaC" + metadata + frames, STREAMINFO NOT back-patched
 * because no seek callback is given); info[] = {audio_offset, nframes, min_framesize,
 * max_framesize, total_samples}; md5_out = STREAMINFO MD5 as delivered to the metadata
 * callback.  Returns 0 on success, else 100+init status / 200+encoder state / 1 overflow. */
int refdrv_encode(const refdrv_params *p, const int32_t *pcm, uint64_t nsamples, uint32_t chunk,
	uint8_t *out, size_t cap, size_t *outlen,
	uint32_t *frame_bytes, uint32_t *frame_samples, uint32_t max_frames,
	uint64_t info[5], uint8_t md5_out[16])
{
	FLAC__StreamEncoder *enc = FLAC__stream_encoder_new();
	FLAC__StreamEncoderInitStatus st;
	sink_t s;
	int rc = 0;
	uint64_t done = 0;
	if(!enc) return 2;
	memset(&s, 0, sizeof s);
	s.out = out; s.cap = cap; s.frame_bytes = frame_bytes; s.frame_samples = frame_samples; s.max_frames = max_frames;

	FLAC__stream_encoder_set_streamable_subset(enc, p->streamable_subset? true:false);
	FLAC__stream_encoder_set_channels(enc, p->channels);
	FLAC__stream_encoder_set_bits_per_sample(enc, p->bits_per_sample);
	FLAC__stream_encoder_set_sample_rate(enc, p->sample_rate);
	FLAC__stream_encoder_set_blocksize(enc, p->blocksize);
	FLAC__stream_encoder_set_do_mid_side_stereo(enc, p->do_mid_side? true:false);
	FLAC__stream_encoder_set_loose_mid_side_stereo(enc, p->loose_mid_side? true:false);
	if(p->apodization) FLAC__stream_encoder_set_apodization(enc, p->apodization);
	FLAC__stream_encoder_set_max_lpc_order(enc, p->max_lpc_order);
	FLAC__stream_encoder_set_qlp_coeff_precision(enc, p->qlp_coeff_precision);
	FLAC__stream_encoder_set_do_qlp_coeff_prec_search(enc, p->do_qlp_coeff_prec_search? true:false);
	FLAC__stream_encoder_set_do_exhaustive_model_search(enc, p->do_exhaustive_model_search? true:false);
	FLAC__stream_encoder_set_min_residual_partition_order(enc, p->min_residual_partition_order);
	FLAC__stream_encoder_set_max_residual_partition_order(enc, p->max_residual_partition_order);
	FLAC__stream_encoder_disable_constant_subframes(enc, p->disable_constant? true:false);
	FLAC__stream_encoder_disable_fixed_subframes(enc, p->disable_fixed? true:false);
	FLAC__stream_encoder_disable_verbatim_subframes(enc, p->disable_verbatim? true:false);

	st = FLAC__stream_encoder_init_stream(enc, write_cb, 0, 0, meta_cb, &s);
	if(st != FLAC__STREAM_ENCODER_INIT_STATUS_OK) { FLAC__stream_encoder_delete(enc); return 100 + (int)st; }

	if(chunk == 0) chunk = 1u << 20;
	while(done < nsamples) {
		uint64_t n = nsamples - done;
		if(n > chunk) n = chunk;
		if(!FLAC__stream_encoder_process_interleaved(enc, pcm + done * p->channels, (unsigned)n)) { rc = 200 + (int)FLAC__stream_encoder_get_state(enc); break; }
		done += n;
	}
	if(!FLAC__stream_encoder_finish(enc) && rc == 0) rc = 200 + (int)FLAC__stream_encoder_get_state(enc);
	FLAC__stream_encoder_delete(enc);
	if(s.overflow) rc = 1;
	*outlen = s.len;
	info[0] = s.audio_offset; info[1] = s.nframes; info[2] = s.min_framesize; info[3] = s.max_framesize; info[4] = s.total_samples;
	memcpy(md5_out, s.md5, 16);
	return rc;
}

/* ---- decoder side ---- */
typedef struct {
	const uint8_t *in; size_t len, pos;
	int32_t *pcm; uint64_t cap_samples, nsamples; uint32_t channels, bps, rate;
	int err; uint32_t nerrors;
	uint8_t md5[16]; uint64_t total_samples;
} src_t;

static FLAC__StreamDecoderReadStatus dread_cb(const FLAC__StreamDecoder *d, FLAC__byte buffer[], size_t *bytes, void *cd)
{
	src_t *s = (src_t*)cd; size_t n = s->len - s->pos;
	(void)d;
	if(n == 0) { *bytes = 0; return FLAC__STREAM_DECODER_READ_STATUS_END_OF_STREAM; }
	if(n > *bytes) n = *bytes;
	memcpy(buffer, s->in + s->pos, n); s->pos += n; *bytes = n;
	return FLAC__STREAM_DECODER_READ_STATUS_CONTINUE;
}
static FLAC__StreamDecoderWriteStatus dwrite_cb(const FLAC__StreamDecoder *d, const FLAC__Frame *f, const FLAC__int32 *const buf[], void *cd)
{
	src_t *s = (src_t*)cd; unsigned i, c; const unsigned ch = f->header.channels, n = f->header.blocksize;
	(void)d;
	s->channels = ch; s->bps = f->header.bits_per_sample; s->rate = f->header.sample_rate;
	if(s->nsamples + n > s->cap_samples) { s->err = 1; return FLAC__STREAM_DECODER_WRITE_STATUS_ABORT; }
	for(i = 0; i < n; i++) for(c = 0; c < ch; c++) s->pcm[(s->nsamples + i) * ch + c] = buf[c][i];
	s->nsamples += n;
	return FLAC__STREAM_DECODER_WRITE_STATUS_CONTINUE;
}
static void dmeta_cb(const FLAC__StreamDecoder *d, const FLAC__StreamMetadata *m, void *cd)
{
	src_t *s = (src_t*)cd; (void)d;
	if(m->type == FLAC__METADATA_TYPE_STREAMINFO) { memcpy(s->md5, m->data.stream_info.md5sum, 16); s->total_samples = m->data.stream_info.total_samples; }
}
static void derr_cb(const FLAC__StreamDecoder *d, FLAC__StreamDecoderErrorStatus st, void *cd)
{
	src_t *s = (src_t*)cd; (void)d; (void)st; s->nerrors++;
}

/* Decode a whole native FLAC stream held in memory into interleaved int32.
 * info[] = {nsamples, channels, bps, rate, error callbacks seen, md5 check passed (1/0)}.
 * Returns 0 on success. */
int refdrv_decode(const uint8_t *flac, size_t len, int32_t *pcm, uint64_t cap_samples, uint64_t info[6])
{
	FLAC__StreamDecoder *dec = FLAC__stream_decoder_new();
	src_t s; int rc = 0; FLAC__bool ok, fin;
	if(!dec) return 2;
	memset(&s, 0, sizeof s);
	s.in = flac; s.len = len; s.pcm = pcm; s.cap_samples = cap_samples;
	FLAC__stream_decoder_set_md5_checking(dec, true);
	if(FLAC__stream_decoder_init_stream(dec, dread_cb, 0, 0, 0, 0, dwrite_cb, dmeta_cb, derr_cb, &s) != FLAC__STREAM_DECODER_INIT_STATUS_OK) { FLAC__stream_decoder_delete(dec); return 3; }
	ok = FLAC__stream_decoder_process_until_end_of_stream(dec);
	fin = FLAC__stream_decoder_finish(dec); /* false = MD5 mismatch */
	if(!ok || s.err) rc = 4;
	FLAC__stream_decoder_delete(dec);
	info[0] = s.nsamples; info[1] = s.channels; info[2] = s.bps; info[3] = s.rate; info[4] = s.nerrors; info[5] = fin? 1:0;
	return rc;
}
