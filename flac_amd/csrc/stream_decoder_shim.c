/* Host side of the libFLAC-style decoder object (include/flac_amd_stream.h) in plain C.
 *
 * Keeps the reference's layer-3 semantics - states, init status codes, metadata search with
 * ID3v2 skipping, STREAMINFO parsing, one frame per process_single, write-callback arguments,
 * MD5 checking, finish / flush / reset / seek - and hands the frame work (frame_sync_,
 * read_frame_header_, read_subframe_*, read_residual_partitioned_rice_, restore_signal,
 * decorrelation, CRC checks) to the device decoder a batch of frames at a time.
 * Citations are to /root/reference/src/libFLAC/stream_decoder.c unless noted.
 */
#define _FILE_OFFSET_BITS 64
#include "../../include/flac_amd_stream.h"
#include "../../include/flac_amd_engine.h"
#include "host_md5.h"
#include "metadata_blocks.h"
#include <stdlib.h>
#include <sys/types.h>
#include <sys/stat.h>

const char * const FLAC__StreamDecoderStateString[] = { /* :108-119 */
	"FLAC__STREAM_DECODER_SEARCH_FOR_METADATA", "FLAC__STREAM_DECODER_READ_METADATA", "FLAC__STREAM_DECODER_SEARCH_FOR_FRAME_SYNC",
	"FLAC__STREAM_DECODER_READ_FRAME", "FLAC__STREAM_DECODER_END_OF_STREAM", "FLAC__STREAM_DECODER_OGG_ERROR", "FLAC__STREAM_DECODER_SEEK_ERROR",
	"FLAC__STREAM_DECODER_ABORTED", "FLAC__STREAM_DECODER_MEMORY_ALLOCATION_ERROR", "FLAC__STREAM_DECODER_UNINITIALIZED" };
const char * const FLAC__StreamDecoderInitStatusString[] = { /* :121-128 */
	"FLAC__STREAM_DECODER_INIT_STATUS_OK", "FLAC__STREAM_DECODER_INIT_STATUS_UNSUPPORTED_CONTAINER", "FLAC__STREAM_DECODER_INIT_STATUS_INVALID_CALLBACKS",
	"FLAC__STREAM_DECODER_INIT_STATUS_MEMORY_ALLOCATION_ERROR", "FLAC__STREAM_DECODER_INIT_STATUS_ERROR_OPENING_FILE", "FLAC__STREAM_DECODER_INIT_STATUS_ALREADY_INITIALIZED" };
const char * const FLAC__StreamDecoderErrorStatusString[] = { /* :162-167 */
	"FLAC__STREAM_DECODER_ERROR_STATUS_LOST_SYNC", "FLAC__STREAM_DECODER_ERROR_STATUS_BAD_HEADER", "FLAC__STREAM_DECODER_ERROR_STATUS_FRAME_CRC_MISMATCH",
	"FLAC__STREAM_DECODER_ERROR_STATUS_UNPARSEABLE_STREAM" };

const char * const FLAC__SubframeTypeString[] = { "CONSTANT", "VERBATIM", "FIXED", "LPC" };                       /* format.c:160-165 */
const char * const FLAC__EntropyCodingMethodTypeString[] = { "PARTITIONED_RICE", "PARTITIONED_RICE2" };             /* format.c:143-146 */
const char * const FLAC__ChannelAssignmentString[] = { "INDEPENDENT", "LEFT_SIDE", "RIGHT_SIDE", "MID_SIDE" };      /* format.c:167-172 */

struct FLAC__StreamDecoder {
	FLAC__StreamDecoderState state;
	FLAC__bool md5_checking, do_md5;
	FLAC__bool respond[128];                 /* metadata_filter (:148-151): block types handed to the metadata callback */
	FLAC__byte *app_ids; size_t app_ids_count;  /* exceptions to respond[APPLICATION], 4 bytes each (:780-865) */
	int device; unsigned batch_frames;
	FLAC__StreamDecoderReadCallback read_cb; FLAC__StreamDecoderSeekCallback seek_cb; FLAC__StreamDecoderTellCallback tell_cb;
	FLAC__StreamDecoderLengthCallback length_cb; FLAC__StreamDecoderEofCallback eof_cb; FLAC__StreamDecoderWriteCallback write_cb;
	FLAC__StreamDecoderMetadataCallback metadata_cb; FLAC__StreamDecoderErrorCallback error_cb; void *client_data;
	FILE *file;
	FLAC__bool has_streaminfo; FLAC__StreamMetadata streaminfo;
	FLAC__uint64 samples_decoded, first_frame_offset;
	/* input window: in[in_pos, in_len) is unread, in[0] sits at absolute stream offset in_base */
	uint8_t *in; size_t in_cap, in_len, in_pos; FLAC__uint64 in_base; int eof, warned_sync;
	/* device decoder and the batch it produced */
	flacamd_decoder *eng; size_t batch_bytes, frame_bound; unsigned eng_frames;
	int32_t *pcm; size_t pcm_cap;
	/* the batch as a queue of events in stream order: frames, and on damaged input also zeroed frames, error
	 * reports and the end of the stream (f_kind); f_sn is the sample number from the frame header where the
	 * walk over damaged input took it, else all ones = the running count */
	uint64_t *f_off, *f_num, *f_end, *f_sn; uint32_t *f_bs, *f_asg; uint8_t *f_kind; unsigned q_n, q_i, ev_cap; FLAC__uint64 batch_base;
	uint64_t *c_pos, *c_end, *c_num, *c_off; uint32_t *c_info, *c_bs;   /* candidate table of a damaged batch */
	/* what the device read of every subframe of the batch (by candidate), and per queued event its candidate; the
	 * per-channel parameter arrays FLAC__Frame.subframes[] point at (include/FLAC/format.h:213-240) */
	flacamd_subframe_info *sub; uint32_t *f_cand, *frame_cand; size_t batch_in_off;
	FLAC__EntropyCodingMethod_PartitionedRiceContents prc[8]; unsigned prc_alloc[8];
	/* frames whose header is valid but differs from STREAMINFO (channels, sample size, a block longer than the maximum):
	 * the reference decodes them with the frame's own parameters (:2002-2145); here a second device decoder built for
	 * those parameters takes the frame, and the event carries its samples */
	flacamd_decoder *alt_eng; unsigned alt_ch, alt_bps, alt_bs; size_t alt_bytes;
	struct alt_frame { int32_t *pcm, *res; flacamd_subframe_info sub[8]; unsigned ch, bps; } **f_alt;
	int32_t *res; int want_res;   /* the batch's residuals as the stream carried them (FLAC__Subframe_*::residual), page-locked like pcm */
	size_t planar_cap, pcm_bytes;
	int search_warned;               /* LOST_SYNC already reported for the sync search in progress (frame_sync_'s `first`) */
	FLAC__uint64 decode_pos;          /* absolute offset just past the last frame handed out */
	int32_t *planar; uint8_t *md5_scratch;
	md5_state md5;
	FLAC__FrameHeader last;
	FLAC__bool is_seeking; FLAC__uint64 target_sample;
	/* the stream's first SEEKTABLE, kept whatever the metadata filter says (the reference keeps it for seeking too, :1623-1650) */
	FLAC__StreamMetadata_SeekPoint *seek_points; unsigned num_seek_points;
	int resync_base;   /* after flush(): the client may have repositioned its input, ask tell_cb where the next read starts */
	int want_grow;     /* a frame longer than the input window: make the window larger and try again */
};

static void dec_defaults(FLAC__StreamDecoder *d) /* set_defaults_, :1247-1268 */
{
	const int device = d->device, want_res = d->want_res; const unsigned batch = d->batch_frames;
	free(d->app_ids);
	memset(d, 0, sizeof *d);
	d->device = device; d->batch_frames = batch ? batch : 2048u; d->want_res = want_res;
	d->state = FLAC__STREAM_DECODER_UNINITIALIZED;
	d->respond[FLAC__METADATA_TYPE_STREAMINFO] = 1; d->md5_checking = 0;
}

FLAC__StreamDecoder *FLAC__stream_decoder_new(void)
{
	FLAC__StreamDecoder *d = (FLAC__StreamDecoder*)calloc(1, sizeof *d);
	if(d) {
		const char *env = getenv("FLACAMD_DECODER_RESIDUALS");   /* for clients that cannot call the setter: "0" turns the residual pass off */
		dec_defaults(d);
		d->want_res = !(env && env[0] == '0' && env[1] == 0);
	}
	return d;
}

static void alt_free(struct alt_frame *a) { if(a) { free(a->pcm); free(a->res); free(a); } }

static void dec_free(FLAC__StreamDecoder *d)
{
	if(d->eng) { flacamd_decoder_destroy(d->eng); d->eng = 0; }
	flacamd_host_free(d->in, d->in_cap); d->in = 0; flacamd_host_free(d->pcm, d->pcm_bytes); d->pcm = 0; flacamd_host_free(d->res, d->pcm_bytes); d->res = 0; free(d->f_off); d->f_off = 0; free(d->f_num); d->f_num = 0; free(d->f_end); d->f_end = 0;
	free(d->f_bs); d->f_bs = 0; free(d->f_asg); d->f_asg = 0; free(d->planar); d->planar = 0; free(d->md5_scratch); d->md5_scratch = 0;
	free(d->f_sn); d->f_sn = 0; free(d->f_kind); d->f_kind = 0; free(d->c_pos); d->c_pos = 0; free(d->c_end); d->c_end = 0; free(d->c_num); d->c_num = 0;
	free(d->c_off); d->c_off = 0; free(d->c_info); d->c_info = 0; free(d->c_bs); d->c_bs = 0;
	free(d->seek_points); d->seek_points = 0; d->num_seek_points = 0;
	free(d->sub); d->sub = 0; free(d->f_cand); d->f_cand = 0; free(d->frame_cand); d->frame_cand = 0;
	if(d->f_alt) { unsigned r; for(r = 0; r < d->ev_cap; r++) if(d->f_alt[r]) { free(d->f_alt[r]->pcm); free(d->f_alt[r]); } free(d->f_alt); d->f_alt = 0; }
	if(d->alt_eng) { flacamd_decoder_destroy(d->alt_eng); d->alt_eng = 0; }
	{ unsigned c; for(c = 0; c < 8; c++) { free(d->prc[c].parameters); free(d->prc[c].raw_bits); d->prc[c].parameters = d->prc[c].raw_bits = 0; d->prc_alloc[c] = 0; } }
}

void FLAC__stream_decoder_delete(FLAC__StreamDecoder *d)
{
	if(!d) return;
	(void)FLAC__stream_decoder_finish(d);
	if(d->file && d->file != stdin) fclose(d->file);   /* left by an init_ogg_* call that was refused */
	free(d->app_ids);
	free(d);
}

#define DEC_SETTER(name, type, stmt) FLAC__bool name(FLAC__StreamDecoder *d, type value) { if(!d || d->state != FLAC__STREAM_DECODER_UNINITIALIZED) return 0; stmt; return 1; }
DEC_SETTER(FLAC__stream_decoder_set_md5_checking, FLAC__bool, d->md5_checking = value)
DEC_SETTER(flacamd_stream_decoder_set_device, int, if(value < 0) return 0; d->device = value)
DEC_SETTER(flacamd_stream_decoder_set_residuals, FLAC__bool, d->want_res = value ? 1 : 0)
DEC_SETTER(flacamd_stream_decoder_set_batch_frames, unsigned, if(value == 0) return 0; d->batch_frames = value)
/* the metadata filter (:762-877) */
#define UNINIT(d) ((d) && (d)->state == FLAC__STREAM_DECODER_UNINITIALIZED)
FLAC__bool FLAC__stream_decoder_set_metadata_respond(FLAC__StreamDecoder *d, int type)
{
	if(!UNINIT(d) || (unsigned)type > 126u) return 0;
	d->respond[type] = 1;
	if(type == FLAC__METADATA_TYPE_APPLICATION) d->app_ids_count = 0;
	return 1;
}
FLAC__bool FLAC__stream_decoder_set_metadata_ignore(FLAC__StreamDecoder *d, int type)
{
	if(!UNINIT(d) || (unsigned)type > 126u) return 0;
	d->respond[type] = 0;
	if(type == FLAC__METADATA_TYPE_APPLICATION) d->app_ids_count = 0;
	return 1;
}
static FLAC__bool add_app_id(FLAC__StreamDecoder *d, const FLAC__byte id[4])
{
	FLAC__byte *n = (FLAC__byte*)realloc(d->app_ids, 4 * (d->app_ids_count + 1));
	if(!n) { d->state = FLAC__STREAM_DECODER_MEMORY_ALLOCATION_ERROR; return 0; }
	d->app_ids = n; memcpy(n + 4 * d->app_ids_count, id, 4); d->app_ids_count++;
	return 1;
}
/* an id listed while APPLICATION blocks are ignored is an exception that is delivered, and the other way round */
FLAC__bool FLAC__stream_decoder_set_metadata_respond_application(FLAC__StreamDecoder *d, const FLAC__byte id[4])
{
	if(!UNINIT(d) || !id) return 0;
	return d->respond[FLAC__METADATA_TYPE_APPLICATION] ? 1 : add_app_id(d, id);
}
FLAC__bool FLAC__stream_decoder_set_metadata_ignore_application(FLAC__StreamDecoder *d, const FLAC__byte id[4])
{
	if(!UNINIT(d) || !id) return 0;
	return !d->respond[FLAC__METADATA_TYPE_APPLICATION] ? 1 : add_app_id(d, id);
}
FLAC__bool FLAC__stream_decoder_set_metadata_respond_all(FLAC__StreamDecoder *d)
{
	unsigned i;
	if(!UNINIT(d)) return 0;
	for(i = 0; i < 128; i++) d->respond[i] = 1;
	d->app_ids_count = 0;
	return 1;
}
FLAC__bool FLAC__stream_decoder_set_metadata_ignore_all(FLAC__StreamDecoder *d) { if(!UNINIT(d)) return 0; memset(d->respond, 0, sizeof d->respond); d->app_ids_count = 0; return 1; }

FLAC__StreamDecoderState FLAC__stream_decoder_get_state(const FLAC__StreamDecoder *d) { return d->state; }
const char *FLAC__stream_decoder_get_resolved_state_string(const FLAC__StreamDecoder *d) { return FLAC__StreamDecoderStateString[d->state]; }
FLAC__bool FLAC__stream_decoder_get_md5_checking(const FLAC__StreamDecoder *d) { return d->md5_checking; }
FLAC__uint64 FLAC__stream_decoder_get_total_samples(const FLAC__StreamDecoder *d) { return d->has_streaminfo ? d->streaminfo.data.stream_info.total_samples : 0; }
unsigned FLAC__stream_decoder_get_channels(const FLAC__StreamDecoder *d) { return d->last.channels; }
FLAC__ChannelAssignment FLAC__stream_decoder_get_channel_assignment(const FLAC__StreamDecoder *d) { return d->last.channel_assignment; }
unsigned FLAC__stream_decoder_get_bits_per_sample(const FLAC__StreamDecoder *d) { return d->last.bits_per_sample; }
unsigned FLAC__stream_decoder_get_sample_rate(const FLAC__StreamDecoder *d) { return d->last.sample_rate; }
unsigned FLAC__stream_decoder_get_blocksize(const FLAC__StreamDecoder *d) { return d->last.blocksize; }

FLAC__bool FLAC__stream_decoder_get_decode_position(const FLAC__StreamDecoder *d, FLAC__uint64 *position) /* :930-950 */
{
	if(!d || !position || !d->tell_cb) return 0;
	*position = d->q_i < d->q_n || d->decode_pos ? d->decode_pos : d->in_base + d->in_pos;
	return 1;
}

/* file callbacks (:3315-3390) */
static FLAC__StreamDecoderReadStatus file_read(const FLAC__StreamDecoder *d, FLAC__byte buffer[], size_t *bytes, void *cd)
{
	(void)cd;
	if(*bytes == 0) return FLAC__STREAM_DECODER_READ_STATUS_ABORT;
	*bytes = fread(buffer, 1, *bytes, d->file);
	if(ferror(d->file)) return FLAC__STREAM_DECODER_READ_STATUS_ABORT;
	return *bytes == 0 ? FLAC__STREAM_DECODER_READ_STATUS_END_OF_STREAM : FLAC__STREAM_DECODER_READ_STATUS_CONTINUE;
}
static FLAC__StreamDecoderSeekStatus file_seek(const FLAC__StreamDecoder *d, FLAC__uint64 off, void *cd)
{
	(void)cd;
	if(d->file == stdin) return FLAC__STREAM_DECODER_SEEK_STATUS_UNSUPPORTED;
	return fseeko(d->file, (off_t)off, SEEK_SET) < 0 ? FLAC__STREAM_DECODER_SEEK_STATUS_ERROR : FLAC__STREAM_DECODER_SEEK_STATUS_OK;
}
static FLAC__StreamDecoderTellStatus file_tell(const FLAC__StreamDecoder *d, FLAC__uint64 *off, void *cd)
{
	off_t o; (void)cd;
	if(d->file == stdin) return FLAC__STREAM_DECODER_TELL_STATUS_UNSUPPORTED;
	if((o = ftello(d->file)) < 0) return FLAC__STREAM_DECODER_TELL_STATUS_ERROR;
	*off = (FLAC__uint64)o; return FLAC__STREAM_DECODER_TELL_STATUS_OK;
}
static FLAC__StreamDecoderLengthStatus file_length(const FLAC__StreamDecoder *d, FLAC__uint64 *len, void *cd)
{
	struct stat st; (void)cd;
	if(d->file == stdin) return FLAC__STREAM_DECODER_LENGTH_STATUS_UNSUPPORTED;
	if(fstat(fileno(d->file), &st) != 0) return FLAC__STREAM_DECODER_LENGTH_STATUS_ERROR;
	*len = (FLAC__uint64)st.st_size; return FLAC__STREAM_DECODER_LENGTH_STATUS_OK;
}
static FLAC__bool file_eof(const FLAC__StreamDecoder *d, void *cd) { (void)cd; return feof(d->file) ? 1 : 0; }

static FLAC__StreamDecoderInitStatus dec_init(FLAC__StreamDecoder *d, FLAC__StreamDecoderReadCallback rcb, FLAC__StreamDecoderSeekCallback scb,
	FLAC__StreamDecoderTellCallback tcb, FLAC__StreamDecoderLengthCallback lcb, FLAC__StreamDecoderEofCallback ecb, FLAC__StreamDecoderWriteCallback wcb,
	FLAC__StreamDecoderMetadataCallback mcb, FLAC__StreamDecoderErrorCallback errcb, void *client_data)
{
	/* init_stream_internal_ (:364-470) */
	if(d->state != FLAC__STREAM_DECODER_UNINITIALIZED) return FLAC__STREAM_DECODER_INIT_STATUS_ALREADY_INITIALIZED;
	if(!rcb || !wcb || !errcb || (scb && (!tcb || !lcb || !ecb))) return FLAC__STREAM_DECODER_INIT_STATUS_INVALID_CALLBACKS;
	if(flacamd_device_count() <= d->device) return FLAC__STREAM_DECODER_INIT_STATUS_MEMORY_ALLOCATION_ERROR; /* no device to decode on: there is no CPU path */
	d->read_cb = rcb; d->seek_cb = scb; d->tell_cb = tcb; d->length_cb = lcb; d->eof_cb = ecb; d->write_cb = wcb; d->metadata_cb = mcb; d->error_cb = errcb; d->client_data = client_data;
	d->in_cap = (size_t)1 << 20; d->in = (uint8_t*)flacamd_host_alloc(d->in_cap);   /* page-locked: the window goes to the device at bus speed */
	d->md5_scratch = (uint8_t*)malloc(65536);
	if(!d->in || !d->md5_scratch) { dec_free(d); return FLAC__STREAM_DECODER_INIT_STATUS_MEMORY_ALLOCATION_ERROR; }
	d->in_len = d->in_pos = 0; d->in_base = 0; d->eof = 0; d->warned_sync = 0;
	d->has_streaminfo = 0; d->samples_decoded = 0; d->first_frame_offset = 0; d->q_n = d->q_i = 0; d->decode_pos = 0;
	d->do_md5 = d->md5_checking; d->is_seeking = 0;
	md5_init(&d->md5);
	d->state = FLAC__STREAM_DECODER_SEARCH_FOR_METADATA;
	return FLAC__STREAM_DECODER_INIT_STATUS_OK;
}

FLAC__StreamDecoderInitStatus FLAC__stream_decoder_init_stream(FLAC__StreamDecoder *d, FLAC__StreamDecoderReadCallback rcb, FLAC__StreamDecoderSeekCallback scb,
	FLAC__StreamDecoderTellCallback tcb, FLAC__StreamDecoderLengthCallback lcb, FLAC__StreamDecoderEofCallback ecb, FLAC__StreamDecoderWriteCallback wcb,
	FLAC__StreamDecoderMetadataCallback mcb, FLAC__StreamDecoderErrorCallback errcb, void *client_data)
{
	return dec_init(d, rcb, scb, tcb, lcb, ecb, wcb, mcb, errcb, client_data);
}

FLAC__StreamDecoderInitStatus FLAC__stream_decoder_init_FILE(FLAC__StreamDecoder *d, FILE *file, FLAC__StreamDecoderWriteCallback wcb,
	FLAC__StreamDecoderMetadataCallback mcb, FLAC__StreamDecoderErrorCallback errcb, void *client_data)
{
	FLAC__StreamDecoderInitStatus st;
	if(d->state != FLAC__STREAM_DECODER_UNINITIALIZED) return FLAC__STREAM_DECODER_INIT_STATUS_ALREADY_INITIALIZED;
	if(!wcb || !errcb) return FLAC__STREAM_DECODER_INIT_STATUS_INVALID_CALLBACKS;
	d->file = file;
	st = dec_init(d, file_read, file == stdin ? 0 : file_seek, file == stdin ? 0 : file_tell, file == stdin ? 0 : file_length, file_eof, wcb, mcb, errcb, client_data);
	if(st != FLAC__STREAM_DECODER_INIT_STATUS_OK) d->file = 0;
	return st;
}

FLAC__StreamDecoderInitStatus FLAC__stream_decoder_init_file(FLAC__StreamDecoder *d, const char *filename, FLAC__StreamDecoderWriteCallback wcb,
	FLAC__StreamDecoderMetadataCallback mcb, FLAC__StreamDecoderErrorCallback errcb, void *client_data)
{
	FILE *f; FLAC__StreamDecoderInitStatus st;
	if(d->state != FLAC__STREAM_DECODER_UNINITIALIZED) return FLAC__STREAM_DECODER_INIT_STATUS_ALREADY_INITIALIZED;
	if(!wcb || !errcb) return FLAC__STREAM_DECODER_INIT_STATUS_INVALID_CALLBACKS;
	f = filename ? fopen(filename, "rb") : stdin;
	if(!f) return FLAC__STREAM_DECODER_INIT_STATUS_ERROR_OPENING_FILE;
	st = FLAC__stream_decoder_init_FILE(d, f, wcb, mcb, errcb, client_data);
	if(st != FLAC__STREAM_DECODER_INIT_STATUS_OK && f != stdin) fclose(f);
	return st;
}

/* The Ogg inits as a library built without Ogg transport has them (:517-700 over the `#if !FLAC__HAS_OGG` exit of
 * init_stream_internal_, :394-399): the checks ahead of that exit, then UNSUPPORTED_CONTAINER.  A file that
 * init_ogg_file opened stays with the object, as in the reference, and is closed by finish() or delete(). */
FLAC__StreamDecoderInitStatus FLAC__stream_decoder_init_ogg_stream(FLAC__StreamDecoder *d, FLAC__StreamDecoderReadCallback rcb, FLAC__StreamDecoderSeekCallback scb,
	FLAC__StreamDecoderTellCallback tcb, FLAC__StreamDecoderLengthCallback lcb, FLAC__StreamDecoderEofCallback ecb, FLAC__StreamDecoderWriteCallback wcb,
	FLAC__StreamDecoderMetadataCallback mcb, FLAC__StreamDecoderErrorCallback errcb, void *client_data)
{
	(void)rcb; (void)scb; (void)tcb; (void)lcb; (void)ecb; (void)wcb; (void)mcb; (void)errcb; (void)client_data;
	if(d->state != FLAC__STREAM_DECODER_UNINITIALIZED) return FLAC__STREAM_DECODER_INIT_STATUS_ALREADY_INITIALIZED;
	return FLAC__STREAM_DECODER_INIT_STATUS_UNSUPPORTED_CONTAINER;
}

FLAC__StreamDecoderInitStatus FLAC__stream_decoder_init_ogg_FILE(FLAC__StreamDecoder *d, FILE *file, FLAC__StreamDecoderWriteCallback wcb,
	FLAC__StreamDecoderMetadataCallback mcb, FLAC__StreamDecoderErrorCallback errcb, void *client_data)
{
	(void)mcb; (void)client_data;
	if(d->state != FLAC__STREAM_DECODER_UNINITIALIZED) return FLAC__STREAM_DECODER_INIT_STATUS_ALREADY_INITIALIZED;
	if(!wcb || !errcb) return FLAC__STREAM_DECODER_INIT_STATUS_INVALID_CALLBACKS;
	d->file = file;
	return FLAC__STREAM_DECODER_INIT_STATUS_UNSUPPORTED_CONTAINER;
}

FLAC__StreamDecoderInitStatus FLAC__stream_decoder_init_ogg_file(FLAC__StreamDecoder *d, const char *filename, FLAC__StreamDecoderWriteCallback wcb,
	FLAC__StreamDecoderMetadataCallback mcb, FLAC__StreamDecoderErrorCallback errcb, void *client_data)
{
	FILE *f;
	if(d->state != FLAC__STREAM_DECODER_UNINITIALIZED) return FLAC__STREAM_DECODER_INIT_STATUS_ALREADY_INITIALIZED;
	if(!wcb || !errcb) return FLAC__STREAM_DECODER_INIT_STATUS_INVALID_CALLBACKS;
	f = filename ? fopen(filename, "rb") : stdin;
	if(!f) return FLAC__STREAM_DECODER_INIT_STATUS_ERROR_OPENING_FILE;
	return FLAC__stream_decoder_init_ogg_FILE(d, f, wcb, mcb, errcb, client_data);
}

/* :736-751 without libogg */
FLAC__bool FLAC__stream_decoder_set_ogg_serial_number(FLAC__StreamDecoder *d, long value) { (void)d; (void)value; return 0; }

/* ---- input: make at least `need` unread bytes available (fewer at end of stream) */
static FLAC__bool fill_range(FLAC__StreamDecoder *d, size_t least, size_t want);
static FLAC__bool fill(FLAC__StreamDecoder *d, size_t need) { return fill_range(d, need, need); }

/* at least `least` unread bytes (fewer at end of stream), reading towards `want`: a source that delivers little per
 * call (a pipe, a live stream) is not waited on for a whole window, a file fills it in one call */
static FLAC__bool fill_range(FLAC__StreamDecoder *d, size_t least, size_t need)
{
	if(d->in_len - d->in_pos >= least) return 1;
	if(d->resync_base && d->in_len == d->in_pos) {
		FLAC__uint64 at = 0;
		if(d->tell_cb && d->tell_cb(d, &at, d->client_data) == FLAC__STREAM_DECODER_TELL_STATUS_OK) { d->in_base = at; d->in_len = d->in_pos = 0; }
		d->resync_base = 0;
	}
	if(d->in_pos > 0) { /* compact */
		memmove(d->in, d->in + d->in_pos, d->in_len - d->in_pos);
		d->in_base += d->in_pos; d->in_len -= d->in_pos; d->in_pos = 0;
	}
	if(need > d->in_cap) {
		uint8_t *nb = (uint8_t*)flacamd_host_alloc(need);
		if(!nb) { d->state = FLAC__STREAM_DECODER_MEMORY_ALLOCATION_ERROR; return 0; }
		memcpy(nb, d->in, d->in_len);
		flacamd_host_free(d->in, d->in_cap);
		d->in = nb; d->in_cap = need;
	}
	while(d->in_len < least && !d->eof) {
		size_t n = d->in_cap - d->in_len;
		const FLAC__StreamDecoderReadStatus st = d->read_cb(d, d->in + d->in_len, &n, d->client_data);
		if(st == FLAC__STREAM_DECODER_READ_STATUS_ABORT) { d->state = FLAC__STREAM_DECODER_ABORTED; return 0; }
		d->in_len += n;
		if(st == FLAC__STREAM_DECODER_READ_STATUS_END_OF_STREAM || n == 0) d->eof = 1;
	}
	return 1;
}
static size_t avail(const FLAC__StreamDecoder *d) { return d->in_len - d->in_pos; }

/* skip n bytes of input that may be larger than the window */
static FLAC__bool skip_bytes(FLAC__StreamDecoder *d, FLAC__uint64 n)
{
	while(n > 0) {
		size_t k;
		if(!fill(d, 1)) return 0;
		if(avail(d) == 0) return 1;
		k = avail(d) < n ? avail(d) : (size_t)n;
		d->in_pos += k; n -= k;
	}
	return 1;
}

/* find_metadata_ (:1484-1572): skip an ID3v2 tag, find "fLaC" */
static FLAC__bool find_metadata(FLAC__StreamDecoder *d)
{
	for(;;) {
		const uint8_t *p;
		if(!fill(d, 10)) return 0;
		if(avail(d) < 4) { d->state = FLAC__STREAM_DECODER_END_OF_STREAM; return 1; }
		p = d->in + d->in_pos;
		if(p[0] == 'f' && p[1] == 'L' && p[2] == 'a' && p[3] == 'C') { d->in_pos += 4; d->state = FLAC__STREAM_DECODER_READ_METADATA; return 1; }
		if(p[0] == 'I' && p[1] == 'D' && p[2] == '3' && avail(d) >= 10) {
			const FLAC__uint64 skip = 10u + (((FLAC__uint64)(p[6] & 0x7f) << 21) | ((FLAC__uint64)(p[7] & 0x7f) << 14) | ((FLAC__uint64)(p[8] & 0x7f) << 7) | (FLAC__uint64)(p[9] & 0x7f));
			if(!skip_bytes(d, skip)) return 0;
			continue;
		}
		if(p[0] == 0xff && (p[1] & 0xfe) == 0xf8) { /* a frame before any STREAMINFO: this decoder needs the stream parameters first */
			d->error_cb(d, FLAC__STREAM_DECODER_ERROR_STATUS_UNPARSEABLE_STREAM, d->client_data);
			d->state = FLAC__STREAM_DECODER_ABORTED;
			return 0;
		}
		if(!d->warned_sync) { d->error_cb(d, FLAC__STREAM_DECODER_ERROR_STATUS_LOST_SYNC, d->client_data); d->warned_sync = 1; }
		d->in_pos++;
	}
}

static FLAC__bool make_engine(FLAC__StreamDecoder *d);
static FLAC__bool start_frames(FLAC__StreamDecoder *d)
{
	const FLAC__StreamMetadata_StreamInfo *si = &d->streaminfo.data.stream_info;
	size_t max_frame;
	if(!d->has_streaminfo || si->channels == 0 || si->max_blocksize == 0) {
		d->error_cb(d, FLAC__STREAM_DECODER_ERROR_STATUS_UNPARSEABLE_STREAM, d->client_data);
		d->state = FLAC__STREAM_DECODER_ABORTED; return 0;
	}
	d->first_frame_offset = d->in_base + d->in_pos;
	d->decode_pos = d->first_frame_offset;
	if(!d->eng) {
		max_frame = 16 + (size_t)si->channels * (((size_t)si->max_blocksize * (si->bits_per_sample + 2) + 7) / 8 + 64) + 2;
		d->frame_bound = max_frame;
		/* STREAMINFO's max_framesize is advisory (the reference never sizes anything by it) and comes from the file: it may
		 * make the window larger, within limits - a frame that still does not fit makes the window grow when it is met */
		if(si->max_framesize + 16u > max_frame && si->max_framesize <= ((size_t)1 << 20)) max_frame = (size_t)si->max_framesize + 16u;
		d->batch_bytes = max_frame * d->batch_frames;
		if(d->batch_bytes > ((size_t)256 << 20)) d->batch_bytes = (size_t)256 << 20;
		if(d->batch_bytes < 2 * max_frame) d->batch_bytes = 2 * max_frame;
		if(d->batch_bytes < ((size_t)1 << 16)) d->batch_bytes = (size_t)1 << 16;
		/* the device decoder's capacity counts every candidate that passes the header checks, and a
		 * window of batch_bytes may hold more (shorter) frames than batch_frames: leave room, and let
		 * decode_batch shrink its window when even that is exceeded */
		d->eng_frames = 2u * d->batch_frames + 256u;
		d->pcm_cap = (size_t)d->eng_frames * si->max_blocksize;
		if(!make_engine(d)) return 0;
		d->ev_cap = 4u * d->eng_frames + 64u;
		d->sub = (flacamd_subframe_info*)malloc(sizeof(flacamd_subframe_info) * (size_t)d->eng_frames * si->channels);
		d->f_cand = (uint32_t*)malloc(4u * d->ev_cap); d->frame_cand = (uint32_t*)malloc(4u * d->eng_frames);
		d->f_alt = (struct alt_frame**)calloc(d->ev_cap, sizeof(struct alt_frame*));
		d->planar_cap = (size_t)si->max_blocksize * si->channels;
		if(!d->sub || !d->f_cand || !d->frame_cand || !d->f_alt) { d->state = FLAC__STREAM_DECODER_MEMORY_ALLOCATION_ERROR; return 0; }
		d->pcm_bytes = sizeof(int32_t) * d->pcm_cap * si->channels;
		d->pcm = (int32_t*)flacamd_host_alloc(d->pcm_bytes);   /* page-locked: a batch's samples come back at bus speed */
		if(d->want_res && !(d->res = (int32_t*)flacamd_host_alloc(d->pcm_bytes))) { d->state = FLAC__STREAM_DECODER_MEMORY_ALLOCATION_ERROR; return 0; }
		d->f_off = (uint64_t*)malloc(8u * d->ev_cap); d->f_num = (uint64_t*)malloc(8u * d->ev_cap); d->f_end = (uint64_t*)malloc(8u * d->ev_cap); d->f_sn = (uint64_t*)malloc(8u * d->ev_cap);
		d->f_bs = (uint32_t*)malloc(4u * d->ev_cap); d->f_asg = (uint32_t*)malloc(4u * d->ev_cap); d->f_kind = (uint8_t*)malloc(d->ev_cap);
		d->c_pos = (uint64_t*)malloc(8u * d->eng_frames); d->c_end = (uint64_t*)malloc(8u * d->eng_frames); d->c_num = (uint64_t*)malloc(8u * d->eng_frames); d->c_off = (uint64_t*)malloc(8u * d->eng_frames);
		d->c_info = (uint32_t*)malloc(4u * d->eng_frames); d->c_bs = (uint32_t*)malloc(4u * d->eng_frames);
		d->planar = (int32_t*)calloc((size_t)si->max_blocksize * si->channels, sizeof(int32_t));
		if(!d->pcm || !d->f_off || !d->f_num || !d->f_end || !d->f_sn || !d->f_bs || !d->f_asg || !d->f_kind || !d->planar
			|| !d->c_pos || !d->c_end || !d->c_num || !d->c_off || !d->c_info || !d->c_bs) { d->state = FLAC__STREAM_DECODER_MEMORY_ALLOCATION_ERROR; return 0; }
	}
	d->state = FLAC__STREAM_DECODER_SEARCH_FOR_FRAME_SYNC;
	return 1;
}

/* read_metadata_ (:1574-1790): one block per call */
static FLAC__bool read_metadata(FLAC__StreamDecoder *d)
{
	const uint8_t *p; unsigned type, is_last; FLAC__uint64 length;
	if(!fill(d, 4)) return 0;
	if(avail(d) < 4) { d->state = FLAC__STREAM_DECODER_END_OF_STREAM; return 1; }
	p = d->in + d->in_pos;
	is_last = p[0] >> 7; type = p[0] & 0x7f; length = ((FLAC__uint64)p[1] << 16) | ((FLAC__uint64)p[2] << 8) | p[3];
	d->in_pos += 4;
	if(type == 0 && length < 34) { /* not a STREAMINFO this decoder can use: passed over, never handed out as one */
		d->error_cb(d, FLAC__STREAM_DECODER_ERROR_STATUS_BAD_HEADER, d->client_data);
		if(!skip_bytes(d, length)) return 0;
	}
	else if(type == 0) { /* stream_decoder.c:1792-1869, layout format.c:68-86 */
		FLAC__StreamMetadata_StreamInfo *si = &d->streaminfo.data.stream_info;
		FLAC__uint64 x;
		if(!fill(d, 34)) return 0;
		if(avail(d) < 34) { d->state = FLAC__STREAM_DECODER_END_OF_STREAM; return 1; }
		p = d->in + d->in_pos;
		d->streaminfo.type = 0; d->streaminfo.is_last = (FLAC__bool)is_last; d->streaminfo.length = (unsigned)length;
		si->min_blocksize = ((unsigned)p[0] << 8) | p[1]; si->max_blocksize = ((unsigned)p[2] << 8) | p[3];
		si->min_framesize = ((unsigned)p[4] << 16) | ((unsigned)p[5] << 8) | p[6]; si->max_framesize = ((unsigned)p[7] << 16) | ((unsigned)p[8] << 8) | p[9];
		x = 0; { int i; for(i = 0; i < 8; i++) x = (x << 8) | p[10 + i]; }
		si->sample_rate = (unsigned)(x >> 44); si->channels = (unsigned)((x >> 41) & 7u) + 1u; si->bits_per_sample = (unsigned)((x >> 36) & 31u) + 1u;
		si->total_samples = x & 0xFFFFFFFFFull;
		memcpy(si->md5sum, p + 18, 16);
		d->in_pos += 34;
		if(!skip_bytes(d, length - 34)) return 0;
		d->has_streaminfo = 1;
		{ static const uint8_t zero[16] = {0}; if(0 == memcmp(si->md5sum, zero, 16)) d->do_md5 = 0; } /* :1620-1621 */
		d->last.channels = si->channels; d->last.bits_per_sample = si->bits_per_sample; d->last.sample_rate = si->sample_rate;
		if(d->respond[FLAC__METADATA_TYPE_STREAMINFO] && d->metadata_cb && !d->is_seeking) d->metadata_cb(d, &d->streaminfo, d->client_data);
	}
	else {
		/* :1450-1592: deliver the block if its type passes the filter; for APPLICATION blocks the listed ids invert it */
		FLAC__bool deliver = type < 127 && d->respond[type] && d->metadata_cb && !d->is_seeking;
		if(type == FLAC__METADATA_TYPE_APPLICATION && d->app_ids_count && d->metadata_cb && !d->is_seeking && length >= 4) {
			size_t i;
			if(!fill(d, 4)) return 0;
			if(avail(d) >= 4) for(i = 0; i < d->app_ids_count; i++) if(0 == memcmp(d->in + d->in_pos, d->app_ids + 4 * i, 4)) { deliver = !d->respond[type]; break; }
		}
		if(type == FLAC__METADATA_TYPE_SEEKTABLE && !d->seek_points && length >= FLAC__STREAM_METADATA_SEEKPOINT_LENGTH) { /* :1623-1650: kept for seek_absolute */
			if(!fill(d, (size_t)length)) return 0;
			if(avail(d) >= length) {
				const unsigned np = (unsigned)(length / FLAC__STREAM_METADATA_SEEKPOINT_LENGTH); unsigned i, k;
				d->seek_points = (FLAC__StreamMetadata_SeekPoint*)malloc(sizeof(FLAC__StreamMetadata_SeekPoint) * np);
				if(d->seek_points) {
					for(i = 0; i < np; i++) {
						const uint8_t *q = d->in + d->in_pos + (size_t)i * FLAC__STREAM_METADATA_SEEKPOINT_LENGTH; FLAC__uint64 a = 0, o = 0;
						for(k = 0; k < 8; k++) { a = (a << 8) | q[k]; o = (o << 8) | q[8 + k]; }
						d->seek_points[i].sample_number = a; d->seek_points[i].stream_offset = o; d->seek_points[i].frame_samples = ((unsigned)q[16] << 8) | q[17];
					}
					d->num_seek_points = np;
				}
			}
		}
		if(deliver) {
			FLAC__StreamMetadata block;
			if(!fill(d, (size_t)length)) return 0;
			if(avail(d) < length) { d->state = FLAC__STREAM_DECODER_END_OF_STREAM; return 1; }
			if(flacamd_metadata_block_parse(type, (int)is_last, d->in + d->in_pos, (size_t)length, &block) != 0) {
				/* contents that contradict the block's length (or no memory for them): the block is passed over with a
				 * report and the audio still decodes */
				d->error_cb(d, FLAC__STREAM_DECODER_ERROR_STATUS_BAD_HEADER, d->client_data);
				d->in_pos += (size_t)length;
			}
			else {
				d->in_pos += (size_t)length;
				d->metadata_cb(d, &block, d->client_data);
				flacamd_metadata_block_free(&block);
			}
		}
		else if(!skip_bytes(d, length)) return 0;
	}
	if(is_last) return start_frames(d);
	return 1;
}

/* ---- damaged input: the reference's serial walk, replayed over what the device found ------------------
 * The device decodes every position with a plausible frame header at once and says for each how far the
 * frame parsed and whether its CRC-16 holds.  When those frames do not simply follow one another, the walk
 * below goes through the bytes the way frame_sync_ / read_frame_ do (stream_decoder.c:1947-2145) and queues
 * what the client has to see, in order: error reports, frames, and zeroed frames for CRC failures. */
enum { EV_FRAME = 0, EV_ZERO_FRAME = 1, EV_ERROR = 2, EV_END = 3, EV_END_FAIL = 4, EV_ABORT = 5 };

static void push_event_c(FLAC__StreamDecoder *d, unsigned kind, unsigned status_or_asg, unsigned bs, uint64_t off, uint64_t sn, uint64_t end, uint32_t cand);
static void push_event(FLAC__StreamDecoder *d, unsigned kind, unsigned status_or_asg, unsigned bs, uint64_t off, uint64_t sn, uint64_t end) { push_event_c(d, kind, status_or_asg, bs, off, sn, end, 0); }
static void push_event_c(FLAC__StreamDecoder *d, unsigned kind, unsigned status_or_asg, unsigned bs, uint64_t off, uint64_t sn, uint64_t end, uint32_t cand)
{
	const unsigned r = d->q_n++;
	d->f_cand[r] = cand; d->f_num[r] = 0;
	d->f_kind[r] = (uint8_t)kind; d->f_asg[r] = status_or_asg; d->f_bs[r] = bs; d->f_off[r] = off; d->f_sn[r] = sn; d->f_end[r] = end; d->f_num[r] = 0;
}

/* read_frame_header_ (:2147-2454) on the bytes after a sync code at b[s]: 0 = a header the reference accepts,
 * 1 = BAD_HEADER, 3 = UNPARSEABLE_STREAM (the error status values), -1 = the buffer ends inside the header.
 * *resume = where the search for the next sync code starts after a rejection. */
struct host_header { unsigned blocksize, channels, assignment, bps, hdr_len; uint64_t number; };
static int probe_header(const FLAC__StreamDecoder *d, const uint8_t *b, size_t n, size_t s, size_t *resume, struct host_header *hh)
{
	const FLAC__StreamMetadata_StreamInfo *si = &d->streaminfo.data.stream_info;
	uint8_t raw[16]; unsigned len = 2, i, extra, unparseable = 0, bs_hint = 0, sr_hint = 0, x; size_t p = s + 2; uint8_t crc = 0;
	const int variable = (b[s + 1] & 1) || si->min_blocksize != si->max_blocksize;
	raw[0] = b[s]; raw[1] = b[s + 1];
	for(i = 0; i < 2; i++) { /* :2189-2201: a sync byte inside the header means the sync was false */
		if(p >= n) return -1;
		if(b[p] == 0xff) { *resume = p; return 1; }
		raw[len++] = b[p++];
	}
	x = raw[2] >> 4; if(x == 0) unparseable = 1; else if(x == 6 || x == 7) bs_hint = x;
	x = raw[2] & 15u; if(x == 15) { *resume = p; return 1; } else if(x >= 12) sr_hint = x; /* :2280-2283 */
	x = raw[3] >> 4; if((x & 8) && (x & 7) > 2) unparseable = 1;
	x = (raw[3] >> 1) & 7u; if(x == 3 || x == 7) unparseable = 1;
	if(raw[3] & 1) unparseable = 1;
	/* the coded number, with the lead-byte tests exactly as bitreader.c:1272-1383 makes them */
	if(p >= n) return -1;
	x = b[p]; raw[len++] = b[p++];
	if(!(x & 0x80)) extra = 0; else if(!(x & 0x20)) extra = 1; else if(!(x & 0x10)) extra = 2; else if(!(x & 0x08)) extra = 3;
	else if(!(x & 0x04)) extra = 4; else if(!(x & 0x02)) extra = 5; else if(variable && !(x & 0x01)) extra = 6;
	else { *resume = p - 1; return 1; } /* :2354-2372: the offending byte is looked at again by the sync search */
	for(i = 0; i < extra; i++) {
		if(p >= n) return -1;
		x = b[p]; raw[len++] = b[p++];
		if(!(x & 0x80) || (x & 0x40)) { *resume = p - 1; return 1; }
	}
	if(bs_hint) { if(p >= n) return -1; raw[len++] = b[p++]; if(bs_hint == 7) { if(p >= n) return -1; raw[len++] = b[p++]; } }
	if(sr_hint) { if(p >= n) return -1; raw[len++] = b[p++]; if(sr_hint != 12) { if(p >= n) return -1; raw[len++] = b[p++]; } }
	if(p >= n) return -1;
	for(i = 0; i < len; i++) { unsigned k; crc ^= raw[i]; for(k = 0; k < 8; k++) crc = (uint8_t)((crc & 0x80) ? (crc << 1) ^ 0x07 : crc << 1); }
	*resume = p + 1;
	if(crc != b[p]) return 1;      /* :2416-2420 */
	if(unparseable) return 3;      /* :2447-2451 */
	if(hh) { /* the fields as read_frame_header_ resolves them (:2203-2360, :2378-2414) */
		const unsigned bs_code = raw[2] >> 4, ch_code = raw[3] >> 4, bps_code = (raw[3] >> 1) & 7u; unsigned at = 5 + extra; uint64_t v;
		const uint8_t lead = raw[4];
		v = extra == 0 ? lead : extra == 6 ? 0 : (uint64_t)(lead & (0x7fu >> (extra + 1)));
		for(i = 0; i < extra; i++) v = (v << 6) | (raw[5 + i] & 0x3fu);
		hh->number = v;
		if(bs_code == 1) hh->blocksize = 192; else if(bs_code <= 5) hh->blocksize = 576u << (bs_code - 2); else if(bs_code == 6) hh->blocksize = raw[at] + 1u;
		else if(bs_code == 7) hh->blocksize = (((unsigned)raw[at] << 8) | raw[at + 1]) + 1u; else hh->blocksize = 256u << (bs_code - 8);
		if(ch_code < 8) { hh->channels = ch_code + 1; hh->assignment = 0; } else { hh->channels = 2; hh->assignment = ch_code - 7; }
		switch(bps_code) { case 1: hh->bps = 8; break; case 2: hh->bps = 12; break; case 4: hh->bps = 16; break; case 5: hh->bps = 20; break; case 6: hh->bps = 24; break; default: hh->bps = si->bits_per_sample; break; }
		hh->hdr_len = len + 1;
	}
	return 0;
}

/* One frame whose parameters differ from the stream's, through a device decoder made for them.  Returns 1 when an event
 * (or several) was queued and *end is where the walk goes on, 0 when the frame runs past the bytes at hand, -1 on failure. */
static int decode_alt_frame(FLAC__StreamDecoder *d, const uint8_t *b, size_t n, size_t s, const struct host_header *hh, int variable, size_t *end)
{
	const FLAC__StreamMetadata_StreamInfo *si = &d->streaminfo.data.stream_info;
	const size_t bound = 16 + (size_t)hh->channels * (((size_t)hh->blocksize * (hh->bps + 2) + 7) / 8 + 64) + 2;
	size_t window = n - s < bound ? n - s : bound;
	uint64_t info[4] = { 0, 0, 0, 0 }, c_pos[8], c_end[8], c_num[8], c_off[8]; uint32_t c_info[8], c_bs[8], nc = 0;
	struct alt_frame *af; int rc; uint64_t sn;
	if(hh->bps < 4 || hh->bps > 24) return -1;
	if(!d->alt_eng || d->alt_ch != hh->channels || d->alt_bps != hh->bps || d->alt_bs < hh->blocksize) {
		if(d->alt_eng) { flacamd_decoder_destroy(d->alt_eng); d->alt_eng = 0; }
		d->alt_bytes = 16 + (size_t)hh->channels * (((size_t)hh->blocksize * (hh->bps + 2) + 7) / 8 + 64) + 2 + 64;
		if(flacamd_decoder_create(&d->alt_eng, d->device, hh->channels, hh->bps, hh->blocksize, 8, d->alt_bytes) != FLACAMD_OK) return -1;
		(void)flacamd_decoder_set_tolerant(d->alt_eng, 1);
		if(flacamd_decoder_set_subframe_info(d->alt_eng, 1) != FLACAMD_OK) return -1;
		d->alt_ch = hh->channels; d->alt_bps = hh->bps; d->alt_bs = hh->blocksize;
	}
	af = (struct alt_frame*)calloc(1, sizeof *af);
	if(!af || !(af->pcm = (int32_t*)calloc((size_t)hh->blocksize * hh->channels, sizeof(int32_t)))) { free(af); return -1; }
	af->ch = hh->channels; af->bps = hh->bps;
	/* the window may show more sync codes than the few candidates this decoder holds: shrink it to the frame's bound first */
	for(;;) {
		rc = flacamd_decode_host(d->alt_eng, b + s, window, af->pcm, FLACAMD_PCM_S32, hh->blocksize, info);
		if(rc == FLACAMD_E_CAPACITY && window > 64) { window = window * 3 / 4; continue; }
		break;
	}
	if(rc == FLACAMD_OK) rc = flacamd_decoder_candidate_table(d->alt_eng, 8, &nc, c_pos, c_end, c_info, c_bs, c_num, c_off);
	if(rc != FLACAMD_OK || nc == 0 || c_pos[0] != 0) { alt_free(af); return -1; }
	{
		flacamd_subframe_info recs[8 * 8];
		if(flacamd_decoder_subframe_info(d->alt_eng, 8, recs, 0, 0) == FLACAMD_OK) memcpy(af->sub, recs, sizeof(flacamd_subframe_info) * hh->channels);
		if(d->want_res && (c_info[0] & 1u) && c_off[0] == 0 && (af->res = (int32_t*)calloc((size_t)hh->blocksize * hh->channels, sizeof(int32_t)))
			&& flacamd_decoder_residuals_host(d->alt_eng, af->res, hh->blocksize) != FLACAMD_OK) { free(af->res); af->res = 0; }
	}
	sn = variable ? hh->number : hh->number * (uint64_t)si->min_blocksize;
	*end = s + (size_t)c_end[0];
	if(!(c_info[0] & 1u)) {
		const uint32_t ex = (c_info[0] >> 8) & 15u;
		alt_free(af);
		if(ex == 6u) return 0;
		push_event(d, EV_ERROR, ex == 4u ? FLAC__STREAM_DECODER_ERROR_STATUS_UNPARSEABLE_STREAM : FLAC__STREAM_DECODER_ERROR_STATUS_LOST_SYNC, 0, 0, 0, *end);
		return 1;
	}
	if(!((c_info[0] & 2u) && c_off[0] == 0)) { /* CRC mismatch: an error report and a frame of zeros (:2111-2119) */
		memset(af->pcm, 0, sizeof(int32_t) * (size_t)hh->blocksize * hh->channels);
		push_event(d, EV_ERROR, FLAC__STREAM_DECODER_ERROR_STATUS_FRAME_CRC_MISMATCH, 0, 0, 0, *end);
	}
	push_event(d, EV_FRAME, (c_info[0] >> 4) & 15u, hh->blocksize, 0, sn, *end);
	d->f_num[d->q_n - 1] = s;
	d->f_alt[d->q_n - 1] = af;
	return 1;
}

/* returns bytes consumed; sets d->state only through queued events */
static size_t walk_damaged(FLAC__StreamDecoder *d, const uint8_t *b, size_t n, int at_eof, unsigned ncand)
{
	const FLAC__StreamMetadata_StreamInfo *si = &d->streaminfo.data.stream_info;
	const FLAC__uint64 total = si->total_samples;
	FLAC__uint64 sd = d->samples_decoded;
	size_t pos = 0; unsigned ci = 0;
	for(;;) {
		size_t s, resume = 0; int hr; struct host_header hh;
		if(d->q_n + 4u > d->ev_cap) return pos;
		if(total > 0 && sd >= total) { push_event(d, EV_END, 0, 0, 0, 0, pos); return pos; } /* :1952-1959 */
		/* frame_sync_ :1967-1997 */
		for(s = pos; s + 1 < n; s++) if(b[s] == 0xff && (b[s + 1] >> 1) == 0x7c) break;
		if(s + 1 >= n) {
			if(at_eof) { /* the read that fails ends the stream; a lone last 0xff is not reported (:1976-1979) */
				if(!(n == pos || (n - pos == 1 && b[pos] == 0xff)) && !d->search_warned) push_event(d, EV_ERROR, FLAC__STREAM_DECODER_ERROR_STATUS_LOST_SYNC, 0, 0, 0, n);
				d->search_warned = 0;
				push_event(d, EV_END, 0, 0, 0, 0, n);
				return n;
			}
			if(n - 1 > pos) {
				if(!d->search_warned) { push_event(d, EV_ERROR, FLAC__STREAM_DECODER_ERROR_STATUS_LOST_SYNC, 0, 0, 0, n - 1); d->search_warned = 1; }
				pos = n - 1;
			}
			return pos; /* more input needed */
		}
		if(s > pos && !d->search_warned) push_event(d, EV_ERROR, FLAC__STREAM_DECODER_ERROR_STATUS_LOST_SYNC, 0, 0, 0, s);
		d->search_warned = 0;
		/* read_frame_header_ */
		hr = probe_header(d, b, n, s, &resume, &hh);
		if(hr < 0) {
			if(at_eof) { push_event(d, EV_END_FAIL, 0, 0, 0, 0, n); return n; }
			if(s == 0 && n >= d->batch_bytes) { d->want_grow = 1; return 0; }
			return s;
		}
		if(hr > 0) { push_event(d, EV_ERROR, (unsigned)hr, 0, 0, 0, resume); pos = resume; continue; }
		while(ci < ncand && d->c_pos[ci] < s) ci++;
		if(ci >= ncand || d->c_pos[ci] != s) {
			/* a header the reference would go on to decode with parameters that differ from STREAMINFO's
			 * (or a blocksize above its maximum): the device decoder is built for the stream's parameters */
			size_t aend = s; int ar;
			const int variable = (b[s + 1] & 1) || si->min_blocksize != si->max_blocksize;
			ar = decode_alt_frame(d, b, n, s, &hh, variable, &aend);
			if(ar == 0) { /* the frame runs past the bytes at hand */
				if(at_eof) { push_event(d, EV_END_FAIL, 0, 0, 0, 0, n); return n; }
				if(s == 0 && n >= d->batch_bytes) { d->want_grow = 1; return 0; }
				return s;
			}
			if(ar < 0) { push_event(d, EV_ERROR, FLAC__STREAM_DECODER_ERROR_STATUS_UNPARSEABLE_STREAM, 0, 0, 0, s); push_event(d, EV_ABORT, 0, 0, 0, 0, s); return s; }
			if(d->f_kind[d->q_n - 1] == EV_FRAME) sd = d->f_sn[d->q_n - 1] + hh.blocksize;
			pos = aend; continue;
		}
		{
			const uint32_t info = d->c_info[ci], ex = (info >> 8) & 15u; const uint64_t end = d->c_end[ci];
			if(!(info & 1u)) {
				if(ex == 6u) { /* the frame runs past the buffer */
					if(at_eof) { push_event(d, EV_END_FAIL, 0, 0, 0, 0, n); return n; }
					if(s == 0 && n >= d->batch_bytes) { d->want_grow = 1; return 0; }
					return s;
				}
				/* the subframe-level rejections (:2481, :2494-2508, :2580, :2628, :2732-2745, :2785); fields the
				 * reference does not define end the frame as a lost sync here */
				push_event(d, EV_ERROR, ex == 4u ? FLAC__STREAM_DECODER_ERROR_STATUS_UNPARSEABLE_STREAM : FLAC__STREAM_DECODER_ERROR_STATUS_LOST_SYNC, 0, 0, 0, end);
				pos = (size_t)end; continue;
			}
			{
				const uint32_t bs = d->c_bs[ci];
				const int variable = (b[s + 1] & 1) || si->min_blocksize != si->max_blocksize;
				const uint64_t sn = variable ? d->c_num[ci] : d->c_num[ci] * (uint64_t)si->min_blocksize; /* :2422-2445 */
				if((info & 2u) && d->c_off[ci] != ~(uint64_t)0) { push_event_c(d, EV_FRAME, (info >> 4) & 15u, bs, d->c_off[ci], sn, end, ci); d->f_num[d->q_n - 1] = s; }
				else { /* :2111-2119 */
					push_event(d, EV_ERROR, FLAC__STREAM_DECODER_ERROR_STATUS_FRAME_CRC_MISMATCH, 0, 0, 0, end);
					push_event_c(d, EV_ZERO_FRAME, (info >> 4) & 15u, bs, 0, sn, end, ci); d->f_num[d->q_n - 1] = s;
				}
				sd = sn + bs; pos = (size_t)end;
			}
		}
	}
}

/* forget the queued events (their frames with parameters of their own included) */
static void drop_events(FLAC__StreamDecoder *d)
{
	unsigned k;
	for(k = 0; d->f_alt && k < d->q_n; k++) if(d->f_alt[k]) { alt_free(d->f_alt[k]); d->f_alt[k] = 0; }
	d->q_n = d->q_i = 0;
}

/* a device decoder for the stream's parameters taking up to d->batch_bytes per call */
static FLAC__bool make_engine(FLAC__StreamDecoder *d)
{
	const FLAC__StreamMetadata_StreamInfo *si = &d->streaminfo.data.stream_info;
	if(d->eng) { flacamd_decoder_destroy(d->eng); d->eng = 0; }
	if(flacamd_decoder_create(&d->eng, d->device, si->channels, si->bits_per_sample, si->max_blocksize, d->eng_frames, d->batch_bytes) != FLACAMD_OK
		|| flacamd_decoder_set_tolerant(d->eng, 1) != FLACAMD_OK || flacamd_decoder_set_subframe_info(d->eng, 1) != FLACAMD_OK) {
		d->state = FLAC__STREAM_DECODER_MEMORY_ALLOCATION_ERROR; return 0;
	}
	return 1;
}

/* decode the next batch of frames on the device; leaves q_n > q_i, or a terminal state */
static FLAC__bool decode_batch(FLAC__StreamDecoder *d)
{
	const FLAC__StreamMetadata_StreamInfo *si = &d->streaminfo.data.stream_info;
	size_t window, least; uint64_t info[4] = {0, 0, 0, 0}; int rc; unsigned r;
	drop_events(d);
	if(si->total_samples > 0 && d->samples_decoded >= si->total_samples) { d->state = FLAC__STREAM_DECODER_END_OF_STREAM; return 1; } /* :1966-1970 */
	/* enough input for one frame of the usual size at least; whatever one read call delivers beyond that is taken along */
	least = d->frame_bound < d->batch_bytes ? d->frame_bound : d->batch_bytes;
	for(;;) {
		if(!fill_range(d, least, d->batch_bytes)) return 0;
		if(avail(d) == 0) { d->state = FLAC__STREAM_DECODER_END_OF_STREAM; return 1; }
		window = avail(d) < d->batch_bytes ? avail(d) : d->batch_bytes;
		for(;;) {
			rc = flacamd_decode_host(d->eng, d->in + d->in_pos, window, d->pcm, FLACAMD_PCM_S32_FRAMES, d->pcm_cap, info);   /* every frame as the write callback wants it */
			if(rc == FLACAMD_E_CAPACITY && window > 1024u) { window /= 2; continue; } /* more frames in the window than a batch holds */
			break;
		}
		if(rc != FLACAMD_OK) { d->state = (rc == FLACAMD_E_OUT_OF_MEMORY) ? FLAC__STREAM_DECODER_MEMORY_ALLOCATION_ERROR : FLAC__STREAM_DECODER_ABORTED; return 0; }
		d->batch_base = d->in_base + d->in_pos; d->batch_in_off = d->in_pos;
		if(flacamd_decoder_subframe_info(d->eng, d->eng_frames, d->sub, d->frame_cand, d->eng_frames) != FLACAMD_OK) { d->state = FLAC__STREAM_DECODER_ABORTED; return 0; }
		if(d->res && flacamd_decoder_residuals_host(d->eng, d->res, d->pcm_cap) != FLACAMD_OK) { d->state = FLAC__STREAM_DECODER_ABORTED; return 0; }
		if(flacamd_decoder_damage(d->eng) == 0) {
			const uint8_t *b = d->in + d->in_pos;
			if(flacamd_decoder_frame_table(d->eng, (uint32_t)info[0], d->f_off, d->f_bs, d->f_num, d->f_asg, d->c_pos, d->f_end) != FLACAMD_OK) { d->state = FLAC__STREAM_DECODER_ABORTED; return 0; }
			d->q_n = (unsigned)info[0];
			for(r = 0; r < d->q_n; r++) { /* the sample number as read_frame_header_ derives it (:2346-2376, :2422-2445) */
				const int variable = (b[d->c_pos[r] + 1] & 1) || si->min_blocksize != si->max_blocksize;
				d->f_kind[r] = EV_FRAME; d->f_sn[r] = variable ? d->f_num[r] : d->f_num[r] * (uint64_t)si->min_blocksize;
				d->f_cand[r] = d->frame_cand[r]; d->f_num[r] = d->c_pos[r];   /* from here on: where the frame starts in the batch */
			}
			d->search_warned = 0;
			d->in_pos += (size_t)info[3];
			return 1;
		}
		{
			uint32_t ncand = 0; size_t used;
			if(flacamd_decoder_candidate_table(d->eng, d->eng_frames, &ncand, d->c_pos, d->c_end, d->c_info, d->c_bs, d->c_num, d->c_off) != FLACAMD_OK) { d->state = FLAC__STREAM_DECODER_ABORTED; return 0; }
			d->want_grow = 0;
			used = walk_damaged(d, d->in + d->in_pos, window, d->eof && window == avail(d), ncand);
			d->in_pos += used;
			if(d->q_n > 0 || used > 0) return 1;
			/* nothing could be decided: the first frame of the window is not all there yet */
			if(d->want_grow) { /* it is longer than the window: a larger window (frames of up to 16 MiB exist, format.c:72) */
				if(d->batch_bytes >= ((size_t)1 << 30)) { d->state = FLAC__STREAM_DECODER_ABORTED; return 0; }
				d->batch_bytes *= 2;
				if(!make_engine(d)) return 0;
				least = d->batch_bytes;
				continue;
			}
			if(!d->eof && avail(d) < d->batch_bytes) { least = avail(d) + 1; continue; }   /* more input */
			d->state = FLAC__STREAM_DECODER_ABORTED; return 0;
		}
	}
}

/* the sample rate out of the frame's own header bytes, and the subframe records of its candidate */
static void fill_frame_details(FLAC__StreamDecoder *d, FLAC__Frame *frame, const uint8_t *fb, size_t flen, const flacamd_subframe_info *recs, const int32_t *res)
{
	static const unsigned rates[12] = { 0, 88200, 176400, 192000, 8000, 16000, 22050, 24000, 32000, 44100, 48000, 96000 };
	const unsigned ch = frame->header.channels; unsigned c;
	if(flen >= 8) {
		const unsigned bs_code = fb[2] >> 4, sr_code = fb[2] & 15u; const uint8_t x = fb[4];
		size_t p = 5 + (!(x & 0x80) ? 0u : !(x & 0x20) ? 1u : !(x & 0x10) ? 2u : !(x & 0x08) ? 3u : !(x & 0x04) ? 4u : !(x & 0x02) ? 5u : 6u);
		if(bs_code == 6) p += 1; else if(bs_code == 7) p += 2;
		if(sr_code && sr_code < 12) frame->header.sample_rate = rates[sr_code];
		else if(sr_code == 12 && p < flen) { frame->header.sample_rate = fb[p] * 1000u; p += 1; }
		else if((sr_code == 13 || sr_code == 14) && p + 1 < flen) { frame->header.sample_rate = (((unsigned)fb[p] << 8) | fb[p + 1]) * (sr_code == 14 ? 10u : 1u); p += 2; }
		/* header.crc and footer.crc stay 0: the reference's decoder checks both CRCs but never stores them in the frame it hands out */
	}
	for(c = 0; c < ch && c < 8; c++) {
		const flacamd_subframe_info *rec = &recs[c];
		FLAC__Subframe *sf = &frame->subframes[c];
		const unsigned type = rec->head & 3u, order = (rec->head >> 2) & 63u, po = (rec->head >> 19) & 15u; unsigned i;
		FLAC__EntropyCodingMethod *ecm = 0;
		sf->type = (FLAC__SubframeType)type; sf->wasted_bits = (rec->head >> 8) & 31u;
		if(type == FLAC__SUBFRAME_TYPE_CONSTANT) sf->data.constant.value = rec->value;
		else if(type == FLAC__SUBFRAME_TYPE_VERBATIM) sf->data.verbatim.data = res ? res + (size_t)c * frame->header.blocksize : 0;
		else if(type == FLAC__SUBFRAME_TYPE_FIXED) {
			sf->data.fixed.order = order; sf->data.fixed.residual = res ? res + (size_t)c * frame->header.blocksize : 0; ecm = &sf->data.fixed.entropy_coding_method;
			for(i = 0; i < order && i < FLAC__MAX_FIXED_ORDER; i++) sf->data.fixed.warmup[i] = rec->warmup[i];
		}
		else if(type == FLAC__SUBFRAME_TYPE_LPC) {
			const unsigned lv = (rec->head >> 23) & 31u;
			sf->data.lpc.order = order; sf->data.lpc.residual = res ? res + (size_t)c * frame->header.blocksize : 0; ecm = &sf->data.lpc.entropy_coding_method;
			sf->data.lpc.qlp_coeff_precision = (rec->head >> 13) & 31u;
			sf->data.lpc.quantization_level = (lv & 16u) ? (int)lv - 32 : (int)lv;
			for(i = 0; i < order && i < FLAC__MAX_LPC_ORDER; i++) { sf->data.lpc.qlp_coeff[i] = rec->q[i]; sf->data.lpc.warmup[i] = rec->warmup[i]; }
		}
		if(ecm) {
			const unsigned rice2 = (rec->head >> 18) & 1u, escape = rice2 ? 31u : 15u, cap = po > 8 ? po : 8u;
			FLAC__EntropyCodingMethod_PartitionedRiceContents *pc = &d->prc[c];
			if(d->prc_alloc[c] < cap || !pc->parameters) {
				unsigned *np = (unsigned*)realloc(pc->parameters, sizeof(unsigned) << cap), *nr = (unsigned*)realloc(pc->raw_bits, sizeof(unsigned) << cap);
				if(np) pc->parameters = np;
				if(nr) pc->raw_bits = nr;
				if(!np || !nr) { ecm->data.partitioned_rice.contents = 0; ecm->data.partitioned_rice.order = po; continue; }
				d->prc_alloc[c] = cap; pc->capacity_by_order = cap;
			}
			for(i = 0; i < (1u << po); i++) {
				const unsigned v = i < FLACAMD_SUBINFO_PARTS ? rec->rice[i] : 0u;
				if(v & 0x80u) { pc->parameters[i] = escape; pc->raw_bits[i] = v & 31u; } else { pc->parameters[i] = v; pc->raw_bits[i] = 0; }
			}
			ecm->type = rice2 ? FLAC__ENTROPY_CODING_METHOD_PARTITIONED_RICE2 : FLAC__ENTROPY_CODING_METHOD_PARTITIONED_RICE;
			ecm->data.partitioned_rice.order = po; ecm->data.partitioned_rice.contents = pc;
		}
	}
}

/* the next frame of the stream: error reports on the way, then write_audio_frame_to_client_ (:2915-2990) */
static FLAC__bool next_frame(FLAC__StreamDecoder *d, int deliver)
{
	const FLAC__StreamMetadata_StreamInfo *si = &d->streaminfo.data.stream_info;
	FLAC__Frame frame; const FLAC__int32 *bufs[8]; unsigned r, c, n, ch = si->channels, bps = si->bits_per_sample, skip = 0, kind; const int32_t *src; size_t i; struct alt_frame *alt;
	for(;;) {
		if(d->q_i == d->q_n) {
			if(!decode_batch(d)) return 0;
			if(d->q_i == d->q_n) { if(d->state == FLAC__STREAM_DECODER_END_OF_STREAM || d->state == FLAC__STREAM_DECODER_ABORTED) return 1; continue; }
		}
		r = d->q_i++;
		kind = d->f_kind[r];
		d->decode_pos = d->batch_base + d->f_end[r];
		if(kind == EV_ERROR) { d->error_cb(d, (FLAC__StreamDecoderErrorStatus)d->f_asg[r], d->client_data); continue; }
		if(kind == EV_END) { d->state = FLAC__STREAM_DECODER_END_OF_STREAM; return 1; }
		if(kind == EV_END_FAIL) { d->state = FLAC__STREAM_DECODER_END_OF_STREAM; return 0; } /* the input ended inside a frame (:2018, :2054, :2068 return false) */
		if(kind == EV_ABORT) { d->state = FLAC__STREAM_DECODER_ABORTED; return 0; }
		break;
	}
	n = d->f_bs[r];
	alt = d->f_alt[r]; d->f_alt[r] = 0;
	if(alt) { ch = alt->ch; bps = alt->bps; src = alt->pcm; }   /* a frame with parameters of its own (decode_alt_frame) */
	else src = kind == EV_FRAME ? d->pcm + d->f_off[r] * ch : 0;
	if((size_t)n * ch > d->planar_cap) { /* longer than the stream's maximum says: the reference reallocates its output too (:2012) */
		int32_t *np = (int32_t*)realloc(d->planar, sizeof(int32_t) * (size_t)n * ch);
		if(!np) { if(alt) { alt_free(alt); } d->state = FLAC__STREAM_DECODER_MEMORY_ALLOCATION_ERROR; return 0; }
		d->planar = np; d->planar_cap = (size_t)n * ch;
	}
	if(d->f_sn[r] != ~(uint64_t)0) d->samples_decoded = d->f_sn[r]; /* :2135: the header's number, not a running count */
	memset(&frame, 0, sizeof frame);
	frame.header.blocksize = n; frame.header.sample_rate = si->sample_rate; frame.header.channels = ch;
	frame.header.channel_assignment = (FLAC__ChannelAssignment)d->f_asg[r]; frame.header.bits_per_sample = bps;
	frame.header.number_type = FLAC__FRAME_NUMBER_TYPE_SAMPLE_NUMBER;
	frame.header.number.sample_number = d->samples_decoded; /* frame number x fixed blocksize for the streams this decoder takes (:2422-2445) */
	{ /* what read_frame_header_ / read_subframe_* leave in the frame (:2147-2454, :2456-2719) */
		const uint8_t *fb = d->in + d->batch_in_off + d->f_num[r]; const size_t flen = (size_t)(d->f_end[r] - d->f_num[r]);
		/* residuals: the frame's rows of the batch, the frame's own for one decoded by itself; a frame that is handed out
		 * zeroed (CRC mismatch) points at zeros, as any valid memory is as good as what the reference leaves there */
		const int32_t *res = !d->want_res ? 0 : alt ? (alt->res ? alt->res : d->planar) : (src && d->res) ? d->res + d->f_off[r] * ch : d->planar;
		fill_frame_details(d, &frame, fb, flen, alt ? alt->sub : &d->sub[(size_t)d->f_cand[r] * ch], res);
	}
	d->last = frame.header;
	if(d->do_md5 && deliver) {
		if(src && alt) md5_feed_pcm(&d->md5, src, (size_t)n * ch, bps, d->md5_scratch, 65536);
		else if(src) { for(c = 0; c < ch; c++) bufs[c] = src + (size_t)c * n; md5_feed_planar(&d->md5, bufs, ch, n, bps, d->md5_scratch, 65536); }
		else { /* a zeroed frame goes through the checksum like any other (:2964) */
			memset(d->planar, 0, sizeof(int32_t) * (size_t)n * ch);
			md5_feed_pcm(&d->md5, d->planar, (size_t)n * ch, bps, d->md5_scratch, 65536);
		}
	}
	if(d->is_seeking) { /* :2920-2956: drop frames before the target, trim the one that holds it */
		const FLAC__uint64 first = d->samples_decoded;
		if(d->target_sample >= first + n) { d->samples_decoded += n; if(alt) { alt_free(alt); } return 1; }
		skip = (unsigned)(d->target_sample - first);
		d->is_seeking = 0;
		frame.header.blocksize = n - skip; frame.header.number.sample_number = first + skip;
	}
	d->samples_decoded += n;
	if(!deliver) { if(alt) { alt_free(alt); } return 1; }
	for(c = 0; c < ch; c++) {
		int32_t *dst = d->planar + (size_t)c * n;
		if(src && !alt) { bufs[c] = src + (size_t)c * n + skip; continue; }   /* the batch came back frame by frame, channel by channel: handed on as it lies */
		if(src) for(i = skip; i < n; i++) dst[i - skip] = src[i * ch + c];
		else memset(dst, 0, sizeof(int32_t) * n);
		bufs[c] = dst;
	}
	/* without the residual pass a verbatim subframe points at the channel's output samples (the coded ones, unless the
	 * subframe has wasted bits or the channel was decorrelated) */
	for(c = 0; c < ch; c++) if(frame.subframes[c].type == FLAC__SUBFRAME_TYPE_VERBATIM && !frame.subframes[c].data.verbatim.data) frame.subframes[c].data.verbatim.data = bufs[c];
	{
		const FLAC__StreamDecoderWriteStatus ws = d->write_cb(d, &frame, bufs, d->client_data);
		alt_free(alt);
		/* :2138-2141: read_frame_ returns false and leaves the state it had while reading the frame */
		if(ws != FLAC__STREAM_DECODER_WRITE_STATUS_CONTINUE) { d->state = FLAC__STREAM_DECODER_READ_FRAME; return 0; }
	}
	return 1;
}

FLAC__bool FLAC__stream_decoder_process_single(FLAC__StreamDecoder *d) /* :995-1026 */
{
	if(!d) return 0;
	switch(d->state) {
	case FLAC__STREAM_DECODER_SEARCH_FOR_METADATA: return find_metadata(d);
	case FLAC__STREAM_DECODER_READ_METADATA: return read_metadata(d);
	case FLAC__STREAM_DECODER_SEARCH_FOR_FRAME_SYNC: case FLAC__STREAM_DECODER_READ_FRAME: return next_frame(d, 1);
	case FLAC__STREAM_DECODER_END_OF_STREAM: case FLAC__STREAM_DECODER_ABORTED: return 1;
	default: return 0;
	}
}

FLAC__bool FLAC__stream_decoder_process_until_end_of_metadata(FLAC__StreamDecoder *d) /* :1028-1055 */
{
	if(!d) return 0;
	for(;;) {
		switch(d->state) {
		case FLAC__STREAM_DECODER_SEARCH_FOR_METADATA: if(!find_metadata(d)) return 0; break;
		case FLAC__STREAM_DECODER_READ_METADATA: if(!read_metadata(d)) return 0; break;
		case FLAC__STREAM_DECODER_SEARCH_FOR_FRAME_SYNC: case FLAC__STREAM_DECODER_READ_FRAME: case FLAC__STREAM_DECODER_END_OF_STREAM: case FLAC__STREAM_DECODER_ABORTED: return 1;
		default: return 0;
		}
	}
}

FLAC__bool FLAC__stream_decoder_process_until_end_of_stream(FLAC__StreamDecoder *d) /* :1057-1090 */
{
	if(!d) return 0;
	for(;;) {
		switch(d->state) {
		case FLAC__STREAM_DECODER_SEARCH_FOR_METADATA: if(!find_metadata(d)) return 0; break;
		case FLAC__STREAM_DECODER_READ_METADATA: if(!read_metadata(d)) return 0; break;
		case FLAC__STREAM_DECODER_SEARCH_FOR_FRAME_SYNC: case FLAC__STREAM_DECODER_READ_FRAME: if(!next_frame(d, 1)) return 0; break;
		case FLAC__STREAM_DECODER_END_OF_STREAM: case FLAC__STREAM_DECODER_ABORTED: return 1;
		default: return 0;
		}
	}
}

FLAC__bool FLAC__stream_decoder_skip_single_frame(FLAC__StreamDecoder *d) /* :1092-1122 */
{
	if(!d) return 0;
	for(;;) {
		switch(d->state) {
		case FLAC__STREAM_DECODER_SEARCH_FOR_METADATA: case FLAC__STREAM_DECODER_READ_METADATA: return 0; /* the caller must be past the metadata */
		case FLAC__STREAM_DECODER_SEARCH_FOR_FRAME_SYNC: case FLAC__STREAM_DECODER_READ_FRAME: d->do_md5 = 0; return next_frame(d, 0);
		case FLAC__STREAM_DECODER_END_OF_STREAM: case FLAC__STREAM_DECODER_ABORTED: return 1;
		default: return 0;
		}
	}
}

FLAC__bool FLAC__stream_decoder_flush(FLAC__StreamDecoder *d) /* :952-973 */
{
	if(!d || d->state == FLAC__STREAM_DECODER_UNINITIALIZED) return 0;
	d->samples_decoded = 0; d->do_md5 = 0;
	drop_events(d);
	d->in_base += d->in_len; d->in_len = d->in_pos = 0; d->eof = 0; d->decode_pos = 0;
	d->resync_base = 1;   /* the client may reposition its input now: the next read asks tell_cb where it starts */
	d->state = FLAC__STREAM_DECODER_SEARCH_FOR_FRAME_SYNC;
	return 1;
}

FLAC__bool FLAC__stream_decoder_reset(FLAC__StreamDecoder *d) /* :975-1040 */
{
	if(!FLAC__stream_decoder_flush(d)) return 0;
	if(d->file == stdin) return 0;
	if(d->seek_cb && d->seek_cb(d, 0, d->client_data) == FLAC__STREAM_DECODER_SEEK_STATUS_ERROR) return 0;
	d->in_base = 0; d->warned_sync = 0; d->decode_pos = 0;
	d->state = FLAC__STREAM_DECODER_SEARCH_FOR_METADATA;
	d->has_streaminfo = 0; d->first_frame_offset = 0;
	d->do_md5 = d->md5_checking; md5_init(&d->md5);
	return 1;
}

/* The first CRC-valid frame at or after absolute byte offset `at`: its offset and the number of its first sample.
 * Reads a window of a few frames there and lets the device decode every plausible position in it (sync code +
 * CRC-8 + CRC-16 are what make a frame start certain, as in frame_sync_ / read_frame_, :1947-2145). */
static int probe_at(FLAC__StreamDecoder *d, FLAC__uint64 at, FLAC__uint64 *frame_pos, FLAC__uint64 *frame_sample)
{
	const FLAC__StreamMetadata_StreamInfo *si = &d->streaminfo.data.stream_info;
	size_t want = 4 * d->frame_bound + 65536, got = 0; uint64_t info[4]; uint32_t nc = 0, i; uint8_t *buf; int found = 0;
	if(want > d->batch_bytes) want = d->batch_bytes;
	if(d->seek_cb(d, at, d->client_data) != FLAC__STREAM_DECODER_SEEK_STATUS_OK) return -1;
	if(!(buf = (uint8_t*)malloc(want))) return -1;
	while(got < want) {
		size_t n = want - got;
		const FLAC__StreamDecoderReadStatus st = d->read_cb(d, buf + got, &n, d->client_data);
		if(st == FLAC__STREAM_DECODER_READ_STATUS_ABORT) { free(buf); return -1; }
		got += n;
		if(st == FLAC__STREAM_DECODER_READ_STATUS_END_OF_STREAM || n == 0) break;
	}
	if(got >= 16 && flacamd_decode_host(d->eng, buf, got, d->pcm, FLACAMD_PCM_S32, d->pcm_cap, info) == FLACAMD_OK
		&& flacamd_decoder_candidate_table(d->eng, d->eng_frames, &nc, d->c_pos, d->c_end, d->c_info, d->c_bs, d->c_num, d->c_off) == FLACAMD_OK) {
		for(i = 0; i < nc && !found; i++) if((d->c_info[i] & 3u) == 3u) {
			const int variable = (buf[d->c_pos[i] + 1] & 1) || si->min_blocksize != si->max_blocksize;
			*frame_pos = at + d->c_pos[i]; *frame_sample = variable ? d->c_num[i] : d->c_num[i] * (FLAC__uint64)si->min_blocksize;
			found = 1;
		}
	}
	free(buf);
	return found;
}

FLAC__bool FLAC__stream_decoder_seek_absolute(FLAC__StreamDecoder *d, FLAC__uint64 sample) /* :1124-1180 */
{
	FLAC__uint64 total, lo_byte, lo_sample = 0, hi_byte = 0, hi_sample, length = 0, near; unsigned i, tries;
	if(!d) return 0;
	if(d->state != FLAC__STREAM_DECODER_SEARCH_FOR_METADATA && d->state != FLAC__STREAM_DECODER_READ_METADATA && d->state != FLAC__STREAM_DECODER_SEARCH_FOR_FRAME_SYNC
		&& d->state != FLAC__STREAM_DECODER_READ_FRAME && d->state != FLAC__STREAM_DECODER_END_OF_STREAM) return 0;
	if(!d->seek_cb) return 0;
	d->do_md5 = 0;
	d->is_seeking = 1; /* keeps the metadata callback quiet while the head is read (:1140-1150) */
	if(d->state == FLAC__STREAM_DECODER_SEARCH_FOR_METADATA || d->state == FLAC__STREAM_DECODER_READ_METADATA) {
		if(!FLAC__stream_decoder_process_until_end_of_metadata(d)) { d->is_seeking = 0; return 0; }
	}
	total = FLAC__stream_decoder_get_total_samples(d);
	if(!d->has_streaminfo || !d->eng || (total > 0 && sample >= total)) { d->is_seeking = 0; return 0; }
	drop_events(d);
	/* Where to start decoding: a frame that begins at or before the target and as close to it as is cheap to find.
	 * seek_to_absolute_sample_ (:2979-3192) narrows a byte range with the seek table and then by interpolation, reading one
	 * frame per step; here a step is one small device decode that says which frame starts where, and the forward decode
	 * that follows drops what lies before the target - so the range only has to shrink to about a batch. */
	lo_byte = d->first_frame_offset; hi_sample = total;
	if(d->length_cb && d->length_cb(d, &length, d->client_data) == FLAC__STREAM_DECODER_LENGTH_STATUS_OK) hi_byte = length;
	for(i = 0; i < d->num_seek_points; i++) { /* :3021-3094: the closest points on either side */
		const FLAC__StreamMetadata_SeekPoint *pt = &d->seek_points[i];
		if(pt->sample_number == FLAC__STREAM_METADATA_SEEKPOINT_PLACEHOLDER_VALUE || pt->frame_samples == 0 || (total > 0 && pt->sample_number >= total)) continue;
		if(hi_byte && d->first_frame_offset + pt->stream_offset >= hi_byte) continue;
		if(pt->sample_number <= sample && pt->sample_number >= lo_sample) { lo_sample = pt->sample_number; lo_byte = d->first_frame_offset + pt->stream_offset; }
		else if(pt->sample_number > sample && pt->sample_number < hi_sample) { hi_sample = pt->sample_number; if(hi_byte) hi_byte = d->first_frame_offset + pt->stream_offset; }
	}
	near = (FLAC__uint64)d->batch_frames * d->streaminfo.data.stream_info.max_blocksize / 4;
	for(tries = 0; tries < 24 && hi_byte > lo_byte && total > 0 && hi_sample > lo_sample && sample - lo_sample > near; tries++) {
		/* interpolate (:3124-3135), aiming a little short of the target so that the frame found starts before it */
		FLAC__uint64 aim = sample - lo_sample > near / 2 ? sample - near / 2 : lo_sample, guess, fpos = 0, fsample = 0; int pr;
		guess = lo_byte + (FLAC__uint64)((double)(aim - lo_sample) / (double)(hi_sample - lo_sample) * (double)(hi_byte - lo_byte));
		if(guess >= hi_byte) guess = hi_byte - 1;
		if(guess <= lo_byte) break;
		pr = probe_at(d, guess, &fpos, &fsample);
		if(pr < 0) { d->is_seeking = 0; d->state = FLAC__STREAM_DECODER_SEEK_ERROR; return 0; }
		if(pr == 0 || fsample > sample || fpos >= hi_byte) { hi_byte = guess; if(pr > 0 && fsample > sample && fsample < hi_sample) hi_sample = fsample; continue; }
		if(fpos <= lo_byte) break;
		lo_byte = fpos; lo_sample = fsample;
	}
	if(d->seek_cb(d, lo_byte, d->client_data) != FLAC__STREAM_DECODER_SEEK_STATUS_OK) { d->is_seeking = 0; d->state = FLAC__STREAM_DECODER_SEEK_ERROR; return 0; }
	d->in_base = lo_byte; d->in_len = d->in_pos = 0; d->eof = 0; d->samples_decoded = lo_sample; d->resync_base = 0; d->search_warned = 1;
	d->state = FLAC__STREAM_DECODER_SEARCH_FOR_FRAME_SYNC;
	d->target_sample = sample;
	while(d->is_seeking) {
		if(!next_frame(d, 1)) { d->is_seeking = 0; return 0; }
		if(d->state == FLAC__STREAM_DECODER_END_OF_STREAM && d->is_seeking) { d->is_seeking = 0; d->state = FLAC__STREAM_DECODER_SEEK_ERROR; return 0; }
	}
	return 1;
}

FLAC__bool FLAC__stream_decoder_finish(FLAC__StreamDecoder *d) /* :640-695 */
{
	FLAC__bool md5_failed = 0; uint8_t digest[16];
	if(!d) return 0;
	if(d->state == FLAC__STREAM_DECODER_UNINITIALIZED) return 1;
	md5_final(&d->md5, digest);
	if(d->do_md5 && d->has_streaminfo && memcmp(d->streaminfo.data.stream_info.md5sum, digest, 16)) md5_failed = 1;
	if(d->file) { if(d->file != stdin) fclose(d->file); d->file = 0; }
	dec_free(d);
	dec_defaults(d);
	return !md5_failed;
}
