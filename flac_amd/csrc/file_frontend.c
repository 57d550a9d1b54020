/* Whole-file front end (include/flac_amd_files.h) in plain C: container parsing, sample
 * formatting, metadata layout and header writing of the reference's command-line tool,
 * driving the libFLAC-style objects of this library.  Citations are to
 * /root/reference/src/flac/encode.c (E:) and /root/reference/src/flac/decode.c (D:).
 */
#define _FILE_OFFSET_BITS 64
#define _DEFAULT_SOURCE
#include "../../include/flac_amd_files.h"
#include "../../include/flac_amd_stream.h"
#include "../../include/flac_amd_engine.h"
#include "metadata_blocks.h"
#include <fcntl.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#define DEFAULT_PADDING 8192   /* FLAC_ENCODE__DEFAULT_PADDING, src/flac/encode.h */
#define CHANNEL_MASK_TAG "WAVEFORMATEXTENSIBLE_CHANNEL_MASK"   /* src/flac/utils.c:32 */

void flacamd_file_options_init(flacamd_file_options *o)
{
	memset(o, 0, sizeof *o);
	o->compression_level = 5; o->padding = -1; o->seekpoints = 0; o->input_format = FLACAMD_FILE_AUTO;
}

static int fail(flacamd_file_result *res, int code, const char *fmt, ...)
{
	if(res && !res->error[0]) { va_list ap; va_start(ap, fmt); vsnprintf(res->error, sizeof res->error, fmt, ap); va_end(ap); }
	return code;
}

/* ---- input description ---- */
typedef struct {
	const uint8_t *data; uint64_t data_bytes;       /* sample bytes inside the mapping, as the header declares them */
	uint64_t available_bytes;                       /* what the file really holds of them */
	unsigned channels, bps /* container bits */, shift, sample_rate;
	int big_endian, is_unsigned;
	uint32_t channel_mask;
} pcm_input;

static uint32_t le32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint32_t be32(const uint8_t *p) { return (uint32_t)p[3] | ((uint32_t)p[2] << 8) | ((uint32_t)p[1] << 16) | ((uint32_t)p[0] << 24); }
static unsigned le16(const uint8_t *p) { return (unsigned)p[0] | ((unsigned)p[1] << 8); }
static unsigned be16(const uint8_t *p) { return (unsigned)p[1] | ((unsigned)p[0] << 8); }
static unsigned popcount32(uint32_t v) { unsigned n = 0; for(; v; v &= v - 1) n++; return n; }

static uint64_t le64(const uint8_t *p) { return (uint64_t)le32(p) | ((uint64_t)le32(p + 4) << 32); }

static const uint8_t W64_RIFF[16] = { 'r','i','f','f', 0x2E,0x91,0xCF,0x11, 0xD6,0xA5,0x28,0xDB, 0x04,0xC1,0x00,0x00 };
static const uint8_t W64_WAVE[16] = { 'w','a','v','e', 0xF3,0xAC,0xD3,0x11, 0xD1,0x8C,0x00,0xC0, 0x4F,0x8E,0xDB,0x8A };
static const uint8_t W64_FMT[16]  = { 'f','m','t',' ', 0xF3,0xAC,0xD3,0x11, 0xD1,0x8C,0x00,0xC0, 0x4F,0x8E,0xDB,0x8A };
static const uint8_t W64_DATA[16] = { 'd','a','t','a', 0xF3,0xAC,0xD3,0x11, 0xD1,0x8C,0x00,0xC0, 0x4F,0x8E,0xDB,0x8A };

/* RIFF WAVE, RF64 (a 'ds64' chunk carries the 64-bit sizes) and Wave64 (16-byte chunk ids, 64-bit sizes that count the
 * 24-byte chunk header, 8-byte alignment): get_sample_info_wave, E:202-679 */
/* Position after a chunk body of `len` bytes and `pad` alignment bytes that starts at pos (pos <= size).  A chunk
 * that claims to run past the end of the file leaves the position at the end: the reference seeks past the end
 * without complaint and then finds no further chunk (E:606-660, fskip_ahead), which is what the callers then report. */
static uint64_t skip_chunk(uint64_t pos, uint64_t size, uint64_t len, uint64_t pad)
{
	const uint64_t room = size - pos;
	if(len > room || pad > room - len) return size;
	return pos + len + pad;
}

static int parse_wave(const uint8_t *f, uint64_t size, int variant, pcm_input *in, flacamd_file_result *res)
{
	const int w64 = variant == FLACAMD_FILE_WAVE64, rf64 = variant == FLACAMD_FILE_RF64;
	const unsigned idlen = w64 ? 16u : 4u, hdr = w64 ? 24u : 8u;
	uint64_t pos = w64 ? 40 : 12, ds64_data = 0; int got_fmt = 0, got_ds64 = 0;
	in->big_endian = 0;
	/* pos never passes size, and every comparison against the end of the file subtracts from size instead of adding
	 * to pos: a declared 64-bit length near 2^64 must not wrap the sum */
	while(pos <= size && size - pos >= idlen) {
		const uint8_t *id = f + pos;
		uint64_t len;   /* bytes after the chunk header, as declared */
		if(size - pos < hdr) return fail(res, FLACAMD_E_INVALID_ARGUMENT, "incomplete chunk header");
		if(w64) {
			len = le64(f + pos + 16);
			pos += hdr;
			if(0 == memcmp(id, W64_FMT, 16)) {
				if(len >> 32) return fail(res, FLACAMD_E_INVALID_ARGUMENT, "freakishly large Wave64 'fmt ' chunk");
				if(len < 24) return fail(res, FLACAMD_E_INVALID_ARGUMENT, "freakishly small Wave64 'fmt ' chunk");
			}
			else if(0 == memcmp(id, W64_DATA, 16)) { if(len < 24) return fail(res, FLACAMD_E_INVALID_ARGUMENT, "freakishly small Wave64 'data' chunk"); }
			else { len = (len + 7) & ~(uint64_t)7; if(len < 24) return fail(res, FLACAMD_E_INVALID_ARGUMENT, "freakishly small Wave64 chunk"); }
			len -= 24;
		}
		else { len = le32(f + pos + 4); pos += hdr; }
		if(rf64 && 0 == memcmp(id, "ds64", 4)) {
			if(got_ds64) return fail(res, FLACAMD_E_INVALID_ARGUMENT, "file has multiple 'ds64' chunks");
			if(got_fmt) return fail(res, FLACAMD_E_INVALID_ARGUMENT, "'ds64' chunk appears after 'fmt ' or 'data' chunk");
			if(len < 28 || len > size - pos) return fail(res, FLACAMD_E_INVALID_ARGUMENT, "non-standard 'ds64' chunk has length = %u", (unsigned)len);
			ds64_data = le64(f + pos + 8);
			got_ds64 = 1;
			pos = skip_chunk(pos, size, len, len & 1);
		}
		else if(w64 ? 0 == memcmp(id, W64_FMT, 16) : 0 == memcmp(id, "fmt ", 4)) {
			unsigned tag; const uint64_t body = len;
			if(got_fmt) return fail(res, FLACAMD_E_INVALID_ARGUMENT, "file has multiple 'fmt ' chunks");
			if(body < 16 || body > size - pos) return fail(res, FLACAMD_E_INVALID_ARGUMENT, "non-standard 'fmt ' chunk has length = %u", (unsigned)body);
			tag = le16(f + pos);
			if(tag != 1 && tag != 65534) return fail(res, FLACAMD_E_UNSUPPORTED, "unsupported format type %u", tag);
			in->channels = le16(f + pos + 2); in->sample_rate = le32(f + pos + 4); in->bps = le16(f + pos + 14);
			in->is_unsigned = in->bps <= 8; in->shift = 0; in->channel_mask = 0;
			if(tag == 1) {
				if(in->bps != 8 && in->bps != 16 && in->bps != 24 && in->bps != 32) return fail(res, FLACAMD_E_UNSUPPORTED, "legacy WAVE file has format type 1 but bits-per-sample=%u", in->bps);
				if(in->channels > 2) return fail(res, FLACAMD_E_UNSUPPORTED, "WAVE has >2 channels but is not WAVE_FORMAT_EXTENSIBLE; cannot assign channels");
			}
			else {
				unsigned valid; uint32_t mask;
				if(body < 40) return fail(res, FLACAMD_E_INVALID_ARGUMENT, "invalid WAVEFORMATEXTENSIBLE chunk with size %u", (unsigned)body);
				if(le16(f + pos + 16) < 22) return fail(res, FLACAMD_E_INVALID_ARGUMENT, "invalid WAVEFORMATEXTENSIBLE chunk with cbSize %u", le16(f + pos + 16));
				valid = le16(f + pos + 18);
				if(valid > in->bps) return fail(res, FLACAMD_E_INVALID_ARGUMENT, "invalid WAVEFORMATEXTENSIBLE chunk with wValidBitsPerSample (%u) > wBitsPerSample (%u)", valid, in->bps);
				in->shift = in->bps - valid;
				mask = le32(f + pos + 20);
				if(mask == 0) mask = in->channels == 1 ? 0x0001 : in->channels == 2 ? 0x0003 : 0;
				if(!(mask == 0x0001 || mask == 0x0003 || mask == 0x0007 || mask == 0x0033 || mask == 0x0603 || mask == 0x0037 || mask == 0x0607 || mask == 0x003f || mask == 0x060f))
					return fail(res, FLACAMD_E_UNSUPPORTED, "WAVEFORMATEXTENSIBLE chunk with unsupported channel mask=0x%04X", (unsigned)mask);
				if(popcount32(mask) < in->channels) return fail(res, FLACAMD_E_INVALID_ARGUMENT, "channel mask 0x%04X has unassigned channels (#channels=%u)", (unsigned)mask, in->channels);
				if(popcount32(mask) > in->channels) return fail(res, FLACAMD_E_INVALID_ARGUMENT, "channel mask 0x%04X has extra bits for non-existant channels (#channels=%u)", (unsigned)mask, in->channels);
				if(le16(f + pos + 24) != 1) return fail(res, FLACAMD_E_UNSUPPORTED, "unsupported WAVEFORMATEXTENSIBLE chunk with non-PCM format %u", le16(f + pos + 24));
				in->channel_mask = mask;
			}
			got_fmt = 1;
			pos = skip_chunk(pos, size, body, w64 ? (8 - (body & 7)) & 7 : body & 1);
		}
		else if(w64 ? 0 == memcmp(id, W64_DATA, 16) : 0 == memcmp(id, "data", 4)) {
			if(!got_fmt) return fail(res, FLACAMD_E_INVALID_ARGUMENT, "got 'data' chunk before 'fmt' chunk");
			if(rf64) {
				if(!got_ds64) return fail(res, FLACAMD_E_INVALID_ARGUMENT, "RF64 file has no 'ds64' chunk before 'data' chunk");
				if(len == 0xffffffffu) len = ds64_data;
			}
			if(len == 0) return fail(res, FLACAMD_E_INVALID_ARGUMENT, "'data' chunk has size of 0");
			in->data = f + pos; in->data_bytes = len;
			in->available_bytes = size - pos < len ? size - pos : len;
			return 0;
		}
		else pos = skip_chunk(pos, size, len, w64 ? 0 : len & 1); /* unknown chunk: skipped (E:606-660) */
	}
	return fail(res, FLACAMD_E_INVALID_ARGUMENT, got_fmt ? "didn't find data chunk" : "didn't find fmt chunk");
}

/* AIFF / AIFF-C (E:681-907) */
static int parse_aiff(const uint8_t *f, uint64_t size, int is_aifc, pcm_input *in, flacamd_file_result *res)
{
	uint64_t pos = 12, frames = 0; int got_comm = 0;
	in->big_endian = 1; in->is_unsigned = 0; in->channel_mask = 0;
	while(pos + 4 <= size) {
		const uint8_t *id = f + pos;
		uint32_t len;
		if(pos + 8 > size) return fail(res, FLACAMD_E_INVALID_ARGUMENT, "incomplete chunk header");
		len = be32(f + pos + 4);
		pos += 8;
		if(0 == memcmp(id, "COMM", 4)) {
			const uint32_t min_len = is_aifc ? 22u : 18u;
			const uint8_t *p = f + pos; int e; uint64_t mant = 0; unsigned i, sh;
			if(got_comm) return fail(res, FLACAMD_E_INVALID_ARGUMENT, "file has multiple 'COMM' chunks");
			if(len < min_len || pos + len > size) return fail(res, FLACAMD_E_INVALID_ARGUMENT, "non-standard 'COMM' chunk has length = %u", len);
			in->channels = be16(p); frames = be32(p + 2); in->bps = be16(p + 6);
			if(in->channels > 2) return fail(res, FLACAMD_E_UNSUPPORTED, "unsupported number of channels %u for AIFF", in->channels);
			in->shift = (in->bps % 8) ? 8 - (in->bps % 8) : 0; in->bps += in->shift;   /* SSND data is byte-aligned, left-justified */
			/* 80-bit extended sample rate (E:2789-2817) */
			e = (int)be16(p + 8) - 0x3FFF;
			if((p[8] >> 7) == 1 || e < 0 || e > 63) return fail(res, FLACAMD_E_INVALID_ARGUMENT, "invalid floating-point value");
			for(i = 0; i < 8; i++) mant |= (uint64_t)p[10 + i] << (56 - 8 * i);
			sh = (unsigned)(63 - e);
			in->sample_rate = (uint32_t)((sh < 64 ? mant >> sh : 0) + (sh ? (mant >> (sh - 1)) & 1 : 0));
			if(is_aifc) {
				const uint32_t comp = be32(p + 18);
				if(comp == 0x736F7774) in->big_endian = 0;            /* "sowt" */
				else if(comp != 0x4E4F4E45) return fail(res, FLACAMD_E_UNSUPPORTED, "can't handle AIFF-C compression type");   /* "NONE" */
			}
			got_comm = 1;
			pos += (uint64_t)len + (len & 1);
		}
		else if(0 == memcmp(id, "SSND", 4)) {
			uint64_t bytes; uint32_t offset, block;
			if(!got_comm) return fail(res, FLACAMD_E_INVALID_ARGUMENT, "got 'SSND' chunk before 'COMM' chunk");
			if(len <= 8 || pos + 8 > size) return fail(res, FLACAMD_E_INVALID_ARGUMENT, "'SSND' chunk has size <= 8");
			bytes = (uint64_t)len - 8;
			offset = be32(f + pos); block = be32(f + pos + 4);
			if(offset > bytes) return fail(res, FLACAMD_E_INVALID_ARGUMENT, "'SSND' offset beyond the chunk");
			bytes -= offset;
			if(block) bytes -= (block - (bytes % block));
			pos += 8 + (uint64_t)offset;
			if(pos > size) return fail(res, FLACAMD_E_INVALID_ARGUMENT, "skipping offset in SSND chunk");
			in->data = f + pos; in->data_bytes = bytes;
			in->available_bytes = size - pos < bytes ? size - pos : bytes;
			return 0;
		}
		else pos += (uint64_t)len + (len & 1);
	}
	if(!got_comm) return fail(res, FLACAMD_E_INVALID_ARGUMENT, "didn't find COMM chunk");
	if(frames) return fail(res, FLACAMD_E_INVALID_ARGUMENT, "didn't find SSND chunk");
	in->data = f; in->data_bytes = 0; in->available_bytes = 0;
	return 0;
}

/* format_input (E:2338-2428): n wide samples of container bytes -> interleaved int32 */
static int format_samples(const pcm_input *in, const uint8_t *src, size_t n, int32_t *dst, flacamd_file_result *res)
{
	const size_t total = n * in->channels; size_t i;
	if(in->bps == 8) {
		if(in->is_unsigned) for(i = 0; i < total; i++) dst[i] = (int32_t)src[i] - 0x80;
		else for(i = 0; i < total; i++) dst[i] = (int8_t)src[i];
	}
	else if(in->bps == 16) {
		for(i = 0; i < total; i++) {
			const unsigned v = in->big_endian ? be16(src + 2 * i) : le16(src + 2 * i);
			dst[i] = in->is_unsigned ? (int32_t)v - 0x8000 : (int16_t)v;
		}
	}
	else if(in->bps == 24) {
		for(i = 0; i < total; i++) {
			const uint8_t *p = src + 3 * i;
			const uint32_t v = in->big_endian ? ((uint32_t)p[0] << 16) | ((uint32_t)p[1] << 8) | p[2] : ((uint32_t)p[2] << 16) | ((uint32_t)p[1] << 8) | p[0];
			dst[i] = in->is_unsigned ? (int32_t)v - 0x800000 : (int32_t)(v << 8) >> 8;
		}
	}
	else return fail(res, FLACAMD_E_UNSUPPORTED, "unsupported container width %u", in->bps);
	if(in->shift) {
		const int32_t mask = (1 << in->shift) - 1;
		for(i = 0; i < total; i++) {
			if(dst[i] & mask) return fail(res, FLACAMD_E_INVALID_ARGUMENT, "sample data (channel#%u sample#%u = %d) has non-zero least-significant bits", (unsigned)(i % in->channels), (unsigned)(i / in->channels), dst[i]);
			dst[i] >>= in->shift;
		}
	}
	return 0;
}

/* ---- seek table template (src/share/grabbag/seektable.c:54-132, src/libFLAC/metadata_object.c:1049-1139) ---- */
typedef struct { FLAC__StreamMetadata_SeekPoint *pts; unsigned n, cap; int failed; } point_list;
static void add_point(point_list *l, FLAC__uint64 sample)
{
	if(l->n == l->cap) {
		const unsigned cap = l->cap ? l->cap * 2 : 64;
		FLAC__StreamMetadata_SeekPoint *p = (FLAC__StreamMetadata_SeekPoint*)realloc(l->pts, sizeof *p * cap);
		if(!p) { l->failed = 1; return; }
		l->pts = p; l->cap = cap;
	}
	l->pts[l->n].sample_number = sample; l->pts[l->n].stream_offset = 0; l->pts[l->n].frame_samples = 0; l->n++;
}
static void seek_template(const char *spec, FLAC__uint64 total, unsigned rate, point_list *l)
{
	const char *pt = spec;
	while(pt && *pt) {
		const char *q = strchr(pt, ';');
		if(!q) break;
		if(q > pt) {
			if(0 == strncmp(pt, "X;", 2)) add_point(l, FLAC__STREAM_METADATA_SEEKPOINT_PLACEHOLDER);
			else if(q[-1] == 'x') { /* N evenly spaced points */
				if(total > 0) {
					const int num = atoi(pt); int i;
					for(i = 0; i < num; i++) add_point(l, total * (FLAC__uint64)i / (FLAC__uint64)num);
				}
			}
			else if(q[-1] == 's') { /* a point every so many seconds */
				if(total > 0) {
					const double sec = atof(pt);
					if(sec > 0.0) {
						const unsigned step = (unsigned)(sec * (double)rate);
						if(step > 0) {
							FLAC__uint64 num = 1 + total / step, j;
							if(total % step == 0) num--;
							for(j = 0; j < num; j++) add_point(l, j * step);
						}
					}
				}
			}
			else { /* an explicit sample number */
				char *end;
				const long long v = strtoll(pt, &end, 10);
				if((v > 0 || (end > pt && *end == ';')) && (total == 0 || (FLAC__uint64)v < total)) add_point(l, (FLAC__uint64)v);
			}
		}
		pt = q + 1;
	}
	if(l->n) { /* sort + compact, as the template sort with compact=true does */
		FLAC__StreamMetadata_SeekTable t; t.num_points = l->n; t.points = l->pts;
		l->n = FLAC__format_seektable_sort(&t);
	}
}

/* ---- encode ---- */
/* --skip / --until values: "123456" (samples) or "mm:ss.sss", --until also "+..." (after the skip point) and "-..." (before the
 * end): flac__utils_parse_skip_until_specification and flac__utils_canonicalize_skip_until_specification (src/flac/utils.c:36-76, :170-219) */
typedef struct { int is_relative, value_is_samples; int64_t samples; double seconds; } range_spec;

static int parse_range_spec(const char *s, range_spec *sp)
{
	int negative = 0; const char *p; uint64_t v = 0; int all_digits;
	sp->is_relative = 0; sp->value_is_samples = 1; sp->samples = 0; sp->seconds = 0.0;
	if(!s) return 1;
	if(*s == '-') { negative = 1; sp->is_relative = 1; s++; }
	else if(*s == '+') { sp->is_relative = 1; s++; }
	all_digits = *s != 0;
	for(p = s; *p; p++) { if(*p < '0' || *p > '9') { all_digits = 0; break; } v = v * 10 + (uint64_t)(*p - '0'); }
	if(all_digits) sp->samples = negative ? -(int64_t)v : (int64_t)v;
	else { /* [0-9]+:[0-9]*[.,]?[0-9]* */
		unsigned minutes; char *end; double sec;
		if(*s < '0' || *s > '9') return 0;
		minutes = (unsigned)(*s++ - '0');
		while(*s != ':') { if(*s < '0' || *s > '9') return 0; minutes = minutes * 10 + (unsigned)(*s++ - '0'); }
		s++;
		if(strspn(s, "1234567890.,") != strlen(s)) return 0;
		sec = strtod(s, &end);
		if(end == s || *end) return 0;
		sp->value_is_samples = 0; sp->seconds = (double)minutes * 60. + sec;
		if(negative) sp->seconds = -sp->seconds;
	}
	return 1;
}

static void canonicalize_range_spec(range_spec *sp, unsigned sample_rate)
{
	if(!sp->value_is_samples) { sp->samples = (int64_t)(sp->seconds * (double)sample_rate); sp->value_is_samples = 1; }
}

/* canonicalize_until_specification (src/flac/encode.c:2246-2289, src/flac/decode.c:498-541): 0 = to the end */
static int canonicalize_until(range_spec *sp, unsigned sample_rate, uint64_t skip, uint64_t total_in, flacamd_file_result *res)
{
	canonicalize_range_spec(sp, sample_rate);
	if(sp->is_relative && sp->samples == 0) { sp->is_relative = 0; return 0; }
	if(total_in == 0) return fail(res, FLACAMD_E_INVALID_ARGUMENT, "cannot use --until when input length is unknown");
	if(sp->is_relative) { if(sp->samples <= 0) sp->samples += (int64_t)total_in; else sp->samples += (int64_t)skip; sp->is_relative = 0; }
	if(sp->samples < 0) return fail(res, FLACAMD_E_INVALID_ARGUMENT, "--until value is before beginning of input");
	if((uint64_t)sp->samples <= skip) return fail(res, FLACAMD_E_INVALID_ARGUMENT, "--until value is before --skip point");
	if((uint64_t)sp->samples > total_in) return fail(res, FLACAMD_E_INVALID_ARGUMENT, "--until value is after end of input");
	return 0;
}

/* the two option strings as the tool's main() takes them (src/flac/main.c:1842-1853): --skip absolute, --until defaults to "-0" */
static int parse_range_options(const char *skip, const char *until, range_spec *sk, range_spec *un, flacamd_file_result *res)
{
	if(!parse_range_spec(skip, sk) || sk->is_relative) return fail(res, FLACAMD_E_INVALID_ARGUMENT, "invalid value for --skip");
	if(!parse_range_spec(until, un)) return fail(res, FLACAMD_E_INVALID_ARGUMENT, "invalid value for --until");
	if(!until) un->is_relative = 1;
	return 0;
}

/* The tag block of the command line (-T values), with the channel mask tag set when the input names a mask: the first
 * tag of that name is replaced where it stands, later ones are dropped, none: appended (src/flac/utils.c:279-297,
 * FLAC__metadata_object_vorbiscomment_replace_comment with all=true).  *entries_out is the caller's to free. */
static int build_tags(const flacamd_file_options *opt, uint32_t channel_mask, FLAC__StreamMetadata *tags, FLAC__StreamMetadata_VorbisComment_Entry **entries_out,
	char *mask_tag, size_t mask_tag_cap, flacamd_file_result *res)
{
	unsigned i, n = opt->num_tags, length = 8;
	FLAC__StreamMetadata_VorbisComment_Entry *entries = (FLAC__StreamMetadata_VorbisComment_Entry*)calloc(n + 1, sizeof *entries);
	if(!entries) return fail(res, FLACAMD_E_OUT_OF_MEMORY, "allocating memory for tags");
	*entries_out = entries;
	for(i = 0; i < n; i++) {
		const size_t l = strlen(opt->tags[i]);
		if(!FLAC__format_vorbiscomment_entry_is_legal((const FLAC__byte*)opt->tags[i], (unsigned)l)) return fail(res, FLACAMD_E_INVALID_ARGUMENT, "illegal tag \"%s\"", opt->tags[i]);
		entries[i].length = (FLAC__uint32)l; entries[i].entry = (FLAC__byte*)opt->tags[i];
		length += 4 + (unsigned)l;
	}
	if(channel_mask) {
		const size_t name = strlen(CHANNEL_MASK_TAG);
		unsigned w = 0; int placed = 0;
		snprintf(mask_tag, mask_tag_cap, "%s=0x%04X", CHANNEL_MASK_TAG, (unsigned)channel_mask);
		for(i = 0; i < n; i++) {
			const int same = entries[i].length > name && entries[i].entry[name] == '=' && 0 == strncasecmp((const char*)entries[i].entry, CHANNEL_MASK_TAG, name);
			if(!same) { entries[w++] = entries[i]; continue; }
			length -= 4 + entries[i].length;
			if(!placed) { entries[w].length = (FLAC__uint32)strlen(mask_tag); entries[w].entry = (FLAC__byte*)mask_tag; length += 4 + entries[w].length; w++; placed = 1; }
		}
		n = w;
		if(!placed) { entries[n].length = (FLAC__uint32)strlen(mask_tag); entries[n].entry = (FLAC__byte*)mask_tag; length += 4 + entries[n].length; n++; }
	}
	memset(tags, 0, sizeof *tags);
	tags->type = FLAC__METADATA_TYPE_VORBIS_COMMENT; tags->length = length;
	tags->data.vorbis_comment.num_comments = n; tags->data.vorbis_comment.comments = n ? entries : 0;
	return 0;
}

/* WAVEFORMATEXTENSIBLE_CHANNEL_MASK=0x... of a tag block: flac__utils_get_channel_mask_tag (src/flac/utils.c:299-321) */
static uint32_t channel_mask_of(const FLAC__StreamMetadata *m)
{
	const size_t name = strlen(CHANNEL_MASK_TAG); unsigned i;
	for(i = 0; i < m->data.vorbis_comment.num_comments; i++) {
		const FLAC__StreamMetadata_VorbisComment_Entry *e = &m->data.vorbis_comment.comments[i];
		if(e->length > name + 3 && 0 == strncasecmp((const char*)e->entry, CHANNEL_MASK_TAG, name) && e->entry[name] == '=' && 0 == strncmp((const char*)e->entry + name + 1, "0x", 2)) {
			unsigned v;
			return sscanf((const char*)e->entry + name + 3, "%x", &v) == 1 ? v : 0;
		}
	}
	return 0;
}

/* ---- FLAC input: decode on the device, code again, keep the metadata the way the tool does (E:907-958, :1746-2015, :2524-2557) ---- */
#define MAX_INPUT_BLOCKS 64   /* FLACDecoderData.metadata_blocks[] */
typedef struct { FLAC__StreamEncoder *enc; FLAC__uint64 left; int failed; flacamd_file_result *res; } recode_session;

static FLAC__StreamDecoderWriteStatus recode_write(const FLAC__StreamDecoder *d, const FLAC__Frame *fr, const FLAC__int32 * const buffer[], void *cd)
{
	recode_session *r = (recode_session*)cd;
	const FLAC__uint64 n = r->left < fr->header.blocksize ? r->left : fr->header.blocksize;
	(void)d;
	if(n && !FLAC__stream_encoder_process(r->enc, buffer, (unsigned)n)) {
		r->failed = 1; fail(r->res, FLACAMD_E_DEVICE, "during encoding: %s", FLAC__stream_encoder_get_resolved_state_string(r->enc));
		return FLAC__STREAM_DECODER_WRITE_STATUS_ABORT;
	}
	r->left -= n;
	return FLAC__STREAM_DECODER_WRITE_STATUS_CONTINUE;
}
static void recode_error(const FLAC__StreamDecoder *d, FLAC__StreamDecoderErrorStatus st, void *cd)
{
	recode_session *r = (recode_session*)cd; (void)d;
	r->failed = 1; fail(r->res, FLACAMD_E_BAD_STREAM, "got %s while decoding FLAC input", FLAC__StreamDecoderErrorStatusString[st]);
}

static int encode_flac_input(const uint8_t *map, uint64_t size, const char *in_path, const char *out_path, const flacamd_file_options *opt, flacamd_file_result *res)
{
	FLAC__StreamMetadata in_blocks[MAX_INPUT_BLOCKS], ours_tags, ours_seek, ours_pad, *list[MAX_INPUT_BLOCKS + 4];
	unsigned nin = 0, nlist = 0, i, j, channels, bps, rate; uint64_t pos = 4, total_in, total, skip, until;
	FLAC__StreamMetadata_VorbisComment_Entry *entries = 0; char mask_tag[64]; point_list points = { 0, 0, 0, 0 };
	FLAC__StreamEncoder *enc = 0; FLAC__StreamDecoder *dec = 0; recode_session rs; int rc = 0, last = 0, have_tags = 0, have_seek = 0; long pad_sum = -1;
	uint32_t channel_mask = 0;
	range_spec sk, un;

	/* the blocks of the input, STREAMINFO first (E:940-951) */
	if(size < 4 + 4 + 34 || (map[4] & 0x7f) != FLAC__METADATA_TYPE_STREAMINFO || ((uint32_t)map[5] << 16 | (uint32_t)map[6] << 8 | map[7]) < 34)
		return fail(res, FLACAMD_E_BAD_STREAM, "reading metadata in FLAC input, first metadata block is not STREAMINFO");
	{
		const uint8_t *si = map + 8;
		rate = (unsigned)si[10] << 12 | (unsigned)si[11] << 4 | si[12] >> 4;
		channels = ((si[12] >> 1) & 7u) + 1u; bps = (((unsigned)si[12] & 1u) << 4 | si[13] >> 4) + 1u;
		total_in = ((uint64_t)(si[13] & 15u) << 32) | be32(si + 14);
	}
	if(total_in == 0) return fail(res, FLACAMD_E_UNSUPPORTED, "FLAC input has STREAMINFO with unknown total samples which is not supported");
	/* the header is the file's word, not yet checked by any encoder or decoder: the rate divides sample counts below */
	if(rate == 0 || rate > 655350) return fail(res, FLACAMD_E_UNSUPPORTED, "unsupported sample rate %u", rate);
	if(bps < 4 || bps > 24) return fail(res, FLACAMD_E_UNSUPPORTED, "unsupported bits-per-sample %u", bps);
	while(!last) {
		uint32_t len; unsigned type;
		if(pos + 4 > size) { rc = fail(res, FLACAMD_E_BAD_STREAM, "reading metadata in FLAC input"); goto done; }
		last = map[pos] >> 7; type = map[pos] & 0x7f; len = (uint32_t)map[pos + 1] << 16 | (uint32_t)map[pos + 2] << 8 | map[pos + 3];
		pos += 4;
		if(pos + len > size) { rc = fail(res, FLACAMD_E_BAD_STREAM, "reading metadata in FLAC input"); goto done; }
		if(type != FLAC__METADATA_TYPE_STREAMINFO) {
			if(nin + 1 >= MAX_INPUT_BLOCKS) { rc = fail(res, FLACAMD_E_UNSUPPORTED, "out of memory or too many metadata blocks while reading metadata in FLAC input"); goto done; }
			if(flacamd_metadata_block_parse(type, last, map + pos, len, &in_blocks[nin]) != 0) { rc = fail(res, FLACAMD_E_BAD_STREAM, "reading metadata in FLAC input"); goto done; }
			nin++;
		}
		pos += len;
	}

	/* the sample range (E:1083, :1091-1150) */
	if((rc = parse_range_options(opt->skip, opt->until, &sk, &un, res)) != 0) goto done;
	canonicalize_range_spec(&sk, rate);
	skip = (uint64_t)sk.samples;
	if((rc = canonicalize_until(&un, rate, skip, total_in, res)) != 0) goto done;
	until = (uint64_t)un.samples;
	if(skip >= total_in) { rc = fail(res, FLACAMD_E_INVALID_ARGUMENT, "while skipping samples"); goto done; }
	total = total_in - skip - (until > 0 ? total_in - until : 0);

	/* tags: the command line's replace the input's, which hand on their channel mask; none in the input: an empty block (E:1810-1849) */
	for(i = 0; i < nin; i++) if(in_blocks[i].type == FLAC__METADATA_TYPE_VORBIS_COMMENT) { have_tags = 1; if(opt->num_tags > 0) channel_mask = channel_mask_of(&in_blocks[i]); }
	/* seek table: -S- drops it, -S values replace it, an unchanged length keeps the input's, else the default (E:1906-1957) */
	for(i = 0; i < nin; i++) if(in_blocks[i].type == FLAC__METADATA_TYPE_SEEKTABLE && !(total != total_in || opt->seekpoints)) have_seek = 1;
	if((opt->seekpoints && opt->seekpoints[0]) || (!opt->seekpoints && !have_seek)) {
		seek_template(opt->seekpoints ? opt->seekpoints : "10s;", total, rate, &points);
		if(points.failed) { rc = fail(res, FLACAMD_E_OUT_OF_MEMORY, "allocating memory for seek table"); goto done; }
		memset(&ours_seek, 0, sizeof ours_seek);
		ours_seek.type = FLAC__METADATA_TYPE_SEEKTABLE; ours_seek.length = points.n * FLAC__STREAM_METADATA_SEEKPOINT_LENGTH;
		ours_seek.data.seek_table.num_points = points.n; ours_seek.data.seek_table.points = points.pts;
		list[nlist++] = &ours_seek;
	}
	if(!have_tags || opt->num_tags > 0) {
		if((rc = build_tags(opt, channel_mask, &ours_tags, &entries, mask_tag, sizeof mask_tag, res)) != 0) goto done;
		list[nlist++] = &ours_tags;
	}
	for(i = 0; i < nin; i++) {
		const FLAC__StreamMetadata *m = &in_blocks[i];
		if(m->type == FLAC__METADATA_TYPE_VORBIS_COMMENT && opt->num_tags > 0) continue;
		if(m->type == FLAC__METADATA_TYPE_SEEKTABLE && !have_seek) continue;
		if(m->type == FLAC__METADATA_TYPE_CUESHEET && (m->data.cue_sheet.num_tracks == 0 || total != m->data.cue_sheet.tracks[m->data.cue_sheet.num_tracks - 1].offset)) continue;   /* lead-out no longer at the end (E:1856-1893) */
		if(m->type == FLAC__METADATA_TYPE_PADDING) { if(pad_sum < 0) pad_sum = 0; pad_sum += (long)m->length; continue; }   /* all padding becomes one block at the end (E:1960-2002) */
		list[nlist++] = &in_blocks[i];
	}
	if(opt->padding > 0) pad_sum = opt->padding;
	if(pad_sum < 0) pad_sum = total / rate < 20 * 60 ? DEFAULT_PADDING : DEFAULT_PADDING * 8;
	if(opt->padding != 0 && pad_sum > 0) {
		memset(&ours_pad, 0, sizeof ours_pad);
		ours_pad.type = FLAC__METADATA_TYPE_PADDING; ours_pad.length = (unsigned)pad_sum;
		list[nlist++] = &ours_pad;
	}

	enc = FLAC__stream_encoder_new();
	dec = FLAC__stream_decoder_new();
	if(!enc || !dec) { rc = fail(res, FLACAMD_E_OUT_OF_MEMORY, "creating the encoder instance"); goto done; }
	FLAC__stream_encoder_set_verify(enc, opt->verify);
	FLAC__stream_encoder_set_streamable_subset(enc, !opt->lax);
	FLAC__stream_encoder_set_channels(enc, channels);
	FLAC__stream_encoder_set_bits_per_sample(enc, bps);
	FLAC__stream_encoder_set_sample_rate(enc, rate);
	FLAC__stream_encoder_set_compression_level(enc, opt->compression_level);
	if(opt->blocksize) FLAC__stream_encoder_set_blocksize(enc, opt->blocksize);
	FLAC__stream_encoder_set_total_samples_estimate(enc, total);
	FLAC__stream_encoder_set_metadata(enc, nlist ? list : 0, nlist);
	flacamd_stream_encoder_set_device(enc, opt->device);
	flacamd_stream_encoder_set_batch_frames(enc, opt->batch_frames ? opt->batch_frames : 4096u);
	{
		const FLAC__StreamEncoderInitStatus is = FLAC__stream_encoder_init_file(enc, out_path, 0, 0);
		if(is != FLAC__STREAM_ENCODER_INIT_STATUS_OK) {
			rc = fail(res, is == FLAC__STREAM_ENCODER_INIT_STATUS_ENCODER_ERROR ? FLACAMD_E_DEVICE : FLACAMD_E_INVALID_ARGUMENT, "initializing encoder: %s (%s)",
				FLAC__StreamEncoderInitStatusString[is], FLAC__stream_encoder_get_resolved_state_string(enc));
			goto done;
		}
	}
	memset(&rs, 0, sizeof rs);
	rs.enc = enc; rs.left = total; rs.res = res;
	FLAC__stream_decoder_set_md5_checking(dec, 0);
	flacamd_stream_decoder_set_device(dec, opt->device);
	if(FLAC__stream_decoder_init_file(dec, in_path, recode_write, 0, recode_error, &rs) != FLAC__STREAM_DECODER_INIT_STATUS_OK) { rc = fail(res, FLACAMD_E_INVALID_ARGUMENT, "initializing decoder for FLAC input, state = %s", FLAC__stream_decoder_get_resolved_state_string(dec)); goto done; }
	if(!FLAC__stream_decoder_process_until_end_of_metadata(dec)) { rc = fail(res, FLACAMD_E_BAD_STREAM, "reading metadata in FLAC input, state = %s", FLAC__stream_decoder_get_resolved_state_string(dec)); goto done; }
	if(skip > 0 && !FLAC__stream_decoder_seek_absolute(dec, skip)) { rc = fail(res, FLACAMD_E_BAD_STREAM, "while skipping samples, FLAC decoder state = %s", FLAC__stream_decoder_get_resolved_state_string(dec)); goto done; }   /* E:1247 */
	while(!rs.failed && rs.left > 0) { /* E:1440-1455 */
		if(!FLAC__stream_decoder_process_single(dec) || FLAC__stream_decoder_get_state(dec) == FLAC__STREAM_DECODER_END_OF_STREAM) {
			if(rs.left > 0 && !rs.failed) rc = fail(res, FLACAMD_E_BAD_STREAM, "while decoding FLAC input, state = %s", FLAC__stream_decoder_get_resolved_state_string(dec));
			break;
		}
	}
	if(rs.failed && !rc) rc = FLACAMD_E_BAD_STREAM;
	if(rc) goto done;
	if(!FLAC__stream_encoder_finish(enc)) { rc = fail(res, FLACAMD_E_DEVICE, "finishing the stream: %s", FLAC__stream_encoder_get_resolved_state_string(enc)); goto done; }
	if(res) {
		struct stat os;
		res->samples = total - rs.left; res->channels = channels; res->bits_per_sample = bps; res->sample_rate = rate; res->md5_ok = 1;
		if(stat(out_path, &os) == 0) res->bytes_written = (uint64_t)os.st_size;
	}
done:
	if(dec) { (void)FLAC__stream_decoder_finish(dec); FLAC__stream_decoder_delete(dec); }
	if(enc) { if(rc) (void)FLAC__stream_encoder_finish(enc); FLAC__stream_encoder_delete(enc); }
	for(j = 0; j < nin; j++) flacamd_metadata_block_free(&in_blocks[j]);
	free(entries); free(points.pts);
	return rc;
}

int flacamd_encode_file(const char *in_path, const char *out_path, const flacamd_file_options *opt, flacamd_file_result *res)
{
	flacamd_file_options defaults;
	pcm_input in;
	int fd = -1, rc = 0; struct stat st; const uint8_t *map = 0; uint64_t size = 0;
	FLAC__StreamEncoder *enc = 0;
	FLAC__StreamMetadata seektable, tags, padding, *blocks[3]; unsigned nblocks = 0;
	FLAC__StreamMetadata_VorbisComment_Entry *entries = 0; char mask_tag[64];
	point_list points = { 0, 0, 0, 0 };
	int32_t *buf = 0;
	uint64_t total, bytes_per_wide, fed = 0;
	int format;

	if(res) memset(res, 0, sizeof *res);
	if(!in_path || !out_path) return fail(res, FLACAMD_E_INVALID_ARGUMENT, "missing path");
	if(!opt) { flacamd_file_options_init(&defaults); opt = &defaults; }
	memset(&in, 0, sizeof in);

	fd = open(in_path, O_RDONLY);
	if(fd < 0 || fstat(fd, &st) != 0) { if(fd >= 0) close(fd); return fail(res, FLACAMD_E_INVALID_ARGUMENT, "can't open input file %s", in_path); }
	size = (uint64_t)st.st_size;
	if(size) {
		map = (const uint8_t*)mmap(0, (size_t)size, PROT_READ, MAP_PRIVATE, fd, 0);
		if(map == MAP_FAILED) { close(fd); return fail(res, FLACAMD_E_OUT_OF_MEMORY, "can't map input file %s", in_path); }
		(void)madvise((void*)map, (size_t)size, MADV_SEQUENTIAL);
	}

	format = opt->input_format;
	if(format == FLACAMD_FILE_AUTO) {
		if(size >= 12 && 0 == memcmp(map, "RIFF", 4) && 0 == memcmp(map + 8, "WAVE", 4)) format = FLACAMD_FILE_WAVE;
		else if(size >= 12 && 0 == memcmp(map, "RF64", 4) && 0 == memcmp(map + 8, "WAVE", 4)) format = FLACAMD_FILE_RF64;
		else if(size >= 12 && 0 == memcmp(map, W64_RIFF, 12)) format = FLACAMD_FILE_WAVE64;   /* the tool looks at 12 bytes of the GUID (main.c:1733) */
		else if(size >= 4 && 0 == memcmp(map, "fLaC", 4)) format = FLACAMD_FILE_FLAC;
		else if(size >= 12 && 0 == memcmp(map, "FORM", 4) && (0 == memcmp(map + 8, "AIFF", 4) || 0 == memcmp(map + 8, "AIFC", 4))) format = FLACAMD_FILE_AIFF;
		else { rc = fail(res, FLACAMD_E_UNSUPPORTED, "input is neither WAVE, RF64, Wave64 nor AIFF; raw input needs input_format = RAW and its parameters"); goto done; }
	}
	if(format == FLACAMD_FILE_FLAC) {
		if(size < 4 || memcmp(map, "fLaC", 4)) { rc = fail(res, FLACAMD_E_INVALID_ARGUMENT, "not a FLAC file"); goto done; }
		rc = encode_flac_input(map, size, in_path, out_path, opt, res);
		goto done;
	}
	if(format == FLACAMD_FILE_WAVE) {
		if(size < 12 || memcmp(map, "RIFF", 4) || memcmp(map + 8, "WAVE", 4)) { rc = fail(res, FLACAMD_E_INVALID_ARGUMENT, "not a RIFF WAVE file"); goto done; }
		rc = parse_wave(map, size, FLACAMD_FILE_WAVE, &in, res);
	}
	else if(format == FLACAMD_FILE_RF64) {
		if(size < 12 || memcmp(map, "RF64", 4) || memcmp(map + 8, "WAVE", 4)) { rc = fail(res, FLACAMD_E_INVALID_ARGUMENT, "not an RF64 file"); goto done; }
		rc = parse_wave(map, size, FLACAMD_FILE_RF64, &in, res);
	}
	else if(format == FLACAMD_FILE_WAVE64) {
		if(size < 40 || memcmp(map, W64_RIFF, 12)) { rc = fail(res, FLACAMD_E_INVALID_ARGUMENT, "not a Wave64 file"); goto done; }
		rc = parse_wave(map, size, FLACAMD_FILE_WAVE64, &in, res);
	}
	else if(format == FLACAMD_FILE_AIFF) {
		if(size < 12 || memcmp(map, "FORM", 4) || (memcmp(map + 8, "AIFF", 4) && memcmp(map + 8, "AIFC", 4))) { rc = fail(res, FLACAMD_E_INVALID_ARGUMENT, "not an AIFF file"); goto done; }
		rc = parse_aiff(map, size, 0 == memcmp(map + 8, "AIFC", 4), &in, res);
	}
	else if(format == FLACAMD_FILE_RAW) { /* E:188-200 */
		in.channels = opt->raw_channels; in.bps = opt->raw_bps; in.sample_rate = opt->raw_sample_rate; in.shift = 0;
		in.big_endian = opt->raw_big_endian; in.is_unsigned = opt->raw_unsigned; in.channel_mask = 0;
		if(in.channels == 0 || in.bps == 0) { rc = fail(res, FLACAMD_E_INVALID_ARGUMENT, "raw input needs channels, bps and sample rate"); goto done; }
		in.data = map; in.data_bytes = in.available_bytes = size - size % ((uint64_t)in.channels * ((in.bps + 7) / 8));
	}
	else rc = fail(res, FLACAMD_E_INVALID_ARGUMENT, "unknown input format %d", format);
	if(rc) goto done;

	/* E:1033-1044 */
	if(in.channels == 0 || in.channels > 8) { rc = fail(res, FLACAMD_E_UNSUPPORTED, "unsupported number of channels %u", in.channels); goto done; }
	if(in.sample_rate == 0 || in.sample_rate > 655350) { rc = fail(res, FLACAMD_E_UNSUPPORTED, "unsupported sample rate %u", in.sample_rate); goto done; }
	if(in.bps - in.shift < 4 || in.bps - in.shift > 24) { rc = fail(res, FLACAMD_E_UNSUPPORTED, "unsupported bits-per-sample %u", in.bps - in.shift); goto done; }
	bytes_per_wide = format == FLACAMD_FILE_RAW ? (uint64_t)in.channels * ((in.bps + 7) / 8) : (uint64_t)in.channels * (in.bps / 8);
	if(bytes_per_wide == 0 || (in.bps != 8 && in.bps != 16 && in.bps != 24)) { rc = fail(res, FLACAMD_E_UNSUPPORTED, "unsupported container width %u", in.bps); goto done; }
	total = in.data_bytes / bytes_per_wide;   /* total_samples_to_encode: what the header promises (E:1079) */
	{ /* --skip / --until: the sample range [skip, until) of the input (E:1091-1150) */
		range_spec sk, un; uint64_t skip, until, cut;
		if((rc = parse_range_options(opt->skip, opt->until, &sk, &un, res)) != 0) goto done;
		canonicalize_range_spec(&sk, in.sample_rate);
		skip = (uint64_t)sk.samples;
		if((rc = canonicalize_until(&un, in.sample_rate, skip, total, res)) != 0) goto done;
		until = (uint64_t)un.samples;
		if(skip > total) { rc = fail(res, FLACAMD_E_INVALID_ARGUMENT, "during read while skipping samples"); goto done; }
		if(until > 0) { in.data_bytes -= (total - until) * bytes_per_wide; total = until; }
		cut = skip * bytes_per_wide;
		in.data += cut; in.data_bytes -= cut; total -= skip;
		in.available_bytes = in.available_bytes > cut ? in.available_bytes - cut : 0;
		if(in.available_bytes > in.data_bytes) in.available_bytes = in.data_bytes;
	}

	/* metadata in the tool's order: seek table, tags, padding (E:2016-2053) */
	{
		const char *spec = opt->seekpoints ? opt->seekpoints : "10s;";
		seek_template(spec, total, in.sample_rate, &points);
		if(points.failed) { rc = fail(res, FLACAMD_E_OUT_OF_MEMORY, "allocating memory for seek table"); goto done; }
		if(points.n > 0) {
			memset(&seektable, 0, sizeof seektable);
			seektable.type = FLAC__METADATA_TYPE_SEEKTABLE; seektable.length = points.n * FLAC__STREAM_METADATA_SEEKPOINT_LENGTH;
			seektable.data.seek_table.num_points = points.n; seektable.data.seek_table.points = points.pts;
			blocks[nblocks++] = &seektable;
		}
	}
	if((rc = build_tags(opt, in.channel_mask, &tags, &entries, mask_tag, sizeof mask_tag, res)) != 0) goto done;
	blocks[nblocks++] = &tags;
	if(opt->padding != 0) {
		memset(&padding, 0, sizeof padding);
		padding.type = FLAC__METADATA_TYPE_PADDING;
		padding.length = opt->padding > 0 ? (unsigned)opt->padding : (total / in.sample_rate < 20 * 60 ? DEFAULT_PADDING : DEFAULT_PADDING * 8);
		blocks[nblocks++] = &padding;
	}

	/* encoder set-up in the tool's order (E:2066-2149) */
	enc = FLAC__stream_encoder_new();
	if(!enc) { rc = fail(res, FLACAMD_E_OUT_OF_MEMORY, "creating the encoder instance"); goto done; }
	FLAC__stream_encoder_set_verify(enc, opt->verify);
	FLAC__stream_encoder_set_streamable_subset(enc, !opt->lax);
	FLAC__stream_encoder_set_channels(enc, in.channels);
	FLAC__stream_encoder_set_bits_per_sample(enc, in.bps - in.shift);
	FLAC__stream_encoder_set_sample_rate(enc, in.sample_rate);
	FLAC__stream_encoder_set_compression_level(enc, opt->compression_level);
	if(opt->blocksize) FLAC__stream_encoder_set_blocksize(enc, opt->blocksize);
	FLAC__stream_encoder_set_total_samples_estimate(enc, total);
	FLAC__stream_encoder_set_metadata(enc, nblocks ? blocks : 0, nblocks);
	flacamd_stream_encoder_set_device(enc, opt->device);
	flacamd_stream_encoder_set_batch_frames(enc, opt->batch_frames ? opt->batch_frames : 4096u);
	{
		const FLAC__StreamEncoderInitStatus is = FLAC__stream_encoder_init_file(enc, out_path, 0, 0);
		if(is != FLAC__STREAM_ENCODER_INIT_STATUS_OK) {
			rc = fail(res, is == FLAC__STREAM_ENCODER_INIT_STATUS_ENCODER_ERROR ? FLACAMD_E_DEVICE : FLACAMD_E_INVALID_ARGUMENT, "initializing encoder: %s (%s)",
				FLAC__StreamEncoderInitStatusString[is], FLAC__stream_encoder_get_resolved_state_string(enc));
			goto done;
		}
	}

	/* samples: whole batches at a time */
	{
		const uint64_t have = in.available_bytes / bytes_per_wide;
		const size_t chunk = (size_t)(opt->batch_frames ? opt->batch_frames : 4096u) * FLAC__stream_encoder_get_blocksize(enc);
		buf = (int32_t*)malloc(sizeof(int32_t) * chunk * in.channels);
		if(!buf) { rc = fail(res, FLACAMD_E_OUT_OF_MEMORY, "allocating the sample buffer"); goto done; }
		while(fed < have) {
			const size_t n = have - fed < chunk ? (size_t)(have - fed) : chunk;
			if((rc = format_samples(&in, in.data + fed * bytes_per_wide, n, buf, res)) != 0) goto done;
			if(!FLAC__stream_encoder_process_interleaved(enc, buf, (unsigned)n)) {
				rc = fail(res, FLACAMD_E_DEVICE, "during encoding: %s", FLAC__stream_encoder_get_resolved_state_string(enc));
				goto done;
			}
			fed += n;
		}
	}
	if(!FLAC__stream_encoder_finish(enc)) { rc = fail(res, FLACAMD_E_DEVICE, "finishing the stream: %s", FLAC__stream_encoder_get_resolved_state_string(enc)); goto done; }
	if(res) {
		struct stat os;
		res->samples = fed; res->channels = in.channels; res->bits_per_sample = in.bps - in.shift; res->sample_rate = in.sample_rate; res->md5_ok = 1;
		if(stat(out_path, &os) == 0) res->bytes_written = (uint64_t)os.st_size;
	}
done:
	if(enc) { if(rc) (void)FLAC__stream_encoder_finish(enc); FLAC__stream_encoder_delete(enc); }
	free(buf); free(entries); free(points.pts);
	if(map && size) munmap((void*)map, (size_t)size);
	if(fd >= 0) close(fd);
	return rc;
}

/* ---- decode ---- */
typedef struct {
	FILE *out; int format; flacamd_file_result *res;
	unsigned channels, bps, sample_rate; FLAC__uint64 total_samples, written_samples; uint32_t channel_mask;
	int got_streaminfo, header_written, failed;
	FLAC__uint64 skip, until;   /* the range asked for, until 0 = to the end */
	int stopped_at_until;
	uint8_t *bytes; size_t bytes_cap;
} decode_session;

static void w_le(uint8_t *p, uint64_t v, unsigned n) { unsigned i; for(i = 0; i < n; i++) p[i] = (uint8_t)(v >> (8 * i)); }
static void w_be(uint8_t *p, uint64_t v, unsigned n) { unsigned i; for(i = 0; i < n; i++) p[i] = (uint8_t)(v >> (8 * (n - 1 - i))); }

static int wave_like(int format) { return format == FLACAMD_FILE_WAVE || format == FLACAMD_FILE_RF64 || format == FLACAMD_FILE_WAVE64; }

/* write_iff_headers (D:543-790) for WAVE, RF64, Wave64 and AIFF */
static int write_header(decode_session *s, FILE *f, FLAC__uint64 samples)
{
	uint8_t h[192]; size_t n = 0;
	const unsigned bytes_per_sample = (s->bps + 7) / 8;
	const int w64 = s->format == FLACAMD_FILE_WAVE64, rf64 = s->format == FLACAMD_FILE_RF64;
	const FLAC__uint64 data_size = samples * s->channels * bytes_per_sample, aligned = w64 ? (data_size + 7) & ~(FLAC__uint64)7 : (data_size + 1) & ~(FLAC__uint64)1;
	if(wave_like(s->format)) {
		const int ext = s->channel_mask == 2 || s->channel_mask > 3 || (s->bps % 8) || s->channels > 2;
		const unsigned fmt_body = ext ? 40 : 16;
		const FLAC__uint64 iff = samples == 0 ? 0 : w64 ? 16 + 8 + 16 + 16 + 8 + fmt_body + 16 + 8 + aligned : 4 + (rf64 ? 36 : 0) + 8 + fmt_body + 8 + aligned;
		if(s->format == FLACAMD_FILE_WAVE && iff >= 0xFFFFFFF4u) return -1;
		if(w64) {
			memcpy(h, W64_RIFF, 16); w_le(h + 16, iff, 8); memcpy(h + 24, W64_WAVE, 16); n = 40;
			memcpy(h + n, W64_FMT, 16); w_le(h + n + 16, 16 + 8 + fmt_body, 8); n += 24;
		}
		else {
			memcpy(h, rf64 ? "RF64" : "RIFF", 4); w_le(h + 4, rf64 ? 0xffffffffu : iff, 4); memcpy(h + 8, "WAVE", 4); n = 12;
			if(rf64) { memcpy(h + n, "ds64", 4); w_le(h + n + 4, 28, 4); w_le(h + n + 8, iff, 8); w_le(h + n + 16, data_size, 8); w_le(h + n + 24, samples, 8); w_le(h + n + 32, 0, 4); n += 36; }
			memcpy(h + n, "fmt ", 4); w_le(h + n + 4, fmt_body, 4); n += 8;
		}
		w_le(h + n, ext ? 65534 : 1, 2); w_le(h + n + 2, s->channels, 2); w_le(h + n + 4, s->sample_rate, 4);
		w_le(h + n + 8, (FLAC__uint64)s->sample_rate * s->channels * bytes_per_sample, 4); w_le(h + n + 12, s->channels * bytes_per_sample, 2);
		w_le(h + n + 14, bytes_per_sample * 8, 2); n += 16;
		if(ext) {
			w_le(h + n, 22, 2); w_le(h + n + 2, s->bps, 2); w_le(h + n + 4, s->channel_mask, 4);
			memcpy(h + n + 8, "\x01\x00\x00\x00\x00\x00\x10\x00\x80\x00\x00\xaa\x00\x38\x9b\x71", 16); n += 24;
		}
		if(w64) { memcpy(h + n, W64_DATA, 16); w_le(h + n + 16, 16 + 8 + data_size, 8); n += 24; }
		else { memcpy(h + n, "data", 4); w_le(h + n + 4, rf64 ? 0xffffffffu : data_size, 4); n += 8; }
	}
	else { /* AIFF */
		const FLAC__uint64 iff = samples == 0 ? 0 : 46 + aligned;
		unsigned shift = 0, rate = s->sample_rate;
		if(iff >= 0xFFFFFFF4u) return -1;
		memcpy(h, "FORM", 4); w_be(h + 4, iff, 4); memcpy(h + 8, "AIFFCOMM", 8); w_be(h + 16, 18, 4);
		w_be(h + 20, s->channels, 2); w_be(h + 22, samples, 4); w_be(h + 26, s->bps, 2);
		while((rate >> (31 - shift)) == 0) shift++;     /* 80-bit extended (D:910-939) */
		w_be(h + 28, 63u - (shift + 32u) + 0x3FFFu, 2); w_be(h + 30, (uint32_t)(rate << shift), 4); w_be(h + 34, 0, 4);
		memcpy(h + 38, "SSND", 4); w_be(h + 42, (uint32_t)data_size + 8, 4); w_be(h + 46, 0, 4); w_be(h + 50, 0, 4); n = 54;
	}
	return fwrite(h, 1, n, f) == n ? 0 : -1;
}

static void dec_metadata(const FLAC__StreamDecoder *d, const FLAC__StreamMetadata *m, void *cd)
{
	decode_session *s = (decode_session*)cd; (void)d;
	if(m->type == FLAC__METADATA_TYPE_STREAMINFO) {
		s->got_streaminfo = 1; s->channels = m->data.stream_info.channels; s->bps = m->data.stream_info.bits_per_sample;
		s->sample_rate = m->data.stream_info.sample_rate; s->total_samples = m->data.stream_info.total_samples;
	}
	else if(m->type == FLAC__METADATA_TYPE_VORBIS_COMMENT) { /* flac__utils_get_channel_mask_tag (src/flac/utils.c:299-321) */
		const size_t name = strlen(CHANNEL_MASK_TAG); unsigned i;
		for(i = 0; i < m->data.vorbis_comment.num_comments; i++) {
			const FLAC__StreamMetadata_VorbisComment_Entry *e = &m->data.vorbis_comment.comments[i];
			if(e->length > name + 3 && 0 == strncasecmp((const char*)e->entry, CHANNEL_MASK_TAG, name) && e->entry[name] == '=' && 0 == strncmp((const char*)e->entry + name + 1, "0x", 2)) {
				unsigned v;
				if(sscanf((const char*)e->entry + name + 3, "%x", &v) == 1) s->channel_mask = v;
				break;
			}
		}
	}
}

static void dec_error(const FLAC__StreamDecoder *d, FLAC__StreamDecoderErrorStatus st, void *cd)
{
	decode_session *s = (decode_session*)cd; (void)d;
	s->failed = 1;
	fail(s->res, FLACAMD_E_BAD_STREAM, "decoder error: %s", FLAC__StreamDecoderErrorStatusString[st]);
}

/* write_callback (D:964-1262): planar int32 -> container bytes */
static FLAC__StreamDecoderWriteStatus dec_write(const FLAC__StreamDecoder *d, const FLAC__Frame *fr, const FLAC__int32 * const buffer[], void *cd)
{
	decode_session *s = (decode_session*)cd;
	const unsigned bps = fr->header.bits_per_sample, ch = fr->header.channels; unsigned n = fr->header.blocksize;
	const unsigned shift = (s->format != FLACAMD_FILE_RAW && (bps % 8)) ? 8 - (bps % 8) : 0, width = (bps + shift + 7) / 8;
	const int big = s->format == FLACAMD_FILE_AIFF, uns = wave_like(s->format) && bps <= 8;
	const int32_t bias = uns ? (int32_t)1 << (bps + shift - 1) : 0;
	unsigned i, c; uint8_t *p; size_t need;
	(void)d;
	if(s->until > 0) { /* D:1069-1081: nothing past the --until point, and decoding ends there */
		const FLAC__uint64 passed = s->skip + s->written_samples;
		if(passed + n > s->until) n = (unsigned)(s->until - passed);
		if(n == 0) { s->stopped_at_until = 1; return FLAC__STREAM_DECODER_WRITE_STATUS_ABORT; }
	}
	if(!s->header_written) {
		if(!s->got_streaminfo) { s->channels = ch; s->bps = bps; s->sample_rate = fr->header.sample_rate; }
		if(s->channel_mask == 0) { static const uint32_t dflt[7] = { 0, 0x0001, 0x0003, 0x0007, 0x0033, 0x0607, 0x060f }; if(s->channels <= 6) s->channel_mask = dflt[s->channels]; } /* D:355-382 */
		if(s->format != FLACAMD_FILE_RAW && write_header(s, s->out, s->total_samples) != 0) { s->failed = 1; fail(s->res, FLACAMD_E_UNSUPPORTED, "stream is too big for the container, or the header could not be written"); return FLAC__STREAM_DECODER_WRITE_STATUS_ABORT; }
		s->header_written = 1;
	}
	if(bps != s->bps || ch != s->channels) { s->failed = 1; fail(s->res, FLACAMD_E_BAD_STREAM, "frame parameters differ from STREAMINFO"); return FLAC__STREAM_DECODER_WRITE_STATUS_ABORT; }
	need = (size_t)n * ch * width;
	if(need > s->bytes_cap) { uint8_t *nb = (uint8_t*)realloc(s->bytes, need); if(!nb) { s->failed = 1; return FLAC__STREAM_DECODER_WRITE_STATUS_ABORT; } s->bytes = nb; s->bytes_cap = need; }
	p = s->bytes;
	for(i = 0; i < n; i++) for(c = 0; c < ch; c++) {
		const uint32_t v = (uint32_t)(buffer[c][i] * (1 << shift) + bias);
		if(big) w_be(p, v, width); else w_le(p, v, width);
		p += width;
	}
	if(fwrite(s->bytes, 1, need, s->out) != need) { s->failed = 1; fail(s->res, FLACAMD_E_INVALID_ARGUMENT, "writing the output file"); return FLAC__STREAM_DECODER_WRITE_STATUS_ABORT; }
	s->written_samples += n;
	return FLAC__STREAM_DECODER_WRITE_STATUS_CONTINUE;
}

int flacamd_decode_file(const char *in_path, const char *out_path, int output_format, int device, flacamd_file_result *res)
{
	return flacamd_decode_file_range(in_path, out_path, output_format, device, 0, 0, res);
}

int flacamd_decode_file_range(const char *in_path, const char *out_path, int output_format, int device, const char *skip, const char *until, flacamd_file_result *res)
{
	decode_session s; FLAC__StreamDecoder *dec; int rc = 0; FLAC__bool ok, md5_ok;
	range_spec sk, un;
	if(res) memset(res, 0, sizeof *res);
	if(!in_path || !out_path) return fail(res, FLACAMD_E_INVALID_ARGUMENT, "missing path");
	if(!wave_like(output_format) && output_format != FLACAMD_FILE_AIFF && output_format != FLACAMD_FILE_RAW) return fail(res, FLACAMD_E_INVALID_ARGUMENT, "unknown output format %d", output_format);
	if((rc = parse_range_options(skip, until, &sk, &un, res)) != 0) return rc;
	memset(&s, 0, sizeof s);
	s.format = output_format; s.res = res;
	s.out = fopen(out_path, "wb");
	if(!s.out) return fail(res, FLACAMD_E_INVALID_ARGUMENT, "can't open output file %s", out_path);
	dec = FLAC__stream_decoder_new();
	if(!dec) { fclose(s.out); return fail(res, FLACAMD_E_OUT_OF_MEMORY, "creating the decoder instance"); }
	FLAC__stream_decoder_set_md5_checking(dec, 1);
	/* like the tool without --apply-replaygain-which-is-not-lossless, tag blocks are not requested (D:311-315), so a
	 * WAVEFORMATEXTENSIBLE_CHANNEL_MASK tag is not seen and the mask written is the default for the channel count */
	flacamd_stream_decoder_set_device(dec, device);
	if(FLAC__stream_decoder_init_file(dec, in_path, dec_write, dec_metadata, dec_error, &s) != FLAC__STREAM_DECODER_INIT_STATUS_OK) {
		rc = fail(res, FLACAMD_E_INVALID_ARGUMENT, "initializing decoder for %s: %s", in_path, FLAC__stream_decoder_get_resolved_state_string(dec));
		FLAC__stream_decoder_delete(dec); fclose(s.out); return rc;
	}
	ok = FLAC__stream_decoder_process_until_end_of_metadata(dec);
	if(ok && !s.failed && (skip || until)) { /* the range against STREAMINFO (D:1272-1309), then to the skip point (D:392-399) */
		const FLAC__uint64 total = s.got_streaminfo ? s.total_samples : 0;
		canonicalize_range_spec(&sk, s.sample_rate);
		s.skip = (FLAC__uint64)sk.samples;
		if(total > 0 && s.skip >= total) rc = fail(res, FLACAMD_E_INVALID_ARGUMENT, "trying to --skip more samples than in stream");
		else if(total == 0 && s.skip > 0) rc = fail(res, FLACAMD_E_INVALID_ARGUMENT, "can't --skip when FLAC metadata has total sample count of 0");
		else rc = canonicalize_until(&un, s.sample_rate, s.skip, total, res);
		if(!rc) {
			s.until = (FLAC__uint64)un.samples;
			s.total_samples = total - s.skip;
			if(s.until > 0) s.total_samples -= total - s.until;
			if(s.skip > 0 && !FLAC__stream_decoder_seek_absolute(dec, s.skip)) rc = fail(res, FLACAMD_E_BAD_STREAM, "seeking while skipping bytes: %s", FLAC__stream_decoder_get_resolved_state_string(dec));
		}
	}
	if(!rc) {
		if(ok) ok = FLAC__stream_decoder_process_until_end_of_stream(dec);
		if((!ok && !s.stopped_at_until) || s.failed) rc = fail(res, FLACAMD_E_BAD_STREAM, "while decoding: %s", FLAC__stream_decoder_get_resolved_state_string(dec));
	}
	md5_ok = FLAC__stream_decoder_finish(dec) || s.stopped_at_until;   /* D:448 */
	FLAC__stream_decoder_delete(dec);
	if(!rc && !s.header_written && s.got_streaminfo && s.format != FLACAMD_FILE_RAW) { if(write_header(&s, s.out, 0) != 0) rc = fail(res, FLACAMD_E_INVALID_ARGUMENT, "writing the header"); s.header_written = 1; } /* a stream without frames */
	if(!rc && s.format != FLACAMD_FILE_RAW) {
		const FLAC__uint64 data_size = s.written_samples * s.channels * ((s.bps + 7) / 8);
		unsigned pad = s.format == FLACAMD_FILE_WAVE64 ? (8u - (unsigned)(data_size & 7)) & 7u : (unsigned)(data_size & 1);   /* D:415-441 */
		for(; pad > 0; pad--) fputc(0, s.out);
		if(s.total_samples == 0 && s.written_samples > 0) { /* sizes were unknown up front: go back and fix the header (D:941-962) */
			fflush(s.out);
			if(fseeko(s.out, 0, SEEK_SET) != 0 || write_header(&s, s.out, s.written_samples) != 0) rc = fail(res, FLACAMD_E_INVALID_ARGUMENT, "fixing up the header");
		}
	}
	if(fclose(s.out) != 0 && !rc) rc = fail(res, FLACAMD_E_INVALID_ARGUMENT, "closing the output file");
	free(s.bytes);
	if(res) {
		struct stat os;
		res->samples = s.written_samples; res->channels = s.channels; res->bits_per_sample = s.bps; res->sample_rate = s.sample_rate; res->md5_ok = md5_ok ? 1 : 0;
		if(stat(out_path, &os) == 0) res->bytes_written = (uint64_t)os.st_size;
	}
	if(!rc && !md5_ok) rc = fail(res, FLACAMD_E_BAD_STREAM, "MD5 signature mismatch");
	return rc;
}
