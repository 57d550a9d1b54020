/* Metadata blocks on the host: the legality checks the encoder applies to client blocks at
 * init (/root/reference/src/libFLAC/format.c:248-523) and the byte image of a block as it is
 * written into a stream (/root/reference/src/libFLAC/stream_encoder_framing.c:50-217).
 * Plain C; nothing here touches the device.
 */
#include "metadata_blocks.h"
#include <stdlib.h>
#include <string.h>

const char * const FLAC__MetadataTypeString[] = { "STREAMINFO", "PADDING", "APPLICATION", "SEEKTABLE", "VORBIS_COMMENT", "CUESHEET", "PICTURE" }; /* format.c:118-126 */
const FLAC__uint64 FLAC__STREAM_METADATA_SEEKPOINT_PLACEHOLDER = FLAC__STREAM_METADATA_SEEKPOINT_PLACEHOLDER_VALUE;

/* ---- legality ---- */

/* format.c:248-269: sample numbers strictly ascending, placeholders allowed anywhere after the first point */
FLAC__bool FLAC__format_seektable_is_legal(const FLAC__StreamMetadata_SeekTable *t)
{
	unsigned i;
	for(i = 1; i < t->num_points; i++)
		if(t->points[i].sample_number != FLAC__STREAM_METADATA_SEEKPOINT_PLACEHOLDER && t->points[i].sample_number <= t->points[i - 1].sample_number) return 0;
	return 1;
}

static int by_sample_number(const void *a, const void *b)
{
	const FLAC__uint64 l = ((const FLAC__StreamMetadata_SeekPoint*)a)->sample_number, r = ((const FLAC__StreamMetadata_SeekPoint*)b)->sample_number;
	return l < r ? -1 : l > r;
}

/* format.c:284-314: sort, drop duplicate sample numbers (placeholders are all kept), turn the freed tail into placeholders */
unsigned FLAC__format_seektable_sort(FLAC__StreamMetadata_SeekTable *t)
{
	unsigned i, kept = 0;
	qsort(t->points, t->num_points, sizeof t->points[0], by_sample_number);
	for(i = 0; i < t->num_points; i++) {
		if(kept > 0 && t->points[i].sample_number != FLAC__STREAM_METADATA_SEEKPOINT_PLACEHOLDER && t->points[i].sample_number == t->points[kept - 1].sample_number) continue;
		t->points[kept++] = t->points[i];
	}
	for(i = kept; i < t->num_points; i++) { t->points[i].sample_number = FLAC__STREAM_METADATA_SEEKPOINT_PLACEHOLDER; t->points[i].stream_offset = 0; t->points[i].frame_samples = 0; }
	return kept;
}

/* length of the UTF-8 sequence starting at s, 0 if it is malformed, overlong, a surrogate or
 * U+FFFE/U+FFFF (format.c:322-361; like the reference it accepts the old 5- and 6-byte forms) */
static unsigned utf8_sequence(const FLAC__byte *s)
{
	unsigned n, i;
	if(s[0] < 0x80) return 1;
	if((s[0] & 0xE0) == 0xC0) n = 2; else if((s[0] & 0xF0) == 0xE0) n = 3; else if((s[0] & 0xF8) == 0xF0) n = 4;
	else if((s[0] & 0xFC) == 0xF8) n = 5; else if((s[0] & 0xFE) == 0xFC) n = 6; else return 0;
	for(i = 1; i < n; i++) if((s[i] & 0xC0) != 0x80) return 0; /* stops at a NUL, so never reads past a terminated string */
	if(n == 2) return (s[0] & 0xFE) == 0xC0 ? 0 : 2;
	if(n == 3) {
		if(s[0] == 0xE0 && (s[1] & 0xE0) == 0x80) return 0;
		if(s[0] == 0xED && (s[1] & 0xE0) == 0xA0) return 0;
		if(s[0] == 0xEF && s[1] == 0xBF && (s[2] & 0xFE) == 0xBE) return 0;
		return 3;
	}
	/* overlong: no payload bits in the lead byte and none in the top of the second byte */
	{ const FLAC__byte lead = (FLAC__byte)(0xFF << (8 - n)); if(s[0] == lead && (s[1] & lead) == 0x80) return 0; }
	return n;
}

/* format.c:396-419: NAME=value, name in 0x20..0x7D without '=', value well-formed UTF-8 */
FLAC__bool FLAC__format_vorbiscomment_entry_is_legal(const FLAC__byte *entry, unsigned length)
{
	const FLAC__byte *s = entry, *end = entry + length;
	for(; s < end && *s != '='; s++) if(*s < 0x20 || *s > 0x7D) return 0;
	if(s == end) return 0;
	for(s++; s < end; ) {
		unsigned n = utf8_sequence(s);
		if(n == 0) return 0;
		s += n;
	}
	return s == end;
}

/* format.c:363-370: every character of a tag's name printable ASCII other than '=' and '~' */
FLAC__bool FLAC__format_vorbiscomment_entry_name_is_legal(const char *name)
{
	for(; *name; name++) if(*name < 0x20 || *name == '=' || *name > 0x7D) return 0;
	return 1;
}

/* format.c:372-394: the value is whole well-formed UTF-8 sequences; length (unsigned)(-1) means NUL-terminated */
FLAC__bool FLAC__format_vorbiscomment_entry_value_is_legal(const FLAC__byte *value, unsigned length)
{
	const int terminated = length == (unsigned)(-1);
	const FLAC__byte *end = terminated ? 0 : value + length;
	while(terminated ? *value != 0 : value < end) {
		unsigned n = utf8_sequence(value);
		if(n == 0) return 0;
		value += n;
	}
	return terminated || value == end;
}

/* format.c:422-498 */
FLAC__bool FLAC__format_cuesheet_is_legal(const FLAC__StreamMetadata_CueSheet *cs, FLAC__bool cdda, const char **violation)
{
	unsigned i, j;
#define ILLEGAL(text) do { if(violation) *violation = (text); return 0; } while(0)
	if(cdda && cs->lead_in < 2 * 44100) ILLEGAL("CD-DA cue sheet must have a lead-in length of at least 2 seconds");
	if(cdda && cs->lead_in % 588 != 0) ILLEGAL("CD-DA cue sheet lead-in length must be evenly divisible by 588 samples");
	if(cs->num_tracks == 0) ILLEGAL("cue sheet must have at least one track (the lead-out)");
	if(cdda && cs->tracks[cs->num_tracks - 1].number != 170) ILLEGAL("CD-DA cue sheet must have a lead-out track number 170 (0xAA)");
	for(i = 0; i < cs->num_tracks; i++) {
		const FLAC__StreamMetadata_CueSheet_Track *t = &cs->tracks[i];
		const int lead_out = i == cs->num_tracks - 1;
		if(t->number == 0) ILLEGAL("cue sheet may not have a track number 0");
		if(cdda && !((t->number >= 1 && t->number <= 99) || t->number == 170)) ILLEGAL("CD-DA cue sheet track number must be 1-99 or 170");
		if(cdda && t->offset % 588 != 0) ILLEGAL(lead_out ? "CD-DA cue sheet lead-out offset must be evenly divisible by 588 samples" : "CD-DA cue sheet track offset must be evenly divisible by 588 samples");
		if(!lead_out) {
			if(t->num_indices == 0) ILLEGAL("cue sheet track must have at least one index point");
			if(t->indices[0].number > 1) ILLEGAL("cue sheet track's first index number must be 0 or 1");
		}
		for(j = 0; j < t->num_indices; j++) {
			if(cdda && t->indices[j].offset % 588 != 0) ILLEGAL("CD-DA cue sheet track index offset must be evenly divisible by 588 samples");
			if(j > 0 && t->indices[j].number != t->indices[j - 1].number + 1) ILLEGAL("cue sheet track index numbers must increase by 1");
		}
	}
	return 1;
}

/* format.c:501-523 */
FLAC__bool FLAC__format_picture_is_legal(const FLAC__StreamMetadata_Picture *pic, const char **violation)
{
	const char *p; const FLAC__byte *b;
	for(p = pic->mime_type; *p; p++) if(*p < 0x20 || *p > 0x7e) ILLEGAL("MIME type string must contain only printable ASCII characters (0x20-0x7e)");
	for(b = pic->description; *b; ) {
		unsigned n = utf8_sequence(b);
		if(n == 0) ILLEGAL("description string must be valid UTF-8");
		b += n;
	}
	return 1;
#undef ILLEGAL
}

/* ---- byte image ---- */

typedef struct { uint8_t *p; size_t n, cap; int failed; } sink;
static void need(sink *s, size_t more)
{
	if(s->failed || s->n + more <= s->cap) return;
	{
		size_t cap = s->cap ? s->cap : 256;
		uint8_t *q;
		while(cap < s->n + more) cap *= 2;
		q = (uint8_t*)realloc(s->p, cap);
		if(!q) { s->failed = 1; return; }
		s->p = q; s->cap = cap;
	}
}
static void put_bytes(sink *s, const void *src, size_t n) { need(s, n); if(!s->failed && n) { memcpy(s->p + s->n, src, n); s->n += n; } }
static void put_zero(sink *s, size_t n) { need(s, n); if(!s->failed) { memset(s->p + s->n, 0, n); s->n += n; } }
static void put_be(sink *s, uint64_t v, unsigned nbytes) { uint8_t b[8]; unsigned i; for(i = 0; i < nbytes; i++) b[i] = (uint8_t)(v >> (8 * (nbytes - 1 - i))); put_bytes(s, b, nbytes); }
static void put_le32(sink *s, uint32_t v) { uint8_t b[4]; b[0] = (uint8_t)v; b[1] = (uint8_t)(v >> 8); b[2] = (uint8_t)(v >> 16); b[3] = (uint8_t)(v >> 24); put_bytes(s, b, 4); }

int flacamd_metadata_block_image(const FLAC__StreamMetadata *m, const char *vendor, uint8_t **out, size_t *out_len)
{
	sink s = { 0, 0, 0, 0 };
	const size_t vlen = strlen(vendor);
	unsigned length = m->length, i, j;
	/* the vendor string written is always the library's own (stream_encoder_framing.c:61-72) */
	if(m->type == FLAC__METADATA_TYPE_VORBIS_COMMENT) length = length - m->data.vorbis_comment.vendor_string.length + (unsigned)vlen;
	if(length >= (1u << 24)) return -1;
	put_be(&s, ((uint32_t)(m->is_last ? 0x80u : 0u) | ((unsigned)m->type & 0x7f)), 1);
	put_be(&s, length, 3);
	switch(m->type) {
	case FLAC__METADATA_TYPE_STREAMINFO: {
		const FLAC__StreamMetadata_StreamInfo *si = &m->data.stream_info;
		put_be(&s, si->min_blocksize, 2); put_be(&s, si->max_blocksize, 2); put_be(&s, si->min_framesize, 3); put_be(&s, si->max_framesize, 3);
		put_be(&s, ((uint64_t)si->sample_rate << 44) | ((uint64_t)(si->channels - 1) << 41) | ((uint64_t)(si->bits_per_sample - 1) << 36) | (si->total_samples & 0xFFFFFFFFFull), 8);
		put_bytes(&s, si->md5sum, 16);
		break; }
	case FLAC__METADATA_TYPE_PADDING:
		put_zero(&s, m->length);
		break;
	case FLAC__METADATA_TYPE_APPLICATION:
		put_bytes(&s, m->data.application.id, 4);
		put_bytes(&s, m->data.application.data, m->length - 4);
		break;
	case FLAC__METADATA_TYPE_SEEKTABLE:
		for(i = 0; i < m->data.seek_table.num_points; i++) {
			const FLAC__StreamMetadata_SeekPoint *pt = &m->data.seek_table.points[i];
			put_be(&s, pt->sample_number, 8); put_be(&s, pt->stream_offset, 8); put_be(&s, pt->frame_samples, 2);
		}
		break;
	case FLAC__METADATA_TYPE_VORBIS_COMMENT:
		put_le32(&s, (uint32_t)vlen); put_bytes(&s, vendor, vlen);
		put_le32(&s, m->data.vorbis_comment.num_comments);
		for(i = 0; i < m->data.vorbis_comment.num_comments; i++) {
			put_le32(&s, m->data.vorbis_comment.comments[i].length);
			put_bytes(&s, m->data.vorbis_comment.comments[i].entry, m->data.vorbis_comment.comments[i].length);
		}
		break;
	case FLAC__METADATA_TYPE_CUESHEET: {
		const FLAC__StreamMetadata_CueSheet *cs = &m->data.cue_sheet;
		put_bytes(&s, cs->media_catalog_number, 128); put_be(&s, cs->lead_in, 8);
		put_be(&s, cs->is_cd ? 0x80 : 0, 1); put_zero(&s, 258);             /* 1 flag bit + 7 + 258*8 reserved bits */
		put_be(&s, cs->num_tracks, 1);
		for(i = 0; i < cs->num_tracks; i++) {
			const FLAC__StreamMetadata_CueSheet_Track *t = &cs->tracks[i];
			put_be(&s, t->offset, 8); put_be(&s, t->number, 1); put_bytes(&s, t->isrc, 12);
			put_be(&s, (t->type ? 0x80u : 0u) | (t->pre_emphasis ? 0x40u : 0u), 1); put_zero(&s, 13);  /* 2 flag bits + 6 + 13*8 reserved bits */
			put_be(&s, t->num_indices, 1);
			for(j = 0; j < t->num_indices; j++) { put_be(&s, t->indices[j].offset, 8); put_be(&s, t->indices[j].number, 1); put_zero(&s, 3); }
		}
		break; }
	case FLAC__METADATA_TYPE_PICTURE: {
		const FLAC__StreamMetadata_Picture *pic = &m->data.picture;
		const size_t ml = strlen(pic->mime_type), dl = strlen((const char*)pic->description);
		put_be(&s, (uint32_t)pic->type, 4);
		put_be(&s, ml, 4); put_bytes(&s, pic->mime_type, ml);
		put_be(&s, dl, 4); put_bytes(&s, pic->description, dl);
		put_be(&s, pic->width, 4); put_be(&s, pic->height, 4); put_be(&s, pic->depth, 4); put_be(&s, pic->colors, 4);
		put_be(&s, pic->data_length, 4); put_bytes(&s, pic->data, pic->data_length);
		break; }
	default:
		put_bytes(&s, m->data.unknown.data, m->length);
		break;
	}
	if(s.failed) { free(s.p); return -1; }
	*out = s.p; *out_len = s.n;
	return 0;
}

/* ---- parsing (what read_metadata_ builds for the client, /root/reference/src/libFLAC/stream_decoder.c:1427-1602,
 * :1675-1975): every pointer inside the object is malloc'ed here and released by
 * flacamd_metadata_block_free.  Truncated or inconsistent block data fails instead of reading past `n`. ---- */

typedef struct { const uint8_t *p; size_t n, pos; int bad; } source;
static uint64_t get_be(source *s, unsigned nbytes)
{
	uint64_t v = 0; unsigned i;
	if(s->bad || s->n - s->pos < nbytes) { s->bad = 1; return 0; }
	for(i = 0; i < nbytes; i++) v = (v << 8) | s->p[s->pos++];
	return v;
}
static uint32_t get_le32(source *s)
{
	uint32_t v;
	if(s->bad || s->n - s->pos < 4) { s->bad = 1; return 0; }
	v = (uint32_t)s->p[s->pos] | ((uint32_t)s->p[s->pos + 1] << 8) | ((uint32_t)s->p[s->pos + 2] << 16) | ((uint32_t)s->p[s->pos + 3] << 24);
	s->pos += 4;
	return v;
}
static void get_bytes(source *s, void *dst, size_t n)
{
	if(s->bad || s->n - s->pos < n) { s->bad = 1; return; }
	memcpy(dst, s->p + s->pos, n); s->pos += n;
}
/* n bytes as a fresh NUL-terminated buffer; NULL (and no error) when n == 0 and !always */
static uint8_t *get_string(source *s, size_t n, int always)
{
	uint8_t *b;
	if(s->bad || s->n - s->pos < n) { s->bad = 1; return 0; }
	if(n == 0 && !always) return 0;
	b = (uint8_t*)malloc(n + 1);
	if(!b) { s->bad = 1; return 0; }
	memcpy(b, s->p + s->pos, n); b[n] = 0; s->pos += n;
	return b;
}

void flacamd_metadata_block_free(FLAC__StreamMetadata *m)
{
	unsigned i;
	switch(m->type) {
	case FLAC__METADATA_TYPE_STREAMINFO: case FLAC__METADATA_TYPE_PADDING: break;
	case FLAC__METADATA_TYPE_APPLICATION: free(m->data.application.data); break;
	case FLAC__METADATA_TYPE_SEEKTABLE: free(m->data.seek_table.points); break;
	case FLAC__METADATA_TYPE_VORBIS_COMMENT:
		free(m->data.vorbis_comment.vendor_string.entry);
		if(m->data.vorbis_comment.comments) for(i = 0; i < m->data.vorbis_comment.num_comments; i++) free(m->data.vorbis_comment.comments[i].entry);
		free(m->data.vorbis_comment.comments);
		break;
	case FLAC__METADATA_TYPE_CUESHEET:
		if(m->data.cue_sheet.tracks) for(i = 0; i < m->data.cue_sheet.num_tracks; i++) free(m->data.cue_sheet.tracks[i].indices);
		free(m->data.cue_sheet.tracks);
		break;
	case FLAC__METADATA_TYPE_PICTURE: free(m->data.picture.mime_type); free(m->data.picture.description); free(m->data.picture.data); break;
	default: free(m->data.unknown.data); break;
	}
	memset(&m->data, 0, sizeof m->data);
}

int flacamd_metadata_block_parse(unsigned type, int is_last, const uint8_t *data, size_t n, FLAC__StreamMetadata *m)
{
	source s = { data, n, 0, 0 };
	unsigned i, j;
	memset(m, 0, sizeof *m);
	m->type = (FLAC__MetadataType)type; m->is_last = is_last ? 1 : 0; m->length = (unsigned)n;
	switch(type) {
	case FLAC__METADATA_TYPE_PADDING: break;
	case FLAC__METADATA_TYPE_APPLICATION:
		get_bytes(&s, m->data.application.id, 4);
		if(!s.bad && n > 4) { m->data.application.data = (FLAC__byte*)malloc(n - 4); if(!m->data.application.data) s.bad = 1; else get_bytes(&s, m->data.application.data, n - 4); }
		break;
	case FLAC__METADATA_TYPE_SEEKTABLE: { /* a trailing partial point is ignored (:1706-1712) */
		FLAC__StreamMetadata_SeekTable *t = &m->data.seek_table;
		t->num_points = (unsigned)(n / FLAC__STREAM_METADATA_SEEKPOINT_LENGTH);
		t->points = (FLAC__StreamMetadata_SeekPoint*)calloc(t->num_points ? t->num_points : 1, sizeof t->points[0]);
		if(!t->points) { s.bad = 1; break; }
		for(i = 0; i < t->num_points; i++) { t->points[i].sample_number = get_be(&s, 8); t->points[i].stream_offset = get_be(&s, 8); t->points[i].frame_samples = (unsigned)get_be(&s, 2); }
		break; }
	case FLAC__METADATA_TYPE_VORBIS_COMMENT: {
		FLAC__StreamMetadata_VorbisComment *vc = &m->data.vorbis_comment;
		vc->vendor_string.length = get_le32(&s);
		vc->vendor_string.entry = get_string(&s, vc->vendor_string.length, 0);
		vc->num_comments = get_le32(&s);
		if(s.bad) break;
		if((size_t)vc->num_comments > (n - s.pos) / 4) { s.bad = 1; vc->num_comments = 0; break; } /* every comment needs its length word */
		if(vc->num_comments) {
			vc->comments = (FLAC__StreamMetadata_VorbisComment_Entry*)calloc(vc->num_comments, sizeof vc->comments[0]);
			if(!vc->comments) { s.bad = 1; vc->num_comments = 0; break; }
			for(i = 0; i < vc->num_comments && !s.bad; i++) { vc->comments[i].length = get_le32(&s); vc->comments[i].entry = get_string(&s, vc->comments[i].length, 0); }
		}
		break; }
	case FLAC__METADATA_TYPE_CUESHEET: {
		FLAC__StreamMetadata_CueSheet *cs = &m->data.cue_sheet;
		get_bytes(&s, cs->media_catalog_number, 128);
		cs->lead_in = get_be(&s, 8);
		cs->is_cd = (get_be(&s, 1) & 0x80) ? 1 : 0;
		s.pos += 258; if(s.pos > n) s.bad = 1;
		cs->num_tracks = (unsigned)get_be(&s, 1);
		if(s.bad || cs->num_tracks == 0) break;
		cs->tracks = (FLAC__StreamMetadata_CueSheet_Track*)calloc(cs->num_tracks, sizeof cs->tracks[0]);
		if(!cs->tracks) { s.bad = 1; cs->num_tracks = 0; break; }
		for(i = 0; i < cs->num_tracks && !s.bad; i++) {
			FLAC__StreamMetadata_CueSheet_Track *t = &cs->tracks[i];
			unsigned flags;
			t->offset = get_be(&s, 8); t->number = (FLAC__byte)get_be(&s, 1); get_bytes(&s, t->isrc, 12);
			flags = (unsigned)get_be(&s, 1); t->type = (flags >> 7) & 1u; t->pre_emphasis = (flags >> 6) & 1u;
			s.pos += 13; if(s.pos > n) s.bad = 1;
			t->num_indices = (FLAC__byte)get_be(&s, 1);
			if(s.bad || t->num_indices == 0) continue;
			t->indices = (FLAC__StreamMetadata_CueSheet_Index*)calloc(t->num_indices, sizeof t->indices[0]);
			if(!t->indices) { s.bad = 1; t->num_indices = 0; break; }
			for(j = 0; j < t->num_indices; j++) { t->indices[j].offset = get_be(&s, 8); t->indices[j].number = (FLAC__byte)get_be(&s, 1); s.pos += 3; if(s.pos > n) s.bad = 1; }
		}
		break; }
	case FLAC__METADATA_TYPE_PICTURE: {
		FLAC__StreamMetadata_Picture *pic = &m->data.picture;
		uint32_t len;
		pic->type = (FLAC__StreamMetadata_Picture_Type)get_be(&s, 4);
		len = (uint32_t)get_be(&s, 4); pic->mime_type = (char*)get_string(&s, len, 1);
		len = (uint32_t)get_be(&s, 4); pic->description = get_string(&s, len, 1);
		pic->width = (uint32_t)get_be(&s, 4); pic->height = (uint32_t)get_be(&s, 4); pic->depth = (uint32_t)get_be(&s, 4); pic->colors = (uint32_t)get_be(&s, 4);
		pic->data_length = (uint32_t)get_be(&s, 4);
		pic->data = get_string(&s, pic->data_length, 1);
		break; }
	default:
		if(n) { m->data.unknown.data = (FLAC__byte*)malloc(n); if(!m->data.unknown.data) s.bad = 1; else get_bytes(&s, m->data.unknown.data, n); }
		break;
	}
	if(s.bad) { flacamd_metadata_block_free(m); return -1; }
	return 0;
}
