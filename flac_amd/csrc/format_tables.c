/* The data symbols and small helpers of the reference's public format interface that carry no
 * audio work: version and vendor strings, the stream and frame sync codes, the bit width of
 * every field of the bitstream (include/FLAC/format.h:118-880 declares them `extern const`, so a
 * client compiled against the reference's headers reads them from the library at run time),
 * the status name tables of the callback enums, and the range checks of
 * /root/reference/src/libFLAC/format.c:213-245 and :528-561.  Values are the format's; the
 * grouping below follows the bitstream from the outside in.
 */
#include "../../include/flac_amd_stream.h"

/* format.c:56-67 */
const char *FLAC__VERSION_STRING = "1.2.1";
const char *FLAC__VENDOR_STRING = "reference libFLAC 1.2.1 20070917"; /* decoders and tools compare it */
const FLAC__byte FLAC__STREAM_SYNC_STRING[4] = { 0x66, 0x4C, 0x61, 0x43 }; /* "fLaC" */
const unsigned FLAC__STREAM_SYNC = 0x664C6143u;
/* include/FLAC/export.h:83: this library has no Ogg transport */
int FLAC_API_SUPPORTS_OGG_FLAC = 0;

#define WIDTH(name, bits) const unsigned FLAC__##name##_LEN = (bits);

/* stream head and the 4-byte header of every metadata block (format.c:67, :117-119) */
WIDTH(STREAM_SYNC, 32)
WIDTH(STREAM_METADATA_IS_LAST, 1) WIDTH(STREAM_METADATA_TYPE, 7) WIDTH(STREAM_METADATA_LENGTH, 24)

/* STREAMINFO, 34 bytes (format.c:69-77) */
WIDTH(STREAM_METADATA_STREAMINFO_MIN_BLOCK_SIZE, 16) WIDTH(STREAM_METADATA_STREAMINFO_MAX_BLOCK_SIZE, 16)
WIDTH(STREAM_METADATA_STREAMINFO_MIN_FRAME_SIZE, 24) WIDTH(STREAM_METADATA_STREAMINFO_MAX_FRAME_SIZE, 24)
WIDTH(STREAM_METADATA_STREAMINFO_SAMPLE_RATE, 20) WIDTH(STREAM_METADATA_STREAMINFO_CHANNELS, 3)
WIDTH(STREAM_METADATA_STREAMINFO_BITS_PER_SAMPLE, 5) WIDTH(STREAM_METADATA_STREAMINFO_TOTAL_SAMPLES, 36)
WIDTH(STREAM_METADATA_STREAMINFO_MD5SUM, 128)

/* APPLICATION, SEEKTABLE points (18 bytes each), VORBIS_COMMENT (format.c:79-88) */
WIDTH(STREAM_METADATA_APPLICATION_ID, 32)
WIDTH(STREAM_METADATA_SEEKPOINT_SAMPLE_NUMBER, 64) WIDTH(STREAM_METADATA_SEEKPOINT_STREAM_OFFSET, 64)
WIDTH(STREAM_METADATA_SEEKPOINT_FRAME_SAMPLES, 16)
WIDTH(STREAM_METADATA_VORBIS_COMMENT_ENTRY_LENGTH, 32) WIDTH(STREAM_METADATA_VORBIS_COMMENT_NUM_COMMENTS, 32)

/* CUESHEET: block, then tracks, then index points (format.c:90-106) */
WIDTH(STREAM_METADATA_CUESHEET_MEDIA_CATALOG_NUMBER, 128 * 8) WIDTH(STREAM_METADATA_CUESHEET_LEAD_IN, 64)
WIDTH(STREAM_METADATA_CUESHEET_IS_CD, 1) WIDTH(STREAM_METADATA_CUESHEET_RESERVED, 7 + 258 * 8)
WIDTH(STREAM_METADATA_CUESHEET_NUM_TRACKS, 8)
WIDTH(STREAM_METADATA_CUESHEET_TRACK_OFFSET, 64) WIDTH(STREAM_METADATA_CUESHEET_TRACK_NUMBER, 8)
WIDTH(STREAM_METADATA_CUESHEET_TRACK_ISRC, 12 * 8) WIDTH(STREAM_METADATA_CUESHEET_TRACK_TYPE, 1)
WIDTH(STREAM_METADATA_CUESHEET_TRACK_PRE_EMPHASIS, 1) WIDTH(STREAM_METADATA_CUESHEET_TRACK_RESERVED, 6 + 13 * 8)
WIDTH(STREAM_METADATA_CUESHEET_TRACK_NUM_INDICES, 8)
WIDTH(STREAM_METADATA_CUESHEET_INDEX_OFFSET, 64) WIDTH(STREAM_METADATA_CUESHEET_INDEX_NUMBER, 8)
WIDTH(STREAM_METADATA_CUESHEET_INDEX_RESERVED, 3 * 8)

/* PICTURE: eight 32-bit fields around the three variable-length ones (format.c:108-115) */
WIDTH(STREAM_METADATA_PICTURE_TYPE, 32) WIDTH(STREAM_METADATA_PICTURE_MIME_TYPE_LENGTH, 32)
WIDTH(STREAM_METADATA_PICTURE_DESCRIPTION_LENGTH, 32) WIDTH(STREAM_METADATA_PICTURE_WIDTH, 32)
WIDTH(STREAM_METADATA_PICTURE_HEIGHT, 32) WIDTH(STREAM_METADATA_PICTURE_DEPTH, 32)
WIDTH(STREAM_METADATA_PICTURE_COLORS, 32) WIDTH(STREAM_METADATA_PICTURE_DATA_LENGTH, 32)

/* frame header and footer (format.c:121-132) */
const unsigned FLAC__FRAME_HEADER_SYNC = 0x3ffeu;
WIDTH(FRAME_HEADER_SYNC, 14) WIDTH(FRAME_HEADER_RESERVED, 1) WIDTH(FRAME_HEADER_BLOCKING_STRATEGY, 1)
WIDTH(FRAME_HEADER_BLOCK_SIZE, 4) WIDTH(FRAME_HEADER_SAMPLE_RATE, 4) WIDTH(FRAME_HEADER_CHANNEL_ASSIGNMENT, 4)
WIDTH(FRAME_HEADER_BITS_PER_SAMPLE, 3) WIDTH(FRAME_HEADER_ZERO_PAD, 1) WIDTH(FRAME_HEADER_CRC, 8)
WIDTH(FRAME_FOOTER_CRC, 16)

/* subframe header, the LPC subframe's two small fields (format.c:148-158) */
WIDTH(SUBFRAME_ZERO_PAD, 1) WIDTH(SUBFRAME_TYPE, 6) WIDTH(SUBFRAME_WASTED_BITS_FLAG, 1)
WIDTH(SUBFRAME_LPC_QLP_COEFF_PRECISION, 4) WIDTH(SUBFRAME_LPC_QLP_SHIFT, 5)
const unsigned FLAC__SUBFRAME_TYPE_CONSTANT_BYTE_ALIGNED_MASK = 0x00u;
const unsigned FLAC__SUBFRAME_TYPE_VERBATIM_BYTE_ALIGNED_MASK = 0x02u;
const unsigned FLAC__SUBFRAME_TYPE_FIXED_BYTE_ALIGNED_MASK = 0x10u;
const unsigned FLAC__SUBFRAME_TYPE_LPC_BYTE_ALIGNED_MASK = 0x40u;

/* residual coding (format.c:134-141): the escape parameter is the all-ones value of its field */
WIDTH(ENTROPY_CODING_METHOD_TYPE, 2) WIDTH(ENTROPY_CODING_METHOD_PARTITIONED_RICE_ORDER, 4)
WIDTH(ENTROPY_CODING_METHOD_PARTITIONED_RICE_PARAMETER, 4) WIDTH(ENTROPY_CODING_METHOD_PARTITIONED_RICE2_PARAMETER, 5)
WIDTH(ENTROPY_CODING_METHOD_PARTITIONED_RICE_RAW, 5)
const unsigned FLAC__ENTROPY_CODING_METHOD_PARTITIONED_RICE_ESCAPE_PARAMETER = (1u << 4) - 1u;
const unsigned FLAC__ENTROPY_CODING_METHOD_PARTITIONED_RICE2_ESCAPE_PARAMETER = (1u << 5) - 1u;
#undef WIDTH

/* ---- name tables (the element order is the enum order of the headers) ---- */
const char * const FLAC__FrameNumberTypeString[] = { "FRAME_NUMBER_TYPE_FRAME_NUMBER", "FRAME_NUMBER_TYPE_SAMPLE_NUMBER" }; /* format.c:174-177 */

/* format.c:189-211: the ID3v2 APIC picture kinds, 0..20 */
const char * const FLAC__StreamMetadata_Picture_TypeString[] = {
	"Other", "32x32 pixels 'file icon' (PNG only)", "Other file icon", "Cover (front)", "Cover (back)", "Leaflet page",
	"Media (e.g. label side of CD)", "Lead artist/lead performer/soloist", "Artist/performer", "Conductor", "Band/Orchestra",
	"Composer", "Lyricist/text writer", "Recording Location", "During recording", "During performance",
	"Movie/video screen capture", "A bright coloured fish", "Illustration", "Band/artist logotype", "Publisher/Studio logotype" };

#define ENC(x) "FLAC__STREAM_ENCODER_" x
#define DEC(x) "FLAC__STREAM_DECODER_" x
/* stream_encoder.c:487-509.  The reference defines the read table under a misspelt name (":487
 * FLAC__treamEncoderReadStatusString") while its header declares the right one; both are here, so
 * that a client of either spelling links. */
const char * const FLAC__StreamEncoderReadStatusString[] = { ENC("READ_STATUS_CONTINUE"), ENC("READ_STATUS_END_OF_STREAM"), ENC("READ_STATUS_ABORT"), ENC("READ_STATUS_UNSUPPORTED") };
const char * const FLAC__treamEncoderReadStatusString[] = { ENC("READ_STATUS_CONTINUE"), ENC("READ_STATUS_END_OF_STREAM"), ENC("READ_STATUS_ABORT"), ENC("READ_STATUS_UNSUPPORTED") };
const char * const FLAC__StreamEncoderWriteStatusString[] = { ENC("WRITE_STATUS_OK"), ENC("WRITE_STATUS_FATAL_ERROR") };
const char * const FLAC__StreamEncoderSeekStatusString[] = { ENC("SEEK_STATUS_OK"), ENC("SEEK_STATUS_ERROR"), ENC("SEEK_STATUS_UNSUPPORTED") };
const char * const FLAC__StreamEncoderTellStatusString[] = { ENC("TELL_STATUS_OK"), ENC("TELL_STATUS_ERROR"), ENC("TELL_STATUS_UNSUPPORTED") };
/* stream_decoder.c:243-270 */
const char * const FLAC__StreamDecoderReadStatusString[] = { DEC("READ_STATUS_CONTINUE"), DEC("READ_STATUS_END_OF_STREAM"), DEC("READ_STATUS_ABORT") };
const char * const FLAC__StreamDecoderSeekStatusString[] = { DEC("SEEK_STATUS_OK"), DEC("SEEK_STATUS_ERROR"), DEC("SEEK_STATUS_UNSUPPORTED") };
const char * const FLAC__StreamDecoderTellStatusString[] = { DEC("TELL_STATUS_OK"), DEC("TELL_STATUS_ERROR"), DEC("TELL_STATUS_UNSUPPORTED") };
const char * const FLAC__StreamDecoderLengthStatusString[] = { DEC("LENGTH_STATUS_OK"), DEC("LENGTH_STATUS_ERROR"), DEC("LENGTH_STATUS_UNSUPPORTED") };
const char * const FLAC__StreamDecoderWriteStatusString[] = { DEC("WRITE_STATUS_CONTINUE"), DEC("WRITE_STATUS_ABORT") };
#undef ENC
#undef DEC

/* ---- range checks ---- */

/* format.c:213-220: 1 .. 655350 Hz, the largest rate a frame header can state (16 bits of tens of Hz) */
FLAC__bool FLAC__format_sample_rate_is_valid(unsigned sample_rate) { return sample_rate >= 1u && sample_rate <= FLAC__MAX_SAMPLE_RATE; }

/* format.c:222-230: the streamable subset allows blocks up to 16384 samples, up to 4608 at 48 kHz and below */
FLAC__bool FLAC__format_blocksize_is_subset(unsigned blocksize, unsigned sample_rate) { return blocksize <= (sample_rate <= 48000u ? 4608u : 16384u); }

/* format.c:232-245: a rate of 65536 Hz or more must be expressible by a frame header: in kHz or in tens of Hz */
FLAC__bool FLAC__format_sample_rate_is_subset(unsigned sample_rate)
{
	if(!FLAC__format_sample_rate_is_valid(sample_rate)) return 0;
	return sample_rate < 65536u || sample_rate % 10u == 0; /* a multiple of 1000 is a multiple of 10 */
}

/* format.c:538-546: a block of N samples can be cut into 2^k equal partitions for k up to the number of
 * trailing zero bits of N, and the field holds at most 15 */
unsigned FLAC__format_get_max_rice_partition_order_from_blocksize(unsigned blocksize)
{
	unsigned order = 0;
	while(order < FLAC__MAX_RICE_PARTITION_ORDER && !((blocksize >> order) & 1u)) order++; /* callers pass N >= 1; N = 0, on which the reference's loop never ends, stops at the cap here */
	return order;
}

/* format.c:548-561: the first partition holds (N >> k) - order residuals and must not be empty or negative */
unsigned FLAC__format_get_max_rice_partition_order_from_blocksize_limited_max_and_predictor_order(unsigned limit, unsigned blocksize, unsigned predictor_order)
{
	while(limit > 0 && (blocksize >> limit) <= predictor_order) limit--;
	return limit;
}

/* format.c:528-536 */
unsigned FLAC__format_get_max_rice_partition_order(unsigned blocksize, unsigned predictor_order)
{
	return FLAC__format_get_max_rice_partition_order_from_blocksize_limited_max_and_predictor_order(
		FLAC__format_get_max_rice_partition_order_from_blocksize(blocksize), blocksize, predictor_order);
}
