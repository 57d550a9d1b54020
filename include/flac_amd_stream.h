/* libFLAC-style stream encoder object over the gfx950 frame engine.
 *
 * The functions below have the names, argument meaning and error behaviour of the
 * reference's public encoder API (include/FLAC/stream_encoder.h:693-1760, defined in
 * src/libFLAC/stream_encoder.c:529-2105), so a client written against the reference
 * recompiles against this header and links libflacamd.so instead of libFLAC.  What changes
 * behind the API: blocks are queued on the host, whole batches of frames are coded by the
 * HIP engine (include/flac_amd_engine.h), and the reference's per-frame write_frame_ logic
 * (stream_encoder.c:2431-2518) is then replayed frame by frame in stream order, so the
 * client sees the same callbacks with the same arguments in the same order, only later
 * (when a batch fills, or at finish()).
 *
 * Verify mode decodes every batch on the device before its frames are written and compares
 * the result with the queued input.  Ogg transport is not provided: the init_ogg_* functions
 * and set_ogg_serial_number are exported and behave like the reference built without libogg
 * (..._INIT_STATUS_UNSUPPORTED_CONTAINER, stream_encoder.c:659-662; false, :1378-1393), and
 * FLAC_API_SUPPORTS_OGG_FLAC reads 0.  set_do_escape_coding and
 * set_rice_parameter_search_dist are accepted and have no effect, exactly as in the
 * reference (stream_encoder.c:1635-1649, :1684-1698).
 */
#ifndef FLAC_AMD_STREAM_H
#define FLAC_AMD_STREAM_H
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int FLAC__bool;
typedef uint8_t FLAC__byte;
typedef int32_t FLAC__int32;
typedef uint32_t FLAC__uint32;
typedef uint64_t FLAC__uint64;
typedef uint8_t FLAC__uint8;
typedef uint16_t FLAC__uint16;
typedef int16_t FLAC__int16;
typedef int64_t FLAC__int64;

/* include/FLAC/export.h:58-83 */
#define FLAC_API_VERSION_CURRENT 10
#define FLAC_API_VERSION_REVISION 0
#define FLAC_API_VERSION_AGE 2
extern int FLAC_API_SUPPORTS_OGG_FLAC;   /* 0 */

/* the format's limits, include/FLAC/format.h:92-181 */
#define FLAC__MAX_METADATA_TYPE_CODE (126u)
#define FLAC__MIN_BLOCK_SIZE (16u)
#define FLAC__MAX_BLOCK_SIZE (65535u)
#define FLAC__SUBSET_MAX_BLOCK_SIZE_48000HZ (4608u)
#define FLAC__MAX_CHANNELS (8u)
#define FLAC__MIN_BITS_PER_SAMPLE (4u)
#define FLAC__MAX_BITS_PER_SAMPLE (32u)
#define FLAC__REFERENCE_CODEC_MAX_BITS_PER_SAMPLE (24u)
#define FLAC__MAX_SAMPLE_RATE (655350u)
#define FLAC__MAX_LPC_ORDER (32u)
#define FLAC__SUBSET_MAX_LPC_ORDER_48000HZ (12u)
#define FLAC__MIN_QLP_COEFF_PRECISION (5u)
#define FLAC__MAX_QLP_COEFF_PRECISION (15u)
#define FLAC__MAX_FIXED_ORDER (4u)
#define FLAC__MAX_RICE_PARTITION_ORDER (15u)
#define FLAC__SUBSET_MAX_RICE_PARTITION_ORDER (8u)
#define FLAC__STREAM_SYNC_LENGTH (4u)

/* include/FLAC/format.h:158-181, defined in format.c:56-67 */
extern const char *FLAC__VERSION_STRING;          /* "1.2.1" */
extern const char *FLAC__VENDOR_STRING;           /* "reference libFLAC 1.2.1 20070917": written into every VORBIS_COMMENT block */
extern const FLAC__byte FLAC__STREAM_SYNC_STRING[4]; /* "fLaC" */
extern const unsigned FLAC__STREAM_SYNC;          /* 0x664C6143 */
extern const unsigned FLAC__STREAM_SYNC_LEN;      /* 32 bits */

/* values and order of include/FLAC/stream_encoder.h:240-278 */
typedef enum {
	FLAC__STREAM_ENCODER_OK = 0,
	FLAC__STREAM_ENCODER_UNINITIALIZED,
	FLAC__STREAM_ENCODER_OGG_ERROR,
	FLAC__STREAM_ENCODER_VERIFY_DECODER_ERROR,
	FLAC__STREAM_ENCODER_VERIFY_MISMATCH_IN_AUDIO_DATA,
	FLAC__STREAM_ENCODER_CLIENT_ERROR,
	FLAC__STREAM_ENCODER_IO_ERROR,
	FLAC__STREAM_ENCODER_FRAMING_ERROR,
	FLAC__STREAM_ENCODER_MEMORY_ALLOCATION_ERROR
} FLAC__StreamEncoderState;

/* include/FLAC/stream_encoder.h:292-345 */
typedef enum {
	FLAC__STREAM_ENCODER_INIT_STATUS_OK = 0,
	FLAC__STREAM_ENCODER_INIT_STATUS_ENCODER_ERROR,
	FLAC__STREAM_ENCODER_INIT_STATUS_UNSUPPORTED_CONTAINER,
	FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_CALLBACKS,
	FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_NUMBER_OF_CHANNELS,
	FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_BITS_PER_SAMPLE,
	FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_SAMPLE_RATE,
	FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_BLOCK_SIZE,
	FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_MAX_LPC_ORDER,
	FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_QLP_COEFF_PRECISION,
	FLAC__STREAM_ENCODER_INIT_STATUS_BLOCK_SIZE_TOO_SMALL_FOR_LPC_ORDER,
	FLAC__STREAM_ENCODER_INIT_STATUS_NOT_STREAMABLE,
	FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_METADATA,
	FLAC__STREAM_ENCODER_INIT_STATUS_ALREADY_INITIALIZED
} FLAC__StreamEncoderInitStatus;

typedef enum { FLAC__STREAM_ENCODER_WRITE_STATUS_OK = 0, FLAC__STREAM_ENCODER_WRITE_STATUS_FATAL_ERROR } FLAC__StreamEncoderWriteStatus;
typedef enum { FLAC__STREAM_ENCODER_SEEK_STATUS_OK, FLAC__STREAM_ENCODER_SEEK_STATUS_ERROR, FLAC__STREAM_ENCODER_SEEK_STATUS_UNSUPPORTED } FLAC__StreamEncoderSeekStatus;
typedef enum { FLAC__STREAM_ENCODER_TELL_STATUS_OK, FLAC__STREAM_ENCODER_TELL_STATUS_ERROR, FLAC__STREAM_ENCODER_TELL_STATUS_UNSUPPORTED } FLAC__StreamEncoderTellStatus;

/* include/FLAC/stream_encoder.h:347-365: only the Ogg inits take a read callback */
typedef enum { FLAC__STREAM_ENCODER_READ_STATUS_CONTINUE, FLAC__STREAM_ENCODER_READ_STATUS_END_OF_STREAM, FLAC__STREAM_ENCODER_READ_STATUS_ABORT, FLAC__STREAM_ENCODER_READ_STATUS_UNSUPPORTED } FLAC__StreamEncoderReadStatus;

extern const char * const FLAC__StreamEncoderStateString[];
extern const char * const FLAC__StreamEncoderInitStatusString[];
extern const char * const FLAC__StreamEncoderReadStatusString[];
extern const char * const FLAC__StreamEncoderWriteStatusString[];
extern const char * const FLAC__StreamEncoderSeekStatusString[];
extern const char * const FLAC__StreamEncoderTellStatusString[];

/* Metadata blocks: types, field names and field order of include/FLAC/format.h:488-850, so
 * that block objects built by client code (or by the reference's metadata_object helpers) can
 * be handed to FLAC__stream_encoder_set_metadata unchanged. */
typedef enum {
	FLAC__METADATA_TYPE_STREAMINFO = 0, FLAC__METADATA_TYPE_PADDING = 1, FLAC__METADATA_TYPE_APPLICATION = 2,
	FLAC__METADATA_TYPE_SEEKTABLE = 3, FLAC__METADATA_TYPE_VORBIS_COMMENT = 4, FLAC__METADATA_TYPE_CUESHEET = 5,
	FLAC__METADATA_TYPE_PICTURE = 6, FLAC__METADATA_TYPE_UNDEFINED = 7
} FLAC__MetadataType;
extern const char * const FLAC__MetadataTypeString[];
extern const unsigned FLAC__STREAM_METADATA_IS_LAST_LEN, FLAC__STREAM_METADATA_TYPE_LEN, FLAC__STREAM_METADATA_LENGTH_LEN; /* 1, 7, 24 bits */

typedef struct {
	unsigned min_blocksize, max_blocksize;
	unsigned min_framesize, max_framesize;
	unsigned sample_rate, channels, bits_per_sample;
	FLAC__uint64 total_samples;
	FLAC__byte md5sum[16];
} FLAC__StreamMetadata_StreamInfo;
#define FLAC__STREAM_METADATA_STREAMINFO_LENGTH (34u)
#define FLAC__STREAM_METADATA_HEADER_LENGTH (4u)

typedef struct { int dummy; } FLAC__StreamMetadata_Padding;                    /* `length` zero bytes */
typedef struct { FLAC__byte id[4]; FLAC__byte *data; } FLAC__StreamMetadata_Application; /* data: length-4 bytes */

typedef struct { FLAC__uint64 sample_number, stream_offset; unsigned frame_samples; } FLAC__StreamMetadata_SeekPoint;
#define FLAC__STREAM_METADATA_SEEKPOINT_LENGTH (18u)
#define FLAC__STREAM_METADATA_SEEKPOINT_PLACEHOLDER_VALUE 0xffffffffffffffffull
extern const FLAC__uint64 FLAC__STREAM_METADATA_SEEKPOINT_PLACEHOLDER;
typedef struct { unsigned num_points; FLAC__StreamMetadata_SeekPoint *points; } FLAC__StreamMetadata_SeekTable;

typedef struct { FLAC__uint32 length; FLAC__byte *entry; } FLAC__StreamMetadata_VorbisComment_Entry;
typedef struct {
	FLAC__StreamMetadata_VorbisComment_Entry vendor_string;   /* ignored on write: the library's own is written */
	FLAC__uint32 num_comments;
	FLAC__StreamMetadata_VorbisComment_Entry *comments;
} FLAC__StreamMetadata_VorbisComment;

typedef struct { FLAC__uint64 offset; FLAC__byte number; } FLAC__StreamMetadata_CueSheet_Index;
typedef struct {
	FLAC__uint64 offset;
	FLAC__byte number;
	char isrc[13];
	unsigned type:1;
	unsigned pre_emphasis:1;
	FLAC__byte num_indices;
	FLAC__StreamMetadata_CueSheet_Index *indices;
} FLAC__StreamMetadata_CueSheet_Track;
typedef struct {
	char media_catalog_number[129];
	FLAC__uint64 lead_in;
	FLAC__bool is_cd;
	unsigned num_tracks;
	FLAC__StreamMetadata_CueSheet_Track *tracks;
} FLAC__StreamMetadata_CueSheet;

typedef enum {   /* the ID3v2 APIC picture kinds, include/FLAC/format.h:729-752 */
	FLAC__STREAM_METADATA_PICTURE_TYPE_OTHER = 0, FLAC__STREAM_METADATA_PICTURE_TYPE_FILE_ICON_STANDARD = 1,
	FLAC__STREAM_METADATA_PICTURE_TYPE_FILE_ICON = 2, FLAC__STREAM_METADATA_PICTURE_TYPE_FRONT_COVER = 3,
	FLAC__STREAM_METADATA_PICTURE_TYPE_BACK_COVER = 4, FLAC__STREAM_METADATA_PICTURE_TYPE_LEAFLET_PAGE = 5,
	FLAC__STREAM_METADATA_PICTURE_TYPE_MEDIA = 6, FLAC__STREAM_METADATA_PICTURE_TYPE_LEAD_ARTIST = 7,
	FLAC__STREAM_METADATA_PICTURE_TYPE_ARTIST = 8, FLAC__STREAM_METADATA_PICTURE_TYPE_CONDUCTOR = 9,
	FLAC__STREAM_METADATA_PICTURE_TYPE_BAND = 10, FLAC__STREAM_METADATA_PICTURE_TYPE_COMPOSER = 11,
	FLAC__STREAM_METADATA_PICTURE_TYPE_LYRICIST = 12, FLAC__STREAM_METADATA_PICTURE_TYPE_RECORDING_LOCATION = 13,
	FLAC__STREAM_METADATA_PICTURE_TYPE_DURING_RECORDING = 14, FLAC__STREAM_METADATA_PICTURE_TYPE_DURING_PERFORMANCE = 15,
	FLAC__STREAM_METADATA_PICTURE_TYPE_VIDEO_SCREEN_CAPTURE = 16, FLAC__STREAM_METADATA_PICTURE_TYPE_FISH = 17,
	FLAC__STREAM_METADATA_PICTURE_TYPE_ILLUSTRATION = 18, FLAC__STREAM_METADATA_PICTURE_TYPE_BAND_LOGOTYPE = 19,
	FLAC__STREAM_METADATA_PICTURE_TYPE_PUBLISHER_LOGOTYPE = 20, FLAC__STREAM_METADATA_PICTURE_TYPE_UNDEFINED = 21
} FLAC__StreamMetadata_Picture_Type;
extern const char * const FLAC__StreamMetadata_Picture_TypeString[];   /* 21 names */
typedef struct {
	FLAC__StreamMetadata_Picture_Type type;
	char *mime_type;            /* NUL-terminated printable ASCII */
	FLAC__byte *description;    /* NUL-terminated UTF-8 */
	FLAC__uint32 width, height, depth, colors;
	FLAC__uint32 data_length;
	FLAC__byte *data;
} FLAC__StreamMetadata_Picture;

typedef struct { FLAC__byte *data; } FLAC__StreamMetadata_Unknown;

typedef struct {
	FLAC__MetadataType type;
	FLAC__bool is_last;      /* set by the encoder when the block is written */
	unsigned length;         /* bytes of block data, without the 4-byte block header */
	union {
		FLAC__StreamMetadata_StreamInfo stream_info;
		FLAC__StreamMetadata_Padding padding;
		FLAC__StreamMetadata_Application application;
		FLAC__StreamMetadata_SeekTable seek_table;
		FLAC__StreamMetadata_VorbisComment vorbis_comment;
		FLAC__StreamMetadata_CueSheet cue_sheet;
		FLAC__StreamMetadata_Picture picture;
		FLAC__StreamMetadata_Unknown unknown;
	} data;
} FLAC__StreamMetadata;

/* Bit widths of the fields of the metadata blocks as they lie in a stream (include/FLAC/format.h:506-860,
 * values in format.c:69-119) */
extern const unsigned
	FLAC__STREAM_METADATA_STREAMINFO_MIN_BLOCK_SIZE_LEN, FLAC__STREAM_METADATA_STREAMINFO_MAX_BLOCK_SIZE_LEN,   /* 16, 16 */
	FLAC__STREAM_METADATA_STREAMINFO_MIN_FRAME_SIZE_LEN, FLAC__STREAM_METADATA_STREAMINFO_MAX_FRAME_SIZE_LEN,   /* 24, 24 */
	FLAC__STREAM_METADATA_STREAMINFO_SAMPLE_RATE_LEN, FLAC__STREAM_METADATA_STREAMINFO_CHANNELS_LEN,            /* 20, 3 */
	FLAC__STREAM_METADATA_STREAMINFO_BITS_PER_SAMPLE_LEN, FLAC__STREAM_METADATA_STREAMINFO_TOTAL_SAMPLES_LEN,   /* 5, 36 */
	FLAC__STREAM_METADATA_STREAMINFO_MD5SUM_LEN,                                                                /* 128 */
	FLAC__STREAM_METADATA_APPLICATION_ID_LEN,                                                                   /* 32 */
	FLAC__STREAM_METADATA_SEEKPOINT_SAMPLE_NUMBER_LEN, FLAC__STREAM_METADATA_SEEKPOINT_STREAM_OFFSET_LEN,       /* 64, 64 */
	FLAC__STREAM_METADATA_SEEKPOINT_FRAME_SAMPLES_LEN,                                                          /* 16 */
	FLAC__STREAM_METADATA_VORBIS_COMMENT_ENTRY_LENGTH_LEN, FLAC__STREAM_METADATA_VORBIS_COMMENT_NUM_COMMENTS_LEN, /* 32, 32 */
	FLAC__STREAM_METADATA_CUESHEET_INDEX_OFFSET_LEN, FLAC__STREAM_METADATA_CUESHEET_INDEX_NUMBER_LEN,           /* 64, 8 */
	FLAC__STREAM_METADATA_CUESHEET_INDEX_RESERVED_LEN,                                                          /* 24 */
	FLAC__STREAM_METADATA_CUESHEET_TRACK_OFFSET_LEN, FLAC__STREAM_METADATA_CUESHEET_TRACK_NUMBER_LEN,           /* 64, 8 */
	FLAC__STREAM_METADATA_CUESHEET_TRACK_ISRC_LEN, FLAC__STREAM_METADATA_CUESHEET_TRACK_TYPE_LEN,               /* 96, 1 */
	FLAC__STREAM_METADATA_CUESHEET_TRACK_PRE_EMPHASIS_LEN, FLAC__STREAM_METADATA_CUESHEET_TRACK_RESERVED_LEN,   /* 1, 110 */
	FLAC__STREAM_METADATA_CUESHEET_TRACK_NUM_INDICES_LEN,                                                       /* 8 */
	FLAC__STREAM_METADATA_CUESHEET_MEDIA_CATALOG_NUMBER_LEN, FLAC__STREAM_METADATA_CUESHEET_LEAD_IN_LEN,        /* 1024, 64 */
	FLAC__STREAM_METADATA_CUESHEET_IS_CD_LEN, FLAC__STREAM_METADATA_CUESHEET_RESERVED_LEN,                      /* 1, 2071 */
	FLAC__STREAM_METADATA_CUESHEET_NUM_TRACKS_LEN,                                                              /* 8 */
	FLAC__STREAM_METADATA_PICTURE_TYPE_LEN, FLAC__STREAM_METADATA_PICTURE_MIME_TYPE_LENGTH_LEN,                 /* 32 each */
	FLAC__STREAM_METADATA_PICTURE_DESCRIPTION_LENGTH_LEN, FLAC__STREAM_METADATA_PICTURE_WIDTH_LEN,
	FLAC__STREAM_METADATA_PICTURE_HEIGHT_LEN, FLAC__STREAM_METADATA_PICTURE_DEPTH_LEN,
	FLAC__STREAM_METADATA_PICTURE_COLORS_LEN, FLAC__STREAM_METADATA_PICTURE_DATA_LENGTH_LEN;

/* format.c:248-314, :396-523: the legality checks init applies to client blocks, and the sort the
 * encoder applies to a seek table before it is written back at finish(). */
FLAC__bool FLAC__format_seektable_is_legal(const FLAC__StreamMetadata_SeekTable *seek_table);
unsigned FLAC__format_seektable_sort(FLAC__StreamMetadata_SeekTable *seek_table);
FLAC__bool FLAC__format_vorbiscomment_entry_is_legal(const FLAC__byte *entry, unsigned length);
FLAC__bool FLAC__format_cuesheet_is_legal(const FLAC__StreamMetadata_CueSheet *cue_sheet, FLAC__bool check_cd_da_subset, const char **violation);
FLAC__bool FLAC__format_picture_is_legal(const FLAC__StreamMetadata_Picture *picture, const char **violation);
/* format.c:363-394: the two halves of a tag on their own; length (unsigned)(-1) = value is NUL-terminated */
FLAC__bool FLAC__format_vorbiscomment_entry_name_is_legal(const char *name);
FLAC__bool FLAC__format_vorbiscomment_entry_value_is_legal(const FLAC__byte *value, unsigned length);
/* format.c:213-245: range checks of the format and of its streamable subset */
FLAC__bool FLAC__format_sample_rate_is_valid(unsigned sample_rate);
FLAC__bool FLAC__format_sample_rate_is_subset(unsigned sample_rate);
FLAC__bool FLAC__format_blocksize_is_subset(unsigned blocksize, unsigned sample_rate);
/* format.c:528-561 (private/format.h in the reference, exported by its library): the finest partition order a
 * block of `blocksize` samples allows, and allows under a predictor of `predictor_order` */
unsigned FLAC__format_get_max_rice_partition_order(unsigned blocksize, unsigned predictor_order);
unsigned FLAC__format_get_max_rice_partition_order_from_blocksize(unsigned blocksize);
unsigned FLAC__format_get_max_rice_partition_order_from_blocksize_limited_max_and_predictor_order(unsigned limit, unsigned blocksize, unsigned predictor_order);

typedef struct FLAC__StreamEncoder FLAC__StreamEncoder;

/* include/FLAC/stream_encoder.h:518-677 */
typedef FLAC__StreamEncoderWriteStatus (*FLAC__StreamEncoderWriteCallback)(const FLAC__StreamEncoder *encoder, const FLAC__byte buffer[], size_t bytes, unsigned samples, unsigned current_frame, void *client_data);
typedef FLAC__StreamEncoderSeekStatus (*FLAC__StreamEncoderSeekCallback)(const FLAC__StreamEncoder *encoder, FLAC__uint64 absolute_byte_offset, void *client_data);
typedef FLAC__StreamEncoderTellStatus (*FLAC__StreamEncoderTellCallback)(const FLAC__StreamEncoder *encoder, FLAC__uint64 *absolute_byte_offset, void *client_data);
typedef void (*FLAC__StreamEncoderMetadataCallback)(const FLAC__StreamEncoder *encoder, const FLAC__StreamMetadata *metadata, void *client_data);
typedef void (*FLAC__StreamEncoderProgressCallback)(const FLAC__StreamEncoder *encoder, FLAC__uint64 bytes_written, FLAC__uint64 samples_written, unsigned frames_written, unsigned total_frames_estimate, void *client_data);

FLAC__StreamEncoder *FLAC__stream_encoder_new(void);
void FLAC__stream_encoder_delete(FLAC__StreamEncoder *encoder);

/* setters: legal only before init, return false afterwards (stream_encoder.c:1430-1745) */
FLAC__bool FLAC__stream_encoder_set_verify(FLAC__StreamEncoder *encoder, FLAC__bool value);
FLAC__bool FLAC__stream_encoder_set_streamable_subset(FLAC__StreamEncoder *encoder, FLAC__bool value);
FLAC__bool FLAC__stream_encoder_set_channels(FLAC__StreamEncoder *encoder, unsigned value);
FLAC__bool FLAC__stream_encoder_set_bits_per_sample(FLAC__StreamEncoder *encoder, unsigned value);
FLAC__bool FLAC__stream_encoder_set_sample_rate(FLAC__StreamEncoder *encoder, unsigned value);
FLAC__bool FLAC__stream_encoder_set_compression_level(FLAC__StreamEncoder *encoder, unsigned value);
FLAC__bool FLAC__stream_encoder_set_blocksize(FLAC__StreamEncoder *encoder, unsigned value);
FLAC__bool FLAC__stream_encoder_set_do_mid_side_stereo(FLAC__StreamEncoder *encoder, FLAC__bool value);
FLAC__bool FLAC__stream_encoder_set_loose_mid_side_stereo(FLAC__StreamEncoder *encoder, FLAC__bool value);
FLAC__bool FLAC__stream_encoder_set_apodization(FLAC__StreamEncoder *encoder, const char *specification);
FLAC__bool FLAC__stream_encoder_set_max_lpc_order(FLAC__StreamEncoder *encoder, unsigned value);
FLAC__bool FLAC__stream_encoder_set_qlp_coeff_precision(FLAC__StreamEncoder *encoder, unsigned value);
FLAC__bool FLAC__stream_encoder_set_do_qlp_coeff_prec_search(FLAC__StreamEncoder *encoder, FLAC__bool value);
FLAC__bool FLAC__stream_encoder_set_do_escape_coding(FLAC__StreamEncoder *encoder, FLAC__bool value);
FLAC__bool FLAC__stream_encoder_set_do_exhaustive_model_search(FLAC__StreamEncoder *encoder, FLAC__bool value);
FLAC__bool FLAC__stream_encoder_set_min_residual_partition_order(FLAC__StreamEncoder *encoder, unsigned value);
FLAC__bool FLAC__stream_encoder_set_max_residual_partition_order(FLAC__StreamEncoder *encoder, unsigned value);
FLAC__bool FLAC__stream_encoder_set_rice_parameter_search_dist(FLAC__StreamEncoder *encoder, unsigned value);
FLAC__bool FLAC__stream_encoder_set_total_samples_estimate(FLAC__StreamEncoder *encoder, FLAC__uint64 value);
/* :1711-1741: the pointer array is copied, the blocks are not: they must stay valid until finish(), and a
 * SEEKTABLE block is filled in (and sorted) in place */
FLAC__bool FLAC__stream_encoder_set_metadata(FLAC__StreamEncoder *encoder, FLAC__StreamMetadata **metadata, unsigned num_blocks);
/* semi-private hooks used by the reference's tools (stream_encoder.c:1419, :1747-1778) */
FLAC__bool FLAC__stream_encoder_disable_constant_subframes(FLAC__StreamEncoder *encoder, FLAC__bool value);
FLAC__bool FLAC__stream_encoder_disable_fixed_subframes(FLAC__StreamEncoder *encoder, FLAC__bool value);
FLAC__bool FLAC__stream_encoder_disable_verbatim_subframes(FLAC__StreamEncoder *encoder, FLAC__bool value);
FLAC__bool FLAC__stream_encoder_set_do_md5(FLAC__StreamEncoder *encoder, FLAC__bool value);
FLAC__bool FLAC__stream_encoder_get_do_md5(const FLAC__StreamEncoder *encoder);   /* :1845-1851 */
/* :1378-1393 without libogg: false, nothing stored */
FLAC__bool FLAC__stream_encoder_set_ogg_serial_number(FLAC__StreamEncoder *encoder, long serial_number);

FLAC__StreamEncoderState FLAC__stream_encoder_get_state(const FLAC__StreamEncoder *encoder);
const char *FLAC__stream_encoder_get_resolved_state_string(const FLAC__StreamEncoder *encoder);
FLAC__bool FLAC__stream_encoder_get_verify(const FLAC__StreamEncoder *encoder);
FLAC__bool FLAC__stream_encoder_get_streamable_subset(const FLAC__StreamEncoder *encoder);
unsigned FLAC__stream_encoder_get_channels(const FLAC__StreamEncoder *encoder);
unsigned FLAC__stream_encoder_get_bits_per_sample(const FLAC__StreamEncoder *encoder);
unsigned FLAC__stream_encoder_get_sample_rate(const FLAC__StreamEncoder *encoder);
unsigned FLAC__stream_encoder_get_blocksize(const FLAC__StreamEncoder *encoder);
FLAC__bool FLAC__stream_encoder_get_do_mid_side_stereo(const FLAC__StreamEncoder *encoder);
FLAC__bool FLAC__stream_encoder_get_loose_mid_side_stereo(const FLAC__StreamEncoder *encoder);
unsigned FLAC__stream_encoder_get_max_lpc_order(const FLAC__StreamEncoder *encoder);
unsigned FLAC__stream_encoder_get_qlp_coeff_precision(const FLAC__StreamEncoder *encoder);
FLAC__bool FLAC__stream_encoder_get_do_qlp_coeff_prec_search(const FLAC__StreamEncoder *encoder);
FLAC__bool FLAC__stream_encoder_get_do_escape_coding(const FLAC__StreamEncoder *encoder);
FLAC__bool FLAC__stream_encoder_get_do_exhaustive_model_search(const FLAC__StreamEncoder *encoder);
unsigned FLAC__stream_encoder_get_min_residual_partition_order(const FLAC__StreamEncoder *encoder);
unsigned FLAC__stream_encoder_get_max_residual_partition_order(const FLAC__StreamEncoder *encoder);
unsigned FLAC__stream_encoder_get_rice_parameter_search_dist(const FLAC__StreamEncoder *encoder);
FLAC__uint64 FLAC__stream_encoder_get_total_samples_estimate(const FLAC__StreamEncoder *encoder);

FLAC__StreamEncoderInitStatus FLAC__stream_encoder_init_stream(FLAC__StreamEncoder *encoder, FLAC__StreamEncoderWriteCallback write_callback, FLAC__StreamEncoderSeekCallback seek_callback, FLAC__StreamEncoderTellCallback tell_callback, FLAC__StreamEncoderMetadataCallback metadata_callback, void *client_data);
FLAC__StreamEncoderInitStatus FLAC__stream_encoder_init_FILE(FLAC__StreamEncoder *encoder, FILE *file, FLAC__StreamEncoderProgressCallback progress_callback, void *client_data);
FLAC__StreamEncoderInitStatus FLAC__stream_encoder_init_file(FLAC__StreamEncoder *encoder, const char *filename, FLAC__StreamEncoderProgressCallback progress_callback, void *client_data);

/* The Ogg inits of a library without Ogg transport (stream_encoder.c:1156-1308 over :640-662): the entrance
 * checks run (ALREADY_INITIALIZED; init_ogg_file creates / truncates the file and ENCODER_ERROR + state IO_ERROR
 * if that fails; the FILE is kept and closed by finish() / delete()), then ..._UNSUPPORTED_CONTAINER. */
typedef FLAC__StreamEncoderReadStatus (*FLAC__StreamEncoderReadCallback)(const FLAC__StreamEncoder *encoder, FLAC__byte buffer[], size_t *bytes, void *client_data);
FLAC__StreamEncoderInitStatus FLAC__stream_encoder_init_ogg_stream(FLAC__StreamEncoder *encoder, FLAC__StreamEncoderReadCallback read_callback, FLAC__StreamEncoderWriteCallback write_callback, FLAC__StreamEncoderSeekCallback seek_callback, FLAC__StreamEncoderTellCallback tell_callback, FLAC__StreamEncoderMetadataCallback metadata_callback, void *client_data);
FLAC__StreamEncoderInitStatus FLAC__stream_encoder_init_ogg_FILE(FLAC__StreamEncoder *encoder, FILE *file, FLAC__StreamEncoderProgressCallback progress_callback, void *client_data);
FLAC__StreamEncoderInitStatus FLAC__stream_encoder_init_ogg_file(FLAC__StreamEncoder *encoder, const char *filename, FLAC__StreamEncoderProgressCallback progress_callback, void *client_data);

FLAC__bool FLAC__stream_encoder_process(FLAC__StreamEncoder *encoder, const FLAC__int32 * const buffer[], unsigned samples);
FLAC__bool FLAC__stream_encoder_process_interleaved(FLAC__StreamEncoder *encoder, const FLAC__int32 buffer[], unsigned samples);
FLAC__bool FLAC__stream_encoder_finish(FLAC__StreamEncoder *encoder);

/* ---- additions of this implementation (no counterpart in the reference) ---- */
/* HIP device to run on (default 0) and frames per engine batch (default 2048). */
FLAC__bool flacamd_stream_encoder_set_device(FLAC__StreamEncoder *encoder, int device);
FLAC__bool flacamd_stream_encoder_set_batch_frames(FLAC__StreamEncoder *encoder, unsigned frames);
/* Several GPUs of one node: batches of frames go to the listed devices in turn and their frames are written out
 * in stream order (frames are independent given their samples and number, stream_encoder.c:2866-2923; batch
 * borders are kept on loose mid/side decision periods, :2958-2967).  The same id may be listed more than once. */
FLAC__bool flacamd_stream_encoder_set_devices(FLAC__StreamEncoder *encoder, unsigned num_devices, const int *device_ids);
/* Batches kept in flight per device (2..8, default 3 on one device, 2 on several): while one is coded, the next
 * is uploaded and filled and the one before is downloaded. */
FLAC__bool flacamd_stream_encoder_set_batches_in_flight(FLAC__StreamEncoder *encoder, unsigned per_device);
/* Threads (1..8, default 4, the calling one included) over which the copy of a large process_interleaved call
 * out of the client's buffer is spread; calls below half a million values are copied by the caller alone. */
FLAC__bool flacamd_stream_encoder_set_fill_threads(FLAC__StreamEncoder *encoder, unsigned threads);

/* ======================================================================================
 * libFLAC-style stream decoder object over the frame-parallel device decoder.
 *
 * Names, argument meaning, callback types and state / status values of the reference's
 * public decoder API (include/FLAC/stream_decoder.h:203-437 enums, :516-726 callbacks,
 * :742-1551 functions; defined in src/libFLAC/stream_decoder.c:284-1245).  The client's read
 * callback is drained in large pieces, whole batches of frames are decoded on the device
 * (sync search, header CRC-8, Rice decoding, prediction restore, CRC-16, decorrelation), and
 * the reference's per-frame write_audio_frame_to_client_ logic (stream_decoder.c:2915-2990) is
 * replayed frame by frame in stream order: the write callback sees the same FLAC__FrameHeader
 * fields and the same planar FLAC__int32 buffers as with the reference.
 *
 * Differences, none silent: Ogg transport is not provided; a stream must begin with STREAMINFO;
 * callbacks arrive a batch of frames at a time.  Damaged input is handled as the reference handles
 * it (the same error reports, zeroed frames and resynchronisation points: the serial walk of
 * frame_sync_ / read_frame_ is replayed on the host over what the device decoded), frames whose
 * header differs from STREAMINFO are decoded with their own parameters, and FLAC__Frame carries
 * subframes[] (see FLAC__Frame below for the two pointers that stay NULL / point elsewhere).
 * ====================================================================================== */

typedef enum {
	FLAC__STREAM_DECODER_SEARCH_FOR_METADATA = 0,
	FLAC__STREAM_DECODER_READ_METADATA,
	FLAC__STREAM_DECODER_SEARCH_FOR_FRAME_SYNC,
	FLAC__STREAM_DECODER_READ_FRAME,
	FLAC__STREAM_DECODER_END_OF_STREAM,
	FLAC__STREAM_DECODER_OGG_ERROR,
	FLAC__STREAM_DECODER_SEEK_ERROR,
	FLAC__STREAM_DECODER_ABORTED,
	FLAC__STREAM_DECODER_MEMORY_ALLOCATION_ERROR,
	FLAC__STREAM_DECODER_UNINITIALIZED
} FLAC__StreamDecoderState;

typedef enum {
	FLAC__STREAM_DECODER_INIT_STATUS_OK = 0,
	FLAC__STREAM_DECODER_INIT_STATUS_UNSUPPORTED_CONTAINER,
	FLAC__STREAM_DECODER_INIT_STATUS_INVALID_CALLBACKS,
	FLAC__STREAM_DECODER_INIT_STATUS_MEMORY_ALLOCATION_ERROR,
	FLAC__STREAM_DECODER_INIT_STATUS_ERROR_OPENING_FILE,
	FLAC__STREAM_DECODER_INIT_STATUS_ALREADY_INITIALIZED
} FLAC__StreamDecoderInitStatus;

typedef enum { FLAC__STREAM_DECODER_READ_STATUS_CONTINUE, FLAC__STREAM_DECODER_READ_STATUS_END_OF_STREAM, FLAC__STREAM_DECODER_READ_STATUS_ABORT } FLAC__StreamDecoderReadStatus;
typedef enum { FLAC__STREAM_DECODER_SEEK_STATUS_OK, FLAC__STREAM_DECODER_SEEK_STATUS_ERROR, FLAC__STREAM_DECODER_SEEK_STATUS_UNSUPPORTED } FLAC__StreamDecoderSeekStatus;
typedef enum { FLAC__STREAM_DECODER_TELL_STATUS_OK, FLAC__STREAM_DECODER_TELL_STATUS_ERROR, FLAC__STREAM_DECODER_TELL_STATUS_UNSUPPORTED } FLAC__StreamDecoderTellStatus;
typedef enum { FLAC__STREAM_DECODER_LENGTH_STATUS_OK, FLAC__STREAM_DECODER_LENGTH_STATUS_ERROR, FLAC__STREAM_DECODER_LENGTH_STATUS_UNSUPPORTED } FLAC__StreamDecoderLengthStatus;
typedef enum { FLAC__STREAM_DECODER_WRITE_STATUS_CONTINUE, FLAC__STREAM_DECODER_WRITE_STATUS_ABORT } FLAC__StreamDecoderWriteStatus;
typedef enum {
	FLAC__STREAM_DECODER_ERROR_STATUS_LOST_SYNC,
	FLAC__STREAM_DECODER_ERROR_STATUS_BAD_HEADER,
	FLAC__STREAM_DECODER_ERROR_STATUS_FRAME_CRC_MISMATCH,
	FLAC__STREAM_DECODER_ERROR_STATUS_UNPARSEABLE_STREAM
} FLAC__StreamDecoderErrorStatus;

extern const char * const FLAC__StreamDecoderStateString[];
extern const char * const FLAC__StreamDecoderInitStatusString[];
extern const char * const FLAC__StreamDecoderErrorStatusString[];
extern const char * const FLAC__StreamDecoderReadStatusString[];
extern const char * const FLAC__StreamDecoderSeekStatusString[];
extern const char * const FLAC__StreamDecoderTellStatusString[];
extern const char * const FLAC__StreamDecoderLengthStatusString[];
extern const char * const FLAC__StreamDecoderWriteStatusString[];

/* include/FLAC/format.h:376-443 */
typedef enum {
	FLAC__CHANNEL_ASSIGNMENT_INDEPENDENT = 0,
	FLAC__CHANNEL_ASSIGNMENT_LEFT_SIDE = 1,
	FLAC__CHANNEL_ASSIGNMENT_RIGHT_SIDE = 2,
	FLAC__CHANNEL_ASSIGNMENT_MID_SIDE = 3
} FLAC__ChannelAssignment;
typedef enum { FLAC__FRAME_NUMBER_TYPE_FRAME_NUMBER, FLAC__FRAME_NUMBER_TYPE_SAMPLE_NUMBER } FLAC__FrameNumberType;
typedef struct {
	unsigned blocksize;
	unsigned sample_rate;
	unsigned channels;
	FLAC__ChannelAssignment channel_assignment;
	unsigned bits_per_sample;
	FLAC__FrameNumberType number_type;   /* always SAMPLE_NUMBER when handed to the client (stream_decoder.c:2422-2445) */
	union { FLAC__uint32 frame_number; FLAC__uint64 sample_number; } number;
	FLAC__uint8 crc;
} FLAC__FrameHeader;
typedef struct { FLAC__uint16 crc; } FLAC__FrameFooter;
extern const char * const FLAC__FrameNumberTypeString[];
/* the frame header's sync code and field widths (include/FLAC/format.h:445-470, format.c:121-132) */
extern const unsigned FLAC__FRAME_HEADER_SYNC;   /* 0x3ffe */
extern const unsigned
	FLAC__FRAME_HEADER_SYNC_LEN, FLAC__FRAME_HEADER_RESERVED_LEN, FLAC__FRAME_HEADER_BLOCKING_STRATEGY_LEN,      /* 14, 1, 1 */
	FLAC__FRAME_HEADER_BLOCK_SIZE_LEN, FLAC__FRAME_HEADER_SAMPLE_RATE_LEN, FLAC__FRAME_HEADER_CHANNEL_ASSIGNMENT_LEN, /* 4, 4, 4 */
	FLAC__FRAME_HEADER_BITS_PER_SAMPLE_LEN, FLAC__FRAME_HEADER_ZERO_PAD_LEN, FLAC__FRAME_HEADER_CRC_LEN,         /* 3, 1, 8 */
	FLAC__FRAME_FOOTER_CRC_LEN;                                                                                  /* 16 */

/* include/FLAC/format.h:190-352: what the decoder tells the client about every subframe of a frame */
typedef enum { FLAC__ENTROPY_CODING_METHOD_PARTITIONED_RICE = 0, FLAC__ENTROPY_CODING_METHOD_PARTITIONED_RICE2 = 1 } FLAC__EntropyCodingMethodType;
typedef struct {
	unsigned *parameters;       /* the Rice parameter of each partition (the escape code for an escaped one) */
	unsigned *raw_bits;         /* width of the samples of an escaped partition, 0 for the others */
	unsigned capacity_by_order; /* the arrays hold 2^capacity_by_order entries */
} FLAC__EntropyCodingMethod_PartitionedRiceContents;
typedef struct {
	unsigned order;
	const FLAC__EntropyCodingMethod_PartitionedRiceContents *contents;
} FLAC__EntropyCodingMethod_PartitionedRice;
typedef struct {
	FLAC__EntropyCodingMethodType type;
	union { FLAC__EntropyCodingMethod_PartitionedRice partitioned_rice; } data;
} FLAC__EntropyCodingMethod;
typedef enum { FLAC__SUBFRAME_TYPE_CONSTANT = 0, FLAC__SUBFRAME_TYPE_VERBATIM = 1, FLAC__SUBFRAME_TYPE_FIXED = 2, FLAC__SUBFRAME_TYPE_LPC = 3 } FLAC__SubframeType;
typedef struct { FLAC__int32 value; } FLAC__Subframe_Constant;
typedef struct { const FLAC__int32 *data; } FLAC__Subframe_Verbatim;
typedef struct {
	FLAC__EntropyCodingMethod entropy_coding_method;
	unsigned order;
	FLAC__int32 warmup[FLAC__MAX_FIXED_ORDER];
	const FLAC__int32 *residual;
} FLAC__Subframe_Fixed;
typedef struct {
	FLAC__EntropyCodingMethod entropy_coding_method;
	unsigned order;
	unsigned qlp_coeff_precision;
	int quantization_level;
	FLAC__int32 qlp_coeff[FLAC__MAX_LPC_ORDER];
	FLAC__int32 warmup[FLAC__MAX_LPC_ORDER];
	const FLAC__int32 *residual;
} FLAC__Subframe_LPC;
typedef struct {
	FLAC__SubframeType type;
	union {
		FLAC__Subframe_Constant constant;
		FLAC__Subframe_Fixed fixed;
		FLAC__Subframe_LPC lpc;
		FLAC__Subframe_Verbatim verbatim;
	} data;
	unsigned wasted_bits;
} FLAC__Subframe;
/* subframe and residual field widths, type byte patterns, escape parameters (include/FLAC/format.h:190-370, format.c:134-158) */
extern const unsigned
	FLAC__ENTROPY_CODING_METHOD_TYPE_LEN, FLAC__ENTROPY_CODING_METHOD_PARTITIONED_RICE_ORDER_LEN,                /* 2, 4 */
	FLAC__ENTROPY_CODING_METHOD_PARTITIONED_RICE_PARAMETER_LEN, FLAC__ENTROPY_CODING_METHOD_PARTITIONED_RICE2_PARAMETER_LEN, /* 4, 5 */
	FLAC__ENTROPY_CODING_METHOD_PARTITIONED_RICE_RAW_LEN,                                                        /* 5 */
	FLAC__ENTROPY_CODING_METHOD_PARTITIONED_RICE_ESCAPE_PARAMETER, FLAC__ENTROPY_CODING_METHOD_PARTITIONED_RICE2_ESCAPE_PARAMETER, /* 15, 31 */
	FLAC__SUBFRAME_LPC_QLP_COEFF_PRECISION_LEN, FLAC__SUBFRAME_LPC_QLP_SHIFT_LEN,                               /* 4, 5 */
	FLAC__SUBFRAME_ZERO_PAD_LEN, FLAC__SUBFRAME_TYPE_LEN, FLAC__SUBFRAME_WASTED_BITS_FLAG_LEN,                  /* 1, 6, 1 */
	FLAC__SUBFRAME_TYPE_CONSTANT_BYTE_ALIGNED_MASK, FLAC__SUBFRAME_TYPE_VERBATIM_BYTE_ALIGNED_MASK,             /* 0x00, 0x02 */
	FLAC__SUBFRAME_TYPE_FIXED_BYTE_ALIGNED_MASK, FLAC__SUBFRAME_TYPE_LPC_BYTE_ALIGNED_MASK;                     /* 0x10, 0x40 */
extern const char * const FLAC__SubframeTypeString[];
extern const char * const FLAC__EntropyCodingMethodTypeString[];
extern const char * const FLAC__ChannelAssignmentString[];

/* include/FLAC/format.h:472-476.  The decoder's write callback receives the frame as the reference's does: `residual`
 * of FIXED / LPC subframes and `data` of VERBATIM subframes hold what the stream carried (a second, sample-parallel
 * pass on the device forms them again from the restored signal and they come back with the samples; a client that
 * never reads them can switch the pass off with flacamd_stream_decoder_set_residuals(decoder, 0), after which
 * `residual` is NULL and `data` points at the channel's output samples).  Of a partitioned-Rice residual with a
 * partition order above 8 (outside the streamable subset) the first 256 parameters are filled in, the rest read 0. */
typedef struct {
	FLAC__FrameHeader header;
	FLAC__Subframe subframes[FLAC__MAX_CHANNELS];
	FLAC__FrameFooter footer;
} FLAC__Frame;

typedef struct FLAC__StreamDecoder FLAC__StreamDecoder;

typedef FLAC__StreamDecoderReadStatus (*FLAC__StreamDecoderReadCallback)(const FLAC__StreamDecoder *decoder, FLAC__byte buffer[], size_t *bytes, void *client_data);
typedef FLAC__StreamDecoderSeekStatus (*FLAC__StreamDecoderSeekCallback)(const FLAC__StreamDecoder *decoder, FLAC__uint64 absolute_byte_offset, void *client_data);
typedef FLAC__StreamDecoderTellStatus (*FLAC__StreamDecoderTellCallback)(const FLAC__StreamDecoder *decoder, FLAC__uint64 *absolute_byte_offset, void *client_data);
typedef FLAC__StreamDecoderLengthStatus (*FLAC__StreamDecoderLengthCallback)(const FLAC__StreamDecoder *decoder, FLAC__uint64 *stream_length, void *client_data);
typedef FLAC__bool (*FLAC__StreamDecoderEofCallback)(const FLAC__StreamDecoder *decoder, void *client_data);
typedef FLAC__StreamDecoderWriteStatus (*FLAC__StreamDecoderWriteCallback)(const FLAC__StreamDecoder *decoder, const FLAC__Frame *frame, const FLAC__int32 * const buffer[], void *client_data);
typedef void (*FLAC__StreamDecoderMetadataCallback)(const FLAC__StreamDecoder *decoder, const FLAC__StreamMetadata *metadata, void *client_data);
typedef void (*FLAC__StreamDecoderErrorCallback)(const FLAC__StreamDecoder *decoder, FLAC__StreamDecoderErrorStatus status, void *client_data);

FLAC__StreamDecoder *FLAC__stream_decoder_new(void);
void FLAC__stream_decoder_delete(FLAC__StreamDecoder *decoder);

FLAC__bool FLAC__stream_decoder_set_md5_checking(FLAC__StreamDecoder *decoder, FLAC__bool value);
/* the metadata filter (stream_decoder.c:762-877): by default only STREAMINFO reaches the metadata callback; blocks
 * handed to the callback are only valid during the call */
FLAC__bool FLAC__stream_decoder_set_metadata_respond(FLAC__StreamDecoder *decoder, int type);
FLAC__bool FLAC__stream_decoder_set_metadata_respond_application(FLAC__StreamDecoder *decoder, const FLAC__byte id[4]);
FLAC__bool FLAC__stream_decoder_set_metadata_respond_all(FLAC__StreamDecoder *decoder);
FLAC__bool FLAC__stream_decoder_set_metadata_ignore(FLAC__StreamDecoder *decoder, int type);
FLAC__bool FLAC__stream_decoder_set_metadata_ignore_application(FLAC__StreamDecoder *decoder, const FLAC__byte id[4]);
FLAC__bool FLAC__stream_decoder_set_metadata_ignore_all(FLAC__StreamDecoder *decoder);

FLAC__StreamDecoderState FLAC__stream_decoder_get_state(const FLAC__StreamDecoder *decoder);
const char *FLAC__stream_decoder_get_resolved_state_string(const FLAC__StreamDecoder *decoder);
FLAC__bool FLAC__stream_decoder_get_md5_checking(const FLAC__StreamDecoder *decoder);
FLAC__uint64 FLAC__stream_decoder_get_total_samples(const FLAC__StreamDecoder *decoder);
unsigned FLAC__stream_decoder_get_channels(const FLAC__StreamDecoder *decoder);
FLAC__ChannelAssignment FLAC__stream_decoder_get_channel_assignment(const FLAC__StreamDecoder *decoder);
unsigned FLAC__stream_decoder_get_bits_per_sample(const FLAC__StreamDecoder *decoder);
unsigned FLAC__stream_decoder_get_sample_rate(const FLAC__StreamDecoder *decoder);
unsigned FLAC__stream_decoder_get_blocksize(const FLAC__StreamDecoder *decoder);
FLAC__bool FLAC__stream_decoder_get_decode_position(const FLAC__StreamDecoder *decoder, FLAC__uint64 *position);

FLAC__StreamDecoderInitStatus FLAC__stream_decoder_init_stream(FLAC__StreamDecoder *decoder, FLAC__StreamDecoderReadCallback read_callback, FLAC__StreamDecoderSeekCallback seek_callback, FLAC__StreamDecoderTellCallback tell_callback, FLAC__StreamDecoderLengthCallback length_callback, FLAC__StreamDecoderEofCallback eof_callback, FLAC__StreamDecoderWriteCallback write_callback, FLAC__StreamDecoderMetadataCallback metadata_callback, FLAC__StreamDecoderErrorCallback error_callback, void *client_data);
FLAC__StreamDecoderInitStatus FLAC__stream_decoder_init_FILE(FLAC__StreamDecoder *decoder, FILE *file, FLAC__StreamDecoderWriteCallback write_callback, FLAC__StreamDecoderMetadataCallback metadata_callback, FLAC__StreamDecoderErrorCallback error_callback, void *client_data);
FLAC__StreamDecoderInitStatus FLAC__stream_decoder_init_file(FLAC__StreamDecoder *decoder, const char *filename, FLAC__StreamDecoderWriteCallback write_callback, FLAC__StreamDecoderMetadataCallback metadata_callback, FLAC__StreamDecoderErrorCallback error_callback, void *client_data);

/* The Ogg inits of a library without Ogg transport (stream_decoder.c:517-700 over :394-399): after the
 * entrance checks (ALREADY_INITIALIZED, INVALID_CALLBACKS, ERROR_OPENING_FILE; an opened file is kept and
 * closed by finish()) they return ..._UNSUPPORTED_CONTAINER; set_ogg_serial_number (:736-751) is false. */
FLAC__StreamDecoderInitStatus FLAC__stream_decoder_init_ogg_stream(FLAC__StreamDecoder *decoder, FLAC__StreamDecoderReadCallback read_callback, FLAC__StreamDecoderSeekCallback seek_callback, FLAC__StreamDecoderTellCallback tell_callback, FLAC__StreamDecoderLengthCallback length_callback, FLAC__StreamDecoderEofCallback eof_callback, FLAC__StreamDecoderWriteCallback write_callback, FLAC__StreamDecoderMetadataCallback metadata_callback, FLAC__StreamDecoderErrorCallback error_callback, void *client_data);
FLAC__StreamDecoderInitStatus FLAC__stream_decoder_init_ogg_FILE(FLAC__StreamDecoder *decoder, FILE *file, FLAC__StreamDecoderWriteCallback write_callback, FLAC__StreamDecoderMetadataCallback metadata_callback, FLAC__StreamDecoderErrorCallback error_callback, void *client_data);
FLAC__StreamDecoderInitStatus FLAC__stream_decoder_init_ogg_file(FLAC__StreamDecoder *decoder, const char *filename, FLAC__StreamDecoderWriteCallback write_callback, FLAC__StreamDecoderMetadataCallback metadata_callback, FLAC__StreamDecoderErrorCallback error_callback, void *client_data);
FLAC__bool FLAC__stream_decoder_set_ogg_serial_number(FLAC__StreamDecoder *decoder, long serial_number);

FLAC__bool FLAC__stream_decoder_finish(FLAC__StreamDecoder *decoder);
FLAC__bool FLAC__stream_decoder_flush(FLAC__StreamDecoder *decoder);
FLAC__bool FLAC__stream_decoder_reset(FLAC__StreamDecoder *decoder);
FLAC__bool FLAC__stream_decoder_process_single(FLAC__StreamDecoder *decoder);
FLAC__bool FLAC__stream_decoder_process_until_end_of_metadata(FLAC__StreamDecoder *decoder);
FLAC__bool FLAC__stream_decoder_process_until_end_of_stream(FLAC__StreamDecoder *decoder);
FLAC__bool FLAC__stream_decoder_skip_single_frame(FLAC__StreamDecoder *decoder);
FLAC__bool FLAC__stream_decoder_seek_absolute(FLAC__StreamDecoder *decoder, FLAC__uint64 sample);

/* additions of this implementation: HIP device (default 0), frames per device batch (default 2048) */
FLAC__bool flacamd_stream_decoder_set_device(FLAC__StreamDecoder *decoder, int device);
FLAC__bool flacamd_stream_decoder_set_batch_frames(FLAC__StreamDecoder *decoder, unsigned frames);
/* Whether FLAC__Frame carries the subframes' residuals (default on, as in the reference; the environment variable
 * FLACAMD_DECODER_RESIDUALS=0 makes off the default for clients that cannot call this).  Off saves the residual pass
 * and half of a batch's download. */
FLAC__bool flacamd_stream_decoder_set_residuals(FLAC__StreamDecoder *decoder, FLAC__bool value);

/* ---- encoder verify mode (declared here because they use the decoder's state type) ----
 * stream_encoder.c:1788-1827.  The verify decoder is the device decoder, run on every batch before
 * its frames reach the write callback; frames ahead of a failing frame are still written, like
 * in the reference, and the failing frame is not. */
FLAC__StreamDecoderState FLAC__stream_encoder_get_verify_decoder_state(const FLAC__StreamEncoder *encoder);
void FLAC__stream_encoder_get_verify_decoder_error_stats(const FLAC__StreamEncoder *encoder, FLAC__uint64 *absolute_sample, unsigned *frame_number, unsigned *channel, unsigned *sample, FLAC__int32 *expected, FLAC__int32 *got);
/* Test hook of this implementation (no counterpart in the reference, like its disable_* hooks a
 * way to reach a path that correct code never takes): XOR `mask` into the retained copy of one
 * input sample just before the comparison, as a bit flip in host memory would. */
FLAC__bool flacamd_stream_encoder_inject_verify_fault(FLAC__StreamEncoder *encoder, FLAC__uint64 absolute_sample, unsigned channel, FLAC__int32 mask);
/* Second test hook: XOR `mask` into byte `byte_index` of frame `frame_number` before it is verified. */
FLAC__bool flacamd_stream_encoder_inject_frame_fault(FLAC__StreamEncoder *encoder, unsigned frame_number, unsigned byte_index, unsigned mask);

#ifdef __cplusplus
}
#endif
#endif
