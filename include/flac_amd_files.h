/* Whole-file front end over the libFLAC-style objects of flac_amd_stream.h.
 *
 * flacamd_encode_file does what `flac [options] in.wav -o out.flac` does in the reference's
 * command-line tool for PCM inputs (src/flac/encode.c: WAVE/RF64/Wave64/AIFF/raw parsing :188-907, sample
 * formatting :2338-2428, metadata layout :2016-2053, encoder set-up :2066-2149) and writes the
 * same bytes: STREAMINFO, the seek table (a point every 10 s unless told otherwise), the tag
 * block, padding, frames.  The input file is memory-mapped and handed to the encoder object in
 * pieces of a whole device batch, instead of the tool's 2048-sample process() calls.
 * flacamd_decode_file is `flac -d`: src/flac/decode.c :543-790 (WAVE / AIFF headers),
 * :964-1262 (sample formatting).
 *
 * --skip / --until take the tool's values (src/flac/utils.c:170-219; encode.c:1091-1150, :2246-2289;
 * decode.c:392-399, :1069-1081, :1272-1309).
 *
 * A native FLAC file as input is decoded on the device and coded again, its metadata kept, replaced or dropped by the
 * tool's rules (src/flac/encode.c:907-958, :1746-2015).
 *
 * Not provided: Ogg,
 * --sector-align, ReplayGain, cue sheet files, pictures, foreign metadata.
 */
#ifndef FLAC_AMD_FILES_H
#define FLAC_AMD_FILES_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { FLACAMD_FILE_AUTO = 0, FLACAMD_FILE_RAW = 1, FLACAMD_FILE_WAVE = 2, FLACAMD_FILE_AIFF = 3,
       FLACAMD_FILE_RF64 = 4, FLACAMD_FILE_WAVE64 = 5,
       FLACAMD_FILE_FLAC = 6 };   /* encode only: a native FLAC file as input */   /* the 64-bit WAVE variants (--force-rf64-format, --force-wave64-format) */

typedef struct {
	unsigned compression_level;     /* 0..8: flac -0 .. -8 */
	int verify;                     /* -V */
	int lax;                        /* --lax */
	int padding;                    /* -P: <0 the tool's default (8192, 65536 from 20 minutes up), 0 --no-padding, else bytes */
	const char *seekpoints;         /* NULL: the default "10s;".  "": --no-seektable.  Else the -S values, each
	                                   followed by ';' as the tool joins them: "X;", "25x;", "0.5s;", "123456;" */
	const char * const *tags;       /* -T FIELD=VALUE, UTF-8 */
	unsigned num_tags;
	unsigned blocksize;             /* -b, 0 = the preset's */
	int input_format;               /* FLACAMD_FILE_*: AUTO looks at the first 12 bytes; raw input must say RAW */
	unsigned raw_channels, raw_bps, raw_sample_rate;   /* --channels --bps --sample-rate */
	int raw_big_endian, raw_unsigned;                  /* --endian --sign */
	int device;                     /* HIP device */
	unsigned batch_frames;          /* frames per device batch, 0 = 4096 */
	const char *skip;               /* --skip: "123456" (samples) or "mm:ss.sss"; NULL = from the start */
	const char *until;              /* --until: the same, or "+..." counted from the skip point, "-..." from the end; NULL = to the end */
} flacamd_file_options;

typedef struct {
	uint64_t samples;               /* wide samples coded / decoded */
	uint64_t bytes_written;         /* size of the output file */
	unsigned channels, bits_per_sample, sample_rate;
	int md5_ok;                     /* decode: 1 if the STREAMINFO MD5 matched (or was absent) */
	char error[256];                /* first error in words, "" if none */
} flacamd_file_result;

/* Defaults of the tool: level 5, default seek table and padding, format by magic. */
void flacamd_file_options_init(flacamd_file_options *opt);

/* 0 on success, a FLACAMD_E_* code (flac_amd_engine.h) otherwise; res->error says what failed. */
int flacamd_encode_file(const char *in_path, const char *out_path, const flacamd_file_options *opt, flacamd_file_result *res);

/* Decode in_path (a native FLAC file) to a WAVE, RF64, Wave64, AIFF or raw PCM file (raw: little-endian, signed). */
int flacamd_decode_file(const char *in_path, const char *out_path, int output_format, int device, flacamd_file_result *res);
/* The same for the sample range [skip, until) (`flac -d --skip=.. --until=..`): the header promises the length of the range,
 * the MD5 verdict is not available for a partial decode (md5_ok = 1).  skip / until as in flacamd_file_options, NULL = not given. */
int flacamd_decode_file_range(const char *in_path, const char *out_path, int output_format, int device, const char *skip, const char *until, flacamd_file_result *res);

#ifdef __cplusplus
}
#endif
#endif
