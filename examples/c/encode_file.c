/* A libFLAC client: 16-bit PCM WAVE file in, FLAC file out, through the stream encoder's file layer, 1024 samples
 * per process_interleaved call, verify on, with a progress callback.  Written against the libFLAC API only; it is
 * compiled twice - against include/flac_amd_stream.h + libflacamd.so, and against the reference's own headers and
 * library (the checker) - and both builds must write the same file (tests/test_gpu_c_client.py).  It follows the
 * call sequence of the reference's example client (examples/c/encode/file/main.c:84-158), minus the metadata
 * object helpers, which are outside this library.
 *   -DFLAC_ENCODER_HEADER='"flac_amd_stream.h"'  or  '"FLAC/stream_encoder.h"' */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include FLAC_ENCODER_HEADER

#define READSIZE 1024
static unsigned long long total_samples = 0;
static unsigned progress_calls = 0;

static void progress(const FLAC__StreamEncoder *enc, FLAC__uint64 bytes_written, FLAC__uint64 samples_written, unsigned frames_written, unsigned total_frames_estimate, void *client_data)
{
	(void)enc; (void)client_data; (void)bytes_written; (void)total_frames_estimate;
	if(samples_written > total_samples || frames_written == 0) { fprintf(stderr, "progress out of range\n"); exit(3); }
	progress_calls++;
}

int main(int argc, char **argv)
{
	unsigned char head[44], *raw; FLAC__int32 *pcm;
	unsigned rate, channels, bps, level; size_t left, i;
	FLAC__StreamEncoder *enc; FLAC__StreamEncoderInitStatus st; FLAC__bool ok = 1;
	FILE *in;
	if(argc < 3) { fprintf(stderr, "usage: %s in.wav out.flac [level]\n", argv[0]); return 2; }
	level = argc > 3 ? (unsigned)atoi(argv[3]) : 5u;
	if(!(in = fopen(argv[1], "rb"))) { perror(argv[1]); return 2; }
	if(fread(head, 1, 44, in) != 44 || memcmp(head, "RIFF", 4) || memcmp(head + 8, "WAVEfmt \020\0\0\0\001\0", 14) || memcmp(head + 36, "data", 4)) { fprintf(stderr, "need a canonical PCM WAVE file\n"); return 2; }
	channels = head[22]; rate = head[24] | head[25] << 8 | head[26] << 16 | (unsigned)head[27] << 24; bps = head[34];
	if(bps != 16) { fprintf(stderr, "need 16-bit samples\n"); return 2; }
	total_samples = (head[40] | head[41] << 8 | head[42] << 16 | (unsigned long long)head[43] << 24) / (channels * 2u);
	if(!(enc = FLAC__stream_encoder_new())) return 1;
	ok &= FLAC__stream_encoder_set_verify(enc, 1);
	ok &= FLAC__stream_encoder_set_compression_level(enc, level);
	ok &= FLAC__stream_encoder_set_channels(enc, channels);
	ok &= FLAC__stream_encoder_set_bits_per_sample(enc, bps);
	ok &= FLAC__stream_encoder_set_sample_rate(enc, rate);
	ok &= FLAC__stream_encoder_set_total_samples_estimate(enc, total_samples);
	if(!ok) { fprintf(stderr, "a setter refused its value\n"); return 1; }
	st = FLAC__stream_encoder_init_file(enc, argv[2], progress, 0);
	if(st != FLAC__STREAM_ENCODER_INIT_STATUS_OK) { fprintf(stderr, "init: %s\n", FLAC__StreamEncoderInitStatusString[st]); return 1; }
	raw = (unsigned char*)malloc(READSIZE * channels * 2u); pcm = (FLAC__int32*)malloc(sizeof(FLAC__int32) * READSIZE * channels);
	for(left = (size_t)total_samples; ok && left; ) {
		const size_t need = left > READSIZE ? READSIZE : left;
		if(fread(raw, channels * 2u, need, in) != need) { fprintf(stderr, "short read\n"); ok = 0; break; }
		for(i = 0; i < need * channels; i++) pcm[i] = (FLAC__int32)(short)(raw[2 * i] | raw[2 * i + 1] << 8);
		ok = FLAC__stream_encoder_process_interleaved(enc, pcm, (unsigned)need);
		left -= need;
	}
	ok &= FLAC__stream_encoder_finish(enc);
	printf("%s: %s, %llu samples, progress callbacks %s\n", argv[2], ok ? "ok" : "FAILED", total_samples, progress_calls ? "seen" : "none");
	if(!ok) fprintf(stderr, "state: %s\n", FLAC__StreamEncoderStateString[FLAC__stream_encoder_get_state(enc)]);
	FLAC__stream_encoder_delete(enc);
	free(raw); free(pcm); fclose(in);
	return ok ? 0 : 1;
}
