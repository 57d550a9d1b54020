/* A libFLAC client: FLAC file in, 16-bit WAVE file and an analysis listing out.  The listing has the layout of the
 * reference tool's analysis mode (src/flac/analyze.c:61-125, residuals as a digest per subframe): it reads FLAC__Frame.header,
 * .subframes[] (type, wasted bits, order, coefficients, warm-up samples, partition order, Rice parameters) and the
 * decode position.  Compiled against include/flac_amd_stream.h + libflacamd.so and against the reference's headers
 * and library; both builds must write the same two files (tests/test_gpu_c_client.py).  Call sequence of the
 * reference's example client (examples/c/decode/file/main.c).
 *   -DFLAC_DECODER_HEADER='"flac_amd_stream.h"'  or  '"FLAC/stream_decoder.h"' */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include FLAC_DECODER_HEADER

static FILE *wav, *ana;
static FLAC__uint64 total_samples, last_pos;
static unsigned rate, channels, bps, frame_no;

static void put16(FILE *f, unsigned v) { fputc(v & 0xff, f); fputc((v >> 8) & 0xff, f); }
static void put32(FILE *f, unsigned long v) { put16(f, (unsigned)(v & 0xffff)); put16(f, (unsigned)(v >> 16)); }

static void list_rice(const FLAC__EntropyCodingMethod *m)
{
	const unsigned esc = m->type == FLAC__ENTROPY_CODING_METHOD_PARTITIONED_RICE2 ? 31u : 15u;
	unsigned i;
	for(i = 0; i < (1u << m->data.partitioned_rice.order); i++) {
		const unsigned p = m->data.partitioned_rice.contents->parameters[i];
		if(p == esc) fprintf(ana, "\t\tparameter[%u]=ESCAPE, raw_bits=%u\n", i, m->data.partitioned_rice.contents->raw_bits[i]);
		else fprintf(ana, "\t\tparameter[%u]=%u\n", i, p);
	}
}

/* a digest of the values a subframe carried (every residual, or every sample of a verbatim subframe), so that two
 * decoders can be compared on them without a listing of millions of lines */
static unsigned long digest(const FLAC__int32 *v, unsigned n)
{
	unsigned long h = 2166136261ul; unsigned i;
	for(i = 0; i < n; i++) { h ^= (unsigned long)(FLAC__uint32)v[i]; h = (h * 16777619ul) & 0xfffffffful; }
	return h;
}

static FLAC__StreamDecoderWriteStatus on_frame(const FLAC__StreamDecoder *dec, const FLAC__Frame *frame, const FLAC__int32 * const buffer[], void *client_data)
{
	unsigned c, i; FLAC__uint64 pos = 0;
	(void)client_data;
	if(frame->header.number.sample_number == 0) {
		const unsigned long bytes = (unsigned long)(total_samples * channels * (bps / 8));
		fwrite("RIFF", 1, 4, wav); put32(wav, bytes + 36); fwrite("WAVEfmt ", 1, 8, wav); put32(wav, 16); put16(wav, 1); put16(wav, channels);
		put32(wav, rate); put32(wav, rate * channels * (bps / 8)); put16(wav, channels * (bps / 8)); put16(wav, bps); fwrite("data", 1, 4, wav); put32(wav, bytes);
	}
	for(i = 0; i < frame->header.blocksize; i++) for(c = 0; c < channels; c++) put16(wav, (unsigned)buffer[c][i] & 0xffff);
	if(!FLAC__stream_decoder_get_decode_position(dec, &pos)) pos = 0;
	fprintf(ana, "frame=%u\toffset=%llu\tbits=%u\tblocksize=%u\tsample_rate=%u\tchannels=%u\tchannel_assignment=%s\n", frame_no++, (unsigned long long)last_pos,
		(unsigned)(pos - last_pos) * 8, frame->header.blocksize, frame->header.sample_rate, frame->header.channels, FLAC__ChannelAssignmentString[frame->header.channel_assignment]);
	last_pos = pos;
	for(c = 0; c < frame->header.channels; c++) {
		const FLAC__Subframe *s = &frame->subframes[c];
		fprintf(ana, "\tsubframe=%u\twasted_bits=%u\ttype=%s", c, s->wasted_bits, FLAC__SubframeTypeString[s->type]);
		if(s->type == FLAC__SUBFRAME_TYPE_CONSTANT) fprintf(ana, "\tvalue=%d\n", s->data.constant.value);
		else if(s->type == FLAC__SUBFRAME_TYPE_VERBATIM) fprintf(ana, "\tdata_digest=%08lx\n", digest(s->data.verbatim.data, frame->header.blocksize));
		else if(s->type == FLAC__SUBFRAME_TYPE_FIXED) {
			const int r2 = s->data.fixed.entropy_coding_method.type == FLAC__ENTROPY_CODING_METHOD_PARTITIONED_RICE2;
			fprintf(ana, "\torder=%u\tresidual_type=%s\tpartition_order=%u\n", s->data.fixed.order, r2 ? "RICE2" : "RICE", s->data.fixed.entropy_coding_method.data.partitioned_rice.order);
			for(i = 0; i < s->data.fixed.order; i++) fprintf(ana, "\t\twarmup[%u]=%d\n", i, s->data.fixed.warmup[i]);
			list_rice(&s->data.fixed.entropy_coding_method);
			fprintf(ana, "\t\tresidual_digest=%08lx\n", digest(s->data.fixed.residual, frame->header.blocksize - s->data.fixed.order));
		}
		else {
			const int r2 = s->data.lpc.entropy_coding_method.type == FLAC__ENTROPY_CODING_METHOD_PARTITIONED_RICE2;
			fprintf(ana, "\torder=%u\tqlp_coeff_precision=%u\tquantization_level=%d\tresidual_type=%s\tpartition_order=%u\n", s->data.lpc.order, s->data.lpc.qlp_coeff_precision,
				s->data.lpc.quantization_level, r2 ? "RICE2" : "RICE", s->data.lpc.entropy_coding_method.data.partitioned_rice.order);
			for(i = 0; i < s->data.lpc.order; i++) fprintf(ana, "\t\tqlp_coeff[%u]=%d\n", i, s->data.lpc.qlp_coeff[i]);
			for(i = 0; i < s->data.lpc.order; i++) fprintf(ana, "\t\twarmup[%u]=%d\n", i, s->data.lpc.warmup[i]);
			list_rice(&s->data.lpc.entropy_coding_method);
			fprintf(ana, "\t\tresidual_digest=%08lx\n", digest(s->data.lpc.residual, frame->header.blocksize - s->data.lpc.order));
		}
	}
	fprintf(ana, "\tfooter_crc=%04x\theader_crc=%02x\n", (unsigned)frame->footer.crc, (unsigned)frame->header.crc);
	return FLAC__STREAM_DECODER_WRITE_STATUS_CONTINUE;
}

static void on_metadata(const FLAC__StreamDecoder *dec, const FLAC__StreamMetadata *m, void *client_data)
{
	(void)dec; (void)client_data;
	if(m->type == FLAC__METADATA_TYPE_STREAMINFO) {
		total_samples = m->data.stream_info.total_samples; rate = m->data.stream_info.sample_rate; channels = m->data.stream_info.channels; bps = m->data.stream_info.bits_per_sample;
	}
}

static void on_error(const FLAC__StreamDecoder *dec, FLAC__StreamDecoderErrorStatus status, void *client_data)
{
	(void)dec; (void)client_data;
	fprintf(ana, "error=%s\n", FLAC__StreamDecoderErrorStatusString[status]);
}

int main(int argc, char **argv)
{
	FLAC__StreamDecoder *dec; FLAC__StreamDecoderInitStatus st; FLAC__bool ok;
	if(argc != 4) { fprintf(stderr, "usage: %s in.flac out.wav out.ana\n", argv[0]); return 2; }
	if(!(wav = fopen(argv[2], "wb")) || !(ana = fopen(argv[3], "w"))) { perror("output"); return 2; }
	if(!(dec = FLAC__stream_decoder_new())) return 1;
	(void)FLAC__stream_decoder_set_md5_checking(dec, 1);
	st = FLAC__stream_decoder_init_file(dec, argv[1], on_frame, on_metadata, on_error, 0);
	if(st != FLAC__STREAM_DECODER_INIT_STATUS_OK) { fprintf(stderr, "init: %s\n", FLAC__StreamDecoderInitStatusString[st]); return 1; }
	ok = FLAC__stream_decoder_process_until_end_of_metadata(dec);
	if(ok && !FLAC__stream_decoder_get_decode_position(dec, &last_pos)) last_pos = 0;
	ok = ok && FLAC__stream_decoder_process_until_end_of_stream(dec);
	fprintf(ana, "state=%s\n", FLAC__StreamDecoderStateString[FLAC__stream_decoder_get_state(dec)]);
	ok = FLAC__stream_decoder_finish(dec) && ok;   /* false when the MD5 does not match */
	fprintf(ana, "result=%s\n", ok ? "ok" : "FAILED");
	FLAC__stream_decoder_delete(dec);
	fclose(wav); fclose(ana);
	return ok ? 0 : 1;
}
