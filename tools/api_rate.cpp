// Rates through the boundary a libFLAC client sees, PCM in ordinary (pageable) host memory, frames delivered
// to a C write callback: the stream encoder object with and without MD5, the engine-level pipe, the host MD5
// alone and the pinned-memory upload rate that bounds them.  Prints one JSON object.
//   api_rate <base.raw: int16 stereo interleaved> <repeats> [level] [batch_frames] [ndevices]
#include "../include/flac_amd_stream.h"
#include "../include/flac_amd_engine.h"
extern "C" double tool_md5_rate_pcm16(const int16_t *interleaved, size_t nvalues, uint8_t digest[16]);   // md5_rate.c: seconds for one pass
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
struct Sink { uint64_t bytes, frames; std::vector<uint8_t> *keep; uint64_t pos; };
static FLAC__StreamEncoderWriteStatus on_write(const FLAC__StreamEncoder *, const FLAC__byte buf[], size_t bytes, unsigned samples, unsigned, void *cd)
{
	Sink *s = (Sink*)cd; s->bytes += bytes; if(samples) s->frames++;
	if(s->keep) { if(s->keep->size() < s->pos + bytes) s->keep->resize(s->pos + bytes); memcpy(s->keep->data() + s->pos, buf, bytes); s->pos += bytes; }
	return FLAC__STREAM_ENCODER_WRITE_STATUS_OK;
}
static FLAC__StreamEncoderSeekStatus on_seek(const FLAC__StreamEncoder *, FLAC__uint64 off, void *cd) { ((Sink*)cd)->pos = off; return FLAC__STREAM_ENCODER_SEEK_STATUS_OK; }
static FLAC__StreamEncoderTellStatus on_tell(const FLAC__StreamEncoder *, FLAC__uint64 *off, void *cd) { *off = ((Sink*)cd)->pos; return FLAC__STREAM_ENCODER_TELL_STATUS_OK; }

// the decoder object fed from memory, frames handed to a callback that looks at one sample of each
struct Source { const uint8_t *p; size_t n, at; uint64_t samples; int32_t sum; };
static FLAC__StreamDecoderReadStatus on_read(const FLAC__StreamDecoder *, FLAC__byte buf[], size_t *bytes, void *cd)
{
	Source *s = (Source*)cd; size_t k = s->n - s->at < *bytes ? s->n - s->at : *bytes;
	if(k == 0) { *bytes = 0; return FLAC__STREAM_DECODER_READ_STATUS_END_OF_STREAM; }
	memcpy(buf, s->p + s->at, k); s->at += k; *bytes = k;
	return FLAC__STREAM_DECODER_READ_STATUS_CONTINUE;
}
static FLAC__StreamDecoderWriteStatus on_frame(const FLAC__StreamDecoder *, const FLAC__Frame *fr, const FLAC__int32 *const buf[], void *cd)
{
	Source *s = (Source*)cd; s->samples += fr->header.blocksize; s->sum += buf[0][0] + buf[1][fr->header.blocksize - 1];
	return FLAC__STREAM_DECODER_WRITE_STATUS_CONTINUE;
}
static void on_error(const FLAC__StreamDecoder *, FLAC__StreamDecoderErrorStatus st, void *) { fprintf(stderr, "decoder error %d\n", (int)st); }
static double run_decode(const std::vector<uint8_t> &stream, int md5, uint64_t expect, int residuals = 1)
{
	Source src = { stream.data(), stream.size(), 0, 0, 0 };
	FLAC__StreamDecoder *d = FLAC__stream_decoder_new();
	FLAC__stream_decoder_set_md5_checking(d, md5);
	flacamd_stream_decoder_set_residuals(d, residuals);   // on (the default): FLAC__Frame carries every subframe's residuals, as the reference's does
	if(getenv("API_RATE_DEC_BATCH")) flacamd_stream_decoder_set_batch_frames(d, (unsigned)atoi(getenv("API_RATE_DEC_BATCH")));
	if(FLAC__stream_decoder_init_stream(d, on_read, 0, 0, 0, 0, on_frame, 0, on_error, &src) != FLAC__STREAM_DECODER_INIT_STATUS_OK) { fprintf(stderr, "decoder init failed\n"); exit(2); }
	const double t0 = now();
	if(!FLAC__stream_decoder_process_until_end_of_stream(d)) { fprintf(stderr, "decode failed\n"); exit(2); }
	const FLAC__bool ok = FLAC__stream_decoder_finish(d);
	const double dt = now() - t0;
	FLAC__stream_decoder_delete(d);
	if(src.samples != expect || (md5 && !ok)) { fprintf(stderr, "decode: %llu samples (expected %llu), md5 verdict %d\n", (unsigned long long)src.samples, (unsigned long long)expect, (int)ok); exit(2); }
	return dt;
}

static double run_stream(const int32_t *pcm, uint64_t n, unsigned level, unsigned batch, int md5, unsigned chunk, const int *dev, unsigned ndev, uint64_t *bytes_out, std::vector<uint8_t> *keep = 0)
{
	Sink sink = { 0, 0, keep, 0 };
	FLAC__StreamEncoder *e = FLAC__stream_encoder_new();
	FLAC__stream_encoder_set_compression_level(e, level);
	FLAC__stream_encoder_set_do_md5(e, md5);
	flacamd_stream_encoder_set_batch_frames(e, batch);
	flacamd_stream_encoder_set_devices(e, ndev, dev);
	if(FLAC__stream_encoder_init_stream(e, on_write, keep ? on_seek : 0, keep ? on_tell : 0, 0, &sink) != FLAC__STREAM_ENCODER_INIT_STATUS_OK) { fprintf(stderr, "init failed\n"); exit(2); }
	const double t0 = now();
	for(uint64_t i = 0; i < n; i += chunk) {
		const unsigned k = (unsigned)(n - i < chunk ? n - i : chunk);
		if(!FLAC__stream_encoder_process_interleaved(e, pcm + i * 2, k)) { fprintf(stderr, "process failed\n"); exit(2); }
	}
	const double t1 = now();
	if(!FLAC__stream_encoder_finish(e)) { fprintf(stderr, "finish failed\n"); exit(2); }
	const double dt = now() - t0;
	if(getenv("API_RATE_VERBOSE")) fprintf(stderr, "process %.2f ms, finish %.2f ms (md5=%d chunk=%u)\n", (t1 - t0) * 1e3, (now() - t1) * 1e3, md5, chunk);
	FLAC__stream_encoder_delete(e);
	if(bytes_out) *bytes_out = sink.bytes;
	return dt;
}

int main(int argc, char **argv)
{
	if(argc < 3) { fprintf(stderr, "usage: api_rate base.raw repeats [level] [batch_frames] [ndevices]\n"); return 2; }
	const unsigned repeats = (unsigned)atoi(argv[2]), level = argc > 3 ? (unsigned)atoi(argv[3]) : 8u, batch = argc > 4 ? (unsigned)atoi(argv[4]) : 2048u;
	unsigned ndev = argc > 5 ? (unsigned)atoi(argv[5]) : 1u;
	FILE *f = fopen(argv[1], "rb"); if(!f) { perror(argv[1]); return 2; }
	fseek(f, 0, SEEK_END); const long fb = ftell(f); fseek(f, 0, SEEK_SET);
	std::vector<int16_t> base((size_t)fb / 2);
	if(fread(base.data(), 1, (size_t)fb, f) != (size_t)fb) return 2;
	fclose(f);
	const uint64_t nb = base.size() / 2, n = nb * repeats;
	std::vector<int32_t> pcm32((size_t)n * 2);
	for(unsigned r = 0; r < repeats; r++) for(size_t i = 0; i < base.size(); i++) pcm32[(size_t)r * base.size() + i] = base[i];
	int dev[64]; int have = flacamd_device_count(); if(have < 1) { fprintf(stderr, "no device\n"); return 3; }
	if(ndev > 64) ndev = 64;
	for(unsigned i = 0; i < ndev; i++) dev[i] = (int)(i % (unsigned)have);

	// upload rate from pinned memory: what bounds any host-fed path
	double h2d = 0, d2h = 0;
	{
		const size_t bytes = 256u << 20; void *pin = 0, *d = 0;
		hipHostMalloc(&pin, bytes, hipHostMallocDefault); hipMalloc(&d, bytes); memset(pin, 1, bytes);
		hipMemcpy(d, pin, bytes, hipMemcpyHostToDevice);
		double t0 = now(); for(int i = 0; i < 4; i++) hipMemcpy(d, pin, bytes, hipMemcpyHostToDevice); h2d = 4.0 * bytes / (now() - t0) / 1e9;
		t0 = now(); for(int i = 0; i < 4; i++) hipMemcpy(pin, d, bytes, hipMemcpyDeviceToHost); d2h = 4.0 * bytes / (now() - t0) / 1e9;
		hipFree(d); hipHostFree(pin);
	}
	// the host MD5 alone, as the object feeds it (16-bit samples as bytes); best of 3 passes
	double md5_alone;
	{
		std::vector<int16_t> p16((size_t)n * 2); flacamd_pcm_narrow(p16.data(), pcm32.data(), (size_t)n * 2);
		uint8_t dg[16]; double best = 1e30;
		for(int rep = 0; rep < 3; rep++) { const double dt = tool_md5_rate_pcm16(p16.data(), (size_t)n * 2, dg); if(dt < best) best = dt; }
		md5_alone = n / best / 1e6;
	}
	uint64_t bytes = 0;
	run_stream(pcm32.data(), nb, level, batch, 0, 1u << 20, dev, ndev, 0);   // warm-up: module load, first allocations
	double best_off = 1e30, best_on = 1e30, best_small = 1e30;
	for(int rep = 0; rep < 3; rep++) { const double dt = run_stream(pcm32.data(), n, level, batch, 0, 1u << 20, dev, ndev, &bytes); if(dt < best_off) best_off = dt; }
	for(int rep = 0; rep < 2; rep++) { const double dt = run_stream(pcm32.data(), n, level, batch, 1, 1u << 20, dev, ndev, 0); if(dt < best_on) best_on = dt; }
	for(int rep = 0; rep < 2; rep++) { const double dt = run_stream(pcm32.data(), n, level, batch, 0, 4096, dev, ndev, 0); if(dt < best_small) best_small = dt; }
	// the same stream (with its MD5 in place) back through the decoder object
	double dec_off = 1e30, dec_on = 1e30, dec_lean = 1e30;
	{
		std::vector<uint8_t> stream;
		run_stream(pcm32.data(), n, level, batch, 1, 1u << 20, dev, ndev, 0, &stream);
		run_decode(stream, 0, n);
		for(int rep = 0; rep < 3; rep++) { const double dt = run_decode(stream, 0, n); if(dt < dec_off) dec_off = dt; }
		for(int rep = 0; rep < 2; rep++) { const double dt = run_decode(stream, 1, n); if(dt < dec_on) dec_on = dt; }
		for(int rep = 0; rep < 3; rep++) { const double dt = run_decode(stream, 0, n, 0); if(dt < dec_lean) dec_lean = dt; }
	}
	// the engine-level pipe on the same buffer
	double pipe32 = 0;
	{
		flacamd_params p; memset(&p, 0, sizeof p); p.channels = 2; p.bits_per_sample = 16; p.sample_rate = 44100; flacamd_params_preset(&p, level);
		flacamd_pipe *pp = 0;
		if(flacamd_pipe_create(&pp, dev, ndev, &p, batch, ndev > 1 ? 2u : 3u) != 0) { fprintf(stderr, "pipe create failed\n"); return 2; }
		const uint64_t F = (n + 4095) / 4096; const size_t cap = (size_t)((F + batch - 1) / batch) * flacamd_pipe_max_batch_bytes(pp);
		std::vector<uint8_t> out(cap); std::vector<uint32_t> fbv((size_t)F); uint64_t total = 0; uint32_t nf = 0;
		memset(out.data(), 0, cap);
		double best = 1e30;
		for(int rep = 0; rep < 3; rep++) {
			const double t0 = now();
			const int rc = flacamd_pipe_encode_host(pp, pcm32.data(), FLACAMD_PCM_S32, n, 1, 0, out.data(), cap, fbv.data(), (size_t)F, &total, &nf);
			const double dt = now() - t0; if(rc != 0) { fprintf(stderr, "pipe encode failed %d\n", rc); return 2; }
			if(dt < best) best = dt;
		}
		pipe32 = n / best / 1e6;
		flacamd_pipe_destroy(pp);
		if(total + 86 + 0 < bytes - 200 || total > bytes) fprintf(stderr, "warning: pipe bytes %llu vs stream bytes %llu\n", (unsigned long long)total, (unsigned long long)bytes);
	}
	printf("{\"decoder_md5_off\": %.1f, \"decoder_md5_on\": %.1f, \"decoder_md5_off_without_residuals\": %.1f, ", n / dec_off / 1e6, n / dec_on / 1e6, n / dec_lean / 1e6);
	printf("\"stream_md5_off\": %.1f, \"stream_md5_on\": %.1f, \"stream_md5_off_4096_sample_calls\": %.1f, \"encode_host\": %.1f, \"md5_alone\": %.1f, "
		"\"pcie_h2d_GBs\": %.2f, \"pcie_d2h_GBs\": %.2f, \"h2d_bound_int32_stereo\": %.1f, \"unit\": \"MSamples/s\", \"wide_samples\": %llu, \"level\": %u, \"batch_frames\": %u, "
		"\"devices\": %u, \"stream_bytes\": %llu, \"note\": \"PCM handed over as int32 in pageable host memory in 1 Mi-sample calls (and 4096-sample calls), frames delivered to a C write callback; best of 3\"}\n",
		n / best_off / 1e6, n / best_on / 1e6, n / best_small / 1e6, pipe32, md5_alone, h2d, d2h, h2d * 1e9 / 8.0 / 1e6, (unsigned long long)n, level, batch, ndev, (unsigned long long)bytes);
	return 0;
}
