// Host side of the C ABI declared in include/flac_amd_engine.h: parameter resolution,
// window tables, device workspace, and the launch sequence of the encoder stages.
// No CPU implementation of any stage lives here: without a HIP device every compute entry
// point fails with FLACAMD_E_NO_DEVICE.
#include "engine_types.h"
#include "residual.h"
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <unordered_map>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

extern "C" {
void fa_launch_prep(const void *pcm, const FaCfg *c, const FaWork *w, hipStream_t st, hipStream_t st_short);
int fa_prep_fast_ok(const void *pcm, const FaCfg *c);
int fa_search_fast_ok(const void *pcm, const FaCfg *c);
void fa_launch_autoc(const void *pcm, const FaCfg *c, const FaWork *w, hipStream_t st);
void fa_launch_model(const FaCfg *c, const FaWork *w, hipStream_t st);
size_t fa_search_lds_bytes(const FaCfg *c, uint32_t cg);
void fa_launch_search(const void *pcm, const FaCfg *c, const FaWork *w, uint32_t cg, int sum64, hipStream_t st);
void fa_launch_search2(const void *pcm, const FaCfg *c, const FaWork *w, uint32_t cg, int sum64, hipStream_t st, hipStream_t st_rest);
uint32_t fa_search_rest_frames(const void *pcm, const FaCfg *c);
size_t fa_pack_lds_bytes(const FaCfg *c, uint32_t *bitbuf_words_out);
void fa_launch_pack(const void *pcm, const FaCfg *c, const FaWork *w, hipStream_t st);
void fa_launch_pack_general(const void *pcm, const FaCfg *c, const FaWork *w, hipStream_t st, int large);
void fa_launch_gather(const FaCfg *c, const FaWork *w, void *out, size_t out_cap, uint32_t *frame_bytes_out, hipStream_t st);
void fa_launch_offsets(const FaCfg *c, const FaWork *w, hipStream_t st);
int fa_pack_direct_ok(const void *pcm, const FaCfg *c);
}

static thread_local char g_last_error[256] = "";

static int hip_fail(hipError_t e, const char *what)
{
	snprintf(g_last_error, sizeof g_last_error, "%s: %s", what, hipGetErrorString(e));
	return e == hipErrorOutOfMemory ? FLACAMD_E_OUT_OF_MEMORY : FLACAMD_E_DEVICE;
}
#define HIP_TRY(call) do { hipError_t e_ = (call); if(e_ != hipSuccess) return hip_fail(e_, #call); } while(0)

extern "C" const char *flacamd_last_error(void) { return g_last_error; }
// for the other translation units of the library (the decoder): one message buffer per thread, one getter
extern "C" void fa_set_last_error(const char *msg) { snprintf(g_last_error, sizeof g_last_error, "%s", msg ? msg : ""); }

extern "C" const char *flacamd_strerror(int code)
{
	switch(code) {
	case FLACAMD_OK: return "ok";
	case FLACAMD_E_INVALID_ARGUMENT: return "invalid argument";
	case FLACAMD_E_UNSUPPORTED: return "outside the engine limits";
	case FLACAMD_E_NO_DEVICE: return "no HIP device";
	case FLACAMD_E_OUT_OF_MEMORY: return "out of device memory";
	case FLACAMD_E_DEVICE: return "HIP call failed";
	case FLACAMD_E_CAPACITY: return "capacity too small";
	case FLACAMD_E_FRAME_OVERFLOW: return "frame exceeded its slot";
	case FLACAMD_E_BAD_STREAM: return "bad stream";
	default: return "unknown error";
	}
}

extern "C" int flacamd_device_count(void)
{
	int n = 0;
	if(hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

// ------------------------------------------------------------------ parameters

static unsigned ilog2u(uint32_t v) { unsigned l = 0; while(v >>= 1) l++; return l; }

extern "C" void flacamd_params_preset(flacamd_params *p, unsigned level)
{
	// compression_levels_[] (stream_encoder.c:127-148): mid/side, loose, max lpc order, exhaustive, max partition order
	static const uint8_t t[9][5] = { {0,0,0,0,3},{1,1,0,0,3},{1,0,0,0,3},{0,0,6,0,4},{1,1,8,0,4},{1,0,8,0,5},{1,0,8,0,6},{1,0,8,1,6},{1,0,12,1,6} };
	if(level > 8) level = 8;
	p->do_mid_side = t[level][0]; p->loose_mid_side = t[level][1]; p->max_lpc_order = t[level][2];
	p->qlp_coeff_precision = 0; p->do_qlp_coeff_prec_search = 0; p->do_exhaustive_model_search = t[level][3];
	p->min_residual_partition_order = 0; p->max_residual_partition_order = t[level][4];
	p->num_apodizations = 1; p->apod_type[0] = FLACAMD_WIN_TUKEY; p->apod_param[0] = 0.5f; // stream_encoder.c:1481-1483
}

extern "C" int flacamd_params_resolve(flacamd_params *p)
{
	if(!p) return FLACAMD_E_INVALID_ARGUMENT;
	if(p->channels == 0 || p->channels > FLACAMD_MAX_CHANNELS) return FLACAMD_E_INVALID_ARGUMENT;
	if(p->channels != 2) { p->do_mid_side = 0; p->loose_mid_side = 0; }
	else if(!p->do_mid_side) p->loose_mid_side = 0;
	if(p->bits_per_sample < 4 || p->bits_per_sample > 24) return FLACAMD_E_INVALID_ARGUMENT;
	if(p->sample_rate == 0 || p->sample_rate > 655350) return FLACAMD_E_INVALID_ARGUMENT;
	if(p->blocksize == 0) p->blocksize = p->max_lpc_order == 0 ? 1152 : 4096;
	if(p->blocksize < 16 || p->blocksize > 65535) return FLACAMD_E_INVALID_ARGUMENT;
	if(p->max_lpc_order > FLACAMD_MAX_LPC_ORDER) return FLACAMD_E_INVALID_ARGUMENT;
	if(p->blocksize < p->max_lpc_order) return FLACAMD_E_INVALID_ARGUMENT;
	if(p->qlp_coeff_precision == 0) {
		const unsigned bs = p->blocksize;
		if(p->bits_per_sample < 16) { const unsigned v = 2 + p->bits_per_sample / 2; p->qlp_coeff_precision = v > FA_MIN_QLP_PRECISION ? v : FA_MIN_QLP_PRECISION; }
		else if(p->bits_per_sample == 16) p->qlp_coeff_precision = bs <= 192 ? 7 : bs <= 384 ? 8 : bs <= 576 ? 9 : bs <= 1152 ? 10 : bs <= 2304 ? 11 : bs <= 4608 ? 12 : 13;
		else p->qlp_coeff_precision = bs <= 384 ? FA_MAX_QLP_PRECISION - 2 : bs <= 1152 ? FA_MAX_QLP_PRECISION - 1 : FA_MAX_QLP_PRECISION;
	}
	else if(p->qlp_coeff_precision < FA_MIN_QLP_PRECISION || p->qlp_coeff_precision > FA_MAX_QLP_PRECISION) return FLACAMD_E_INVALID_ARGUMENT;
	if(p->max_residual_partition_order > 15) p->max_residual_partition_order = 15;
	if(p->min_residual_partition_order >= p->max_residual_partition_order) p->min_residual_partition_order = p->max_residual_partition_order;
	if(p->num_apodizations == 0) { p->num_apodizations = 1; p->apod_type[0] = FLACAMD_WIN_TUKEY; p->apod_param[0] = 0.5f; }
	if(p->num_apodizations > FLACAMD_MAX_APODIZATIONS) return FLACAMD_E_INVALID_ARGUMENT;
	// engine limits
	if(p->blocksize > FLACAMD_MAX_BLOCKSIZE) return FLACAMD_E_UNSUPPORTED;
	if(p->max_residual_partition_order > FLACAMD_MAX_PARTITION_ORDER) return FLACAMD_E_UNSUPPORTED;
	return FLACAMD_OK;
}

// window.c:49-223.  The float/double mix of every sub-expression is the reference's: the
// value stored as float depends on it.
extern "C" void flacamd_window(uint32_t type, float param, int32_t L, float *w)
{
	const int32_t N = L - 1; int32_t n;
	const double N2 = (double)N / 2.;
	switch(type) {
	case FLACAMD_WIN_BARTLETT:
		if(L & 1) { for(n = 0; n <= N/2; n++) w[n] = 2.0f * n / (float)N; for(; n <= N; n++) w[n] = 2.0f - 2.0f * n / (float)N; }
		else { for(n = 0; n <= L/2-1; n++) w[n] = 2.0f * n / (float)N; for(; n <= N; n++) w[n] = 2.0f - 2.0f * (N-n) / (float)N; }
		break;
	case FLACAMD_WIN_BARTLETT_HANN:
		for(n = 0; n < L; n++) w[n] = (float)(0.62f - 0.48f * fabs((double)((float)n/(float)N+0.5f))  /* C's fabs() takes and returns double (window.c:74) */ + 0.38f * cos(2.0f * M_PI * ((float)n/(float)N+0.5f)));
		break;
	case FLACAMD_WIN_BLACKMAN:
		for(n = 0; n < L; n++) w[n] = (float)(0.42f - 0.5f * cos(2.0f * M_PI * n / N) + 0.08f * cos(4.0f * M_PI * n / N));
		break;
	case FLACAMD_WIN_BLACKMAN_HARRIS_4TERM:
		for(n = 0; n <= N; n++) w[n] = (float)(0.35875f - 0.48829f * cos(2.0f * M_PI * n / N) + 0.14128f * cos(4.0f * M_PI * n / N) - 0.01168f * cos(6.0f * M_PI * n / N));
		break;
	case FLACAMD_WIN_CONNES:
		for(n = 0; n <= N; n++) { double k = ((double)n - N2) / N2; k = 1.0f - k * k; w[n] = (float)(k * k); }
		break;
	case FLACAMD_WIN_FLATTOP:
		for(n = 0; n < L; n++) w[n] = (float)(1.0f - 1.93f * cos(2.0f * M_PI * n / N) + 1.29f * cos(4.0f * M_PI * n / N) - 0.388f * cos(6.0f * M_PI * n / N) + 0.0322f * cos(8.0f * M_PI * n / N));
		break;
	case FLACAMD_WIN_GAUSS:
		for(n = 0; n <= N; n++) { const double k = ((double)n - N2) / (param * N2); w[n] = (float)exp(-0.5f * k * k); }
		break;
	case FLACAMD_WIN_HAMMING:
		for(n = 0; n < L; n++) w[n] = (float)(0.54f - 0.46f * cos(2.0f * M_PI * n / N));
		break;
	case FLACAMD_WIN_HANN:
		for(n = 0; n < L; n++) w[n] = (float)(0.5f - 0.5f * cos(2.0f * M_PI * n / N));
		break;
	case FLACAMD_WIN_KAISER_BESSEL:
		for(n = 0; n < L; n++) w[n] = (float)(0.402f - 0.498f * cos(2.0f * M_PI * n / N) + 0.098f * cos(4.0f * M_PI * n / N) - 0.001f * cos(6.0f * M_PI * n / N));
		break;
	case FLACAMD_WIN_NUTTALL:
		for(n = 0; n < L; n++) w[n] = (float)(0.3635819f - 0.4891775f*cos(2.0f*M_PI*n/N) + 0.1365995f*cos(4.0f*M_PI*n/N) - 0.0106411f*cos(6.0f*M_PI*n/N));
		break;
	case FLACAMD_WIN_RECTANGLE:
		for(n = 0; n < L; n++) w[n] = 1.0f;
		break;
	case FLACAMD_WIN_TRIANGLE: // window.c:179 reads "L+1/2", i.e. L: the odd-length branch never runs its second loop
		if(L & 1) { for(n = 1; n <= L+1/2; n++) w[n-1] = 2.0f * n / ((float)L + 1.0f); for(; n <= L; n++) w[n-1] = - (float)(2 * (L - n + 1)) / ((float)L + 1.0f); }
		else { for(n = 1; n <= L/2; n++) w[n-1] = 2.0f * n / (float)L; for(; n <= L; n++) w[n-1] = ((float)(2 * (L - n)) + 1.0f) / (float)L; }
		break;
	case FLACAMD_WIN_WELCH:
		for(n = 0; n <= N; n++) { const double k = ((double)n - N2) / N2; w[n] = (float)(1.0f - k * k); }
		break;
	case FLACAMD_WIN_TUKEY: default:
		if(param <= 0.0) flacamd_window(FLACAMD_WIN_RECTANGLE, 0, L, w);
		else if(param >= 1.0) flacamd_window(FLACAMD_WIN_HANN, 0, L, w);
		else {
			const int32_t Np = (int32_t)(param / 2.0f * L) - 1;
			flacamd_window(FLACAMD_WIN_RECTANGLE, 0, L, w);
			if(Np > 0) for(n = 0; n <= Np; n++) {
				w[n] = (float)(0.5f - 0.5f * cos(M_PI * n / Np));
				w[L-Np-1+n] = (float)(0.5f - 0.5f * cos(M_PI * (n+Np) / Np));
			}
		}
		break;
	}
}

// ------------------------------------------------------------------ encoder object

struct flacamd_encoder {
	int device;
	flacamd_params p;
	FaCfg cfg;             // static fields filled at create, batch fields per call
	FaWork work;
	uint32_t max_frames;
	uint32_t cand_group;   // candidates evaluated per LDS round in the search kernel
	int sum64;             // 64-bit |residual| partial sums needed
	void *arena; size_t arena_bytes;
	void *d_pcm_stage; size_t pcm_stage_bytes;   // host-entry staging
	uint8_t *d_out_stage; size_t out_stage_bytes;
	uint32_t *d_fb_stage;
	uint32_t last_F, last_G;
	int timing;
	hipEvent_t ev[7];
	hipEvent_t *cur_ev;                      // the stage events the batch being queued records: ev, or its row of qev
	hipEvent_t qev[FLACAMD_QUEUE_DEPTH][7];  // made when the first batch is queued with timing on
	int qev_made;
	hipStream_t side; hipEvent_t ev_fork, ev_join;   // the autocorrelation stage runs beside the prep stage
	float last_ms[6];
	int have_timing;
	// per-call tables of the multi-stream entry points: pinned host staging and a device buffer, both grow-only
	void *h_tab; size_t h_tab_bytes; void *d_tab; size_t d_tab_bytes;
	void *d_pcm16; size_t pcm16_bytes;   // 16-bit copy of a batch handed over as 32-bit samples (streams of up to 16 bits)
	uint64_t *q_rec; uint32_t q_n; hipStream_t q_stream;   // queued batches: page-locked (status, total) pairs, how many are outstanding, their stream
	hipStream_t md5s; hipEvent_t ev_tab, ev_md5;   // the clip batch hashes on a stream of its own beside the coding stages
	const uint8_t *given_digests; void *given_ready;   // flacamd_encoder_use_digests: for the next clip batch only
};

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

extern "C" const flacamd_params *flacamd_encoder_params(const flacamd_encoder *e) { return e ? &e->p : 0; }

extern "C" size_t flacamd_encoder_max_output_bytes(const flacamd_encoder *e, uint32_t nframes)
{
	return e ? (size_t)nframes * e->cfg.slot_bytes : 0;
}

static int encoder_create_once(flacamd_encoder **out, int device, const flacamd_params *params, uint32_t max_frames);
extern "C" void flacamd_pipe_trim(void);

// Idle pipes that were parked for reuse hold device arenas and pinned slots; before an allocation failure is
// reported they are given back and the creation is tried once more.
extern "C" int flacamd_encoder_create(flacamd_encoder **out, int device, const flacamd_params *params, uint32_t max_frames)
{
	int rc = encoder_create_once(out, device, params, max_frames);
	if(rc == FLACAMD_E_OUT_OF_MEMORY) { (void)hipGetLastError(); flacamd_pipe_trim(); rc = encoder_create_once(out, device, params, max_frames); }
	return rc;
}

static int encoder_create_once(flacamd_encoder **out, int device, const flacamd_params *params, uint32_t max_frames)
{
	if(!out || !params || max_frames == 0) return FLACAMD_E_INVALID_ARGUMENT;
	*out = 0;
	if(flacamd_device_count() <= device || device < 0) { snprintf(g_last_error, sizeof g_last_error, "no HIP device %d", device); return FLACAMD_E_NO_DEVICE; }
	flacamd_encoder *e = (flacamd_encoder*)calloc(1, sizeof *e);
	if(!e) return FLACAMD_E_OUT_OF_MEMORY;
	e->p = *params;
	int rc = flacamd_params_resolve(&e->p);
	if(rc != FLACAMD_OK) { free(e); return rc; }
	e->device = device; e->max_frames = max_frames;
	{ hipError_t e0 = hipSetDevice(device); if(e0 != hipSuccess) { free(e); return hip_fail(e0, "hipSetDevice"); } }

	const flacamd_params &p = e->p;
	FaCfg &c = e->cfg;
	memset(&c, 0, sizeof c);
	c.channels = p.channels; c.bps = p.bits_per_sample; c.rate = p.sample_rate; c.N = p.blocksize;
	c.midside = (p.channels == 2 && p.do_mid_side) ? 1u : 0u;
	c.nsig = c.midside ? 4u : p.channels;
	c.loose = p.loose_mid_side;
	{ // stream_encoder.c:878-890
		unsigned lp = (unsigned)((double)p.sample_rate * 0.4 / (double)p.blocksize + 0.5);
		c.loose_period = lp ? lp : 1u;
	}
	c.max_lpc_order = p.max_lpc_order; c.qlp_precision = p.qlp_coeff_precision; c.prec_search = p.do_qlp_coeff_prec_search;
	c.exhaustive = p.do_exhaustive_model_search; c.min_po = p.min_residual_partition_order; c.max_po = p.max_residual_partition_order;
	c.disable_constant = p.disable_constant; c.disable_fixed = p.disable_fixed; c.disable_verbatim = p.disable_verbatim;
	c.num_apod = p.num_apodizations; c.lag = p.max_lpc_order + 1u;
	c.k_limit = p.bits_per_sample > 16 ? 31u : 15u;
	c.wide_fixed = (p.bits_per_sample + ilog2u(p.blocksize) + 1 > 30) ? 1u : 0u; // stream_encoder.c:894
	{
		const uint32_t nfixed = p.do_exhaustive_model_search ? 5u : 1u;
		const uint32_t norders = p.max_lpc_order ? (p.do_exhaustive_model_search ? p.max_lpc_order : 1u) : 0u;
		const uint32_t nprec = p.do_qlp_coeff_prec_search ? (FA_MAX_QLP_PRECISION - FA_MIN_QLP_PRECISION + 1u) : 1u;
		c.max_cand = nfixed + p.num_apodizations * norders * nprec;
	}
	{
		// frame slot: header + per channel (verbatim bound with slack + subframe head) + footer
		const size_t maxbps = p.bits_per_sample + 1u;
		const size_t sub = ((size_t)p.blocksize * (maxbps + 1u) + 8u + 32u + 32u * (maxbps + 15u) + 64u + 7u) / 8u + 8u;
		c.slot_bytes = (uint32_t)align_up(16u + p.channels * sub + 2u + 16u, 16u);
	}
	{ const char *dbg = getenv("FLACAMD_DEBUG"); c.debug = dbg ? (uint32_t)strtoul(dbg, 0, 0) : 0u; }
	c.spill = p.blocksize > FA_LDS_BLOCKSIZE ? 1u : 0u;
	// all candidates in one round where they are few; else thirty-two at a time (eight lanes price a candidate: a full workgroup):
	// with the signal that is 38 KB of LDS, four workgroups per CU
	e->cand_group = c.max_cand <= 24u ? c.max_cand : 32u;
	if(e->cand_group == 0) e->cand_group = 1;
	e->sum64 = (ilog2u(p.blocksize) + p.bits_per_sample + 1u < 32u) ? 0 : 1;
	// LDS of the general kernels: fewer candidates per pass, then the signal out of LDS, until a workgroup's tables fit
	while(fa_search_lds_bytes(&c, e->cand_group) > 160u * 1024u && e->cand_group > 1u) e->cand_group = (e->cand_group + 1u) / 2u;
	if((fa_search_lds_bytes(&c, e->cand_group) > 160u * 1024u || fa_pack_lds_bytes(&c, 0) > 160u * 1024u) && !c.spill) {
		c.spill = 1u;
		e->cand_group = c.max_cand <= 24u ? (c.max_cand ? c.max_cand : 1u) : 32u;
		while(fa_search_lds_bytes(&c, e->cand_group) > 160u * 1024u && e->cand_group > 1u) e->cand_group = (e->cand_group + 1u) / 2u;
	}
	if(fa_search_lds_bytes(&c, e->cand_group) > 160u * 1024u) { free(e); return FLACAMD_E_UNSUPPORTED; }
	if(fa_pack_lds_bytes(&c, 0) > 160u * 1024u) { free(e); return FLACAMD_E_UNSUPPORTED; }

	// one arena for all per-batch arrays
	const size_t G = (size_t)max_frames * c.nsig, F = max_frames;
	const size_t qrows = p.max_lpc_order ? p.max_lpc_order : 1u;
	c.cand_stride = (uint32_t)((2u + qrows + 3u) & ~(size_t)3u);   // header, coefficients, their sum
	size_t off = 0;
	auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
	const size_t o_wasted = take(G), o_const = take(G), o_ferr = take(5 * G * 8), o_autoc = take((size_t)c.num_apod * c.lag * G * 4),
		o_ncand = take(G * 4), o_cand = take((size_t)c.max_cand * c.cand_stride * G * 4 + 16),
		o_win = take(G * sizeof(FaWinner)), o_slots = take(F * c.slot_bytes), o_fb = take(F * 4), o_asg = take(F * 4), o_prec = take(F * sizeof(FaPackRec)),
		o_status = take(256), o_offs = take((F + 1) * 8), o_flen = take(F * 4), o_fnum = take(F * 4), o_fshift = take(F * 4), o_window = take((size_t)c.num_apod * c.N * 4), o_crc = take(2u * (1024u + (size_t)c.slot_bytes + FA_PACK_LARGE_WORDS + 1024u)), o_dbg = take((size_t)FA_DBG_WORDS * 8u),
		o_spill = take(c.spill ? G * (size_t)xs_words_for(c.N) * 4u : 0u),
		o_short = take(F * 4u),
		o_lb = take(F * 12u),   // look-back states (u64) and the flags of late frames (u32) of the pack kernel that writes in place
		o_fps = take((p.channels == 2 && p.bits_per_sample <= 16 && p.blocksize <= 4096 && (p.blocksize % 16u) == 0) ? G * 320u * 4u : 0u);   // only the 16-bit stereo kernels use it
	e->arena_bytes = off;
	hipError_t he = hipMalloc(&e->arena, e->arena_bytes);
	if(he != hipSuccess) { rc = hip_fail(he, "hipMalloc(arena)"); free(e); return rc; }
	uint8_t *b = (uint8_t*)e->arena;
	FaWork &w = e->work;
	w.wasted = b + o_wasted; w.is_const = b + o_const; w.fixed_err = (uint64_t*)(b + o_ferr); w.autoc = (float*)(b + o_autoc);
	w.ncand = (uint32_t*)(b + o_ncand); w.cand_hdr = (uint32_t*)(b + o_cand); w.cand_q = (int32_t*)(b + o_cand) + 1;
	w.winner = (FaWinner*)(b + o_win); w.slots = b + o_slots; w.frame_bytes = (uint32_t*)(b + o_fb); w.assignment = (uint32_t*)(b + o_asg); w.packrec = (FaPackRec*)(b + o_prec);
	w.xs_spill = (int32_t*)(b + o_spill);
	w.short_list = (uint32_t*)(b + o_short);
	w.lb = (unsigned long long*)(b + o_lb); w.out = 0; w.out_cap = 0; w.frame_bytes_out = 0;
	w.fixed_psum = (p.channels == 2 && p.bits_per_sample <= 16 && p.blocksize <= 4096 && (p.blocksize % 16u) == 0) ? (uint32_t*)(b + o_fps) : (uint32_t*)0;
	w.dbg = (unsigned long long*)(b + o_dbg); w.status = (uint32_t*)(b + o_status); w.offsets = (uint64_t*)(b + o_offs); w.frame_len = (uint32_t*)(b + o_flen); w.frame_num = (uint32_t*)(b + o_fnum); w.frame_shift = (uint32_t*)(b + o_fshift); w.window = (const float*)(b + o_window);

	if(c.debug & 0x400000u) (void)hipMemset(w.dbg, 0, 5 * sizeof(uint64_t));   // totals of the estimate spot check start at zero
	// CRC-16 tables (polynomial x^16 + x^15 + x^2 + 1, crc.c:77)
	{
		uint32_t words = 0;
		fa_pack_lds_bytes(&c, &words);
		if(words < c.slot_bytes / 4u + 2u) words = c.slot_bytes / 4u + 2u;   // x^(32 w) up to a whole frame
		if(words < FA_PACK_LARGE_WORDS + 512u) words = FA_PACK_LARGE_WORDS + 512u;   // and up to the second pass's bit buffer
		uint16_t *tab = (uint16_t*)malloc(sizeof(uint16_t) * (1024u + words + 1u));
		if(!tab) { hipFree(e->arena); free(e); return FLACAMD_E_OUT_OF_MEMORY; }
		for(unsigned bb = 0; bb < 256; bb++) { unsigned v = bb << 8; for(int i = 0; i < 8; i++) v = (v & 0x8000u) ? ((v << 1) ^ 0x8005u) & 0xffffu : (v << 1) & 0xffffu; tab[bb] = (uint16_t)v; }
		// tab[256*z + b]: the byte followed by z zero bytes (feeding a zero byte multiplies by x^8)
		for(unsigned z = 1; z < 4; z++) for(unsigned bb = 0; bb < 256; bb++) { const unsigned v = tab[256 * (z - 1) + bb]; tab[256 * z + bb] = (uint16_t)(((v << 8) & 0xffffu) ^ tab[(v >> 8) & 0xffu]); }
		// x^(32 w): feed 4 w zero bytes after a register holding 1
		unsigned v = 1;
		for(unsigned wv = 0; wv <= words; wv++) {
			tab[1024 + wv] = (uint16_t)v;
			for(unsigned z = 0; z < 4; z++) v = ((v << 8) & 0xffffu) ^ tab[(v >> 8) & 0xffu];
		}
		he = hipMemcpy(b + o_crc, tab, sizeof(uint16_t) * (1024u + words + 1u), hipMemcpyHostToDevice);
		free(tab);
		if(he != hipSuccess) { rc = hip_fail(he, "hipMemcpy(crc tables)"); hipFree(e->arena); free(e); return rc; }
		w.crc16_table4 = (const uint16_t*)(b + o_crc); w.crc16_xpw = (const uint16_t*)(b + o_crc) + 1024;
	}

	// window tables, computed on the host exactly once per stream (resize_buffers_, stream_encoder.c:2318-2373)
	{
		float *hw = (float*)malloc(sizeof(float) * (size_t)c.num_apod * c.N);
		if(!hw) { hipFree(e->arena); free(e); return FLACAMD_E_OUT_OF_MEMORY; }
		for(uint32_t a = 0; a < c.num_apod; a++) flacamd_window(p.apod_type[a], p.apod_param[a], (int32_t)c.N, hw + (size_t)a * c.N);
		he = hipMemcpy(b + o_window, hw, sizeof(float) * (size_t)c.num_apod * c.N, hipMemcpyHostToDevice);
		free(hw);
		if(he != hipSuccess) { rc = hip_fail(he, "hipMemcpy(window)"); hipFree(e->arena); free(e); return rc; }
	}
	for(int i = 0; i < 7; i++) (void)hipEventCreate(&e->ev[i]);
	e->cur_ev = e->ev;
	if(hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming) != hipSuccess
		|| hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming) != hipSuccess) { flacamd_encoder_destroy(e); return FLACAMD_E_DEVICE; }
	*out = e;
	return FLACAMD_OK;
}

extern "C" void flacamd_encoder_destroy(flacamd_encoder *e)
{
	if(!e) return;
	hipSetDevice(e->device);
	if(e->q_n) { (void)hipStreamSynchronize(e->q_stream); e->q_n = 0; }   // queued batches still use the work arrays
	for(int i = 0; i < 7; i++) if(e->ev[i]) (void)hipEventDestroy(e->ev[i]);
	if(e->qev_made) for(uint32_t b = 0; b < FLACAMD_QUEUE_DEPTH; b++) for(int i = 0; i < 7; i++) (void)hipEventDestroy(e->qev[b][i]);
	if(e->side) (void)hipStreamDestroy(e->side);
	if(e->q_rec) (void)hipHostFree(e->q_rec);
	if(e->d_pcm16) (void)hipFree(e->d_pcm16);
	if(e->ev_fork) (void)hipEventDestroy(e->ev_fork);
	if(e->ev_join) (void)hipEventDestroy(e->ev_join);
	if(e->arena) hipFree(e->arena);
	if(e->d_pcm_stage) hipFree(e->d_pcm_stage);
	if(e->d_out_stage) hipFree(e->d_out_stage);
	if(e->d_fb_stage) hipFree(e->d_fb_stage);
	if(e->h_tab) (void)hipHostFree(e->h_tab);
	if(e->d_tab) (void)hipFree(e->d_tab);
	if(e->md5s) (void)hipStreamDestroy(e->md5s);
	if(e->ev_tab) (void)hipEventDestroy(e->ev_tab);
	if(e->ev_md5) (void)hipEventDestroy(e->ev_md5);
	free(e);
}

extern "C" void flacamd_encoder_set_timing(flacamd_encoder *e, int enabled) { if(e) e->timing = enabled; }

extern "C" int flacamd_encoder_last_timing(flacamd_encoder *e, float ms[6])
{
	if(!e || !e->have_timing) return FLACAMD_E_INVALID_ARGUMENT;
	for(int i = 0; i < 6; i++) ms[i] = e->last_ms[i];
	return FLACAMD_OK;
}

static int run_batch(flacamd_encoder *e, const FaCfg &c, const void *d_pcm, void *d_out, size_t out_cap, uint32_t *d_frame_bytes, uint64_t *h_total_bytes, hipStream_t st);
static int enqueue_batch(flacamd_encoder *e, const FaCfg &c_in, const void *d_pcm, void *d_out, size_t out_cap, uint32_t *d_frame_bytes, hipStream_t st);

extern "C" int flacamd_encode_device(flacamd_encoder *e, const void *d_pcm, int pcm_format, uint64_t nsamples,
	int final_block, uint32_t first_frame_number, uint32_t prev_assignment,
	void *d_out, size_t out_cap, uint32_t *d_frame_bytes, uint64_t *h_total_bytes, void *stream)
{
	if(!e || !d_pcm || !d_out || nsamples == 0) return FLACAMD_E_INVALID_ARGUMENT;
	if(pcm_format != FLACAMD_PCM_S32 && pcm_format != FLACAMD_PCM_S16) return FLACAMD_E_INVALID_ARGUMENT;
	if(pcm_format == FLACAMD_PCM_S16 && e->p.bits_per_sample > 16) return FLACAMD_E_INVALID_ARGUMENT;
	const uint32_t N = e->cfg.N;
	const uint64_t F64 = (nsamples + N - 1) / N;
	if(!final_block && nsamples % N != 0) return FLACAMD_E_INVALID_ARGUMENT;
	if(F64 > e->max_frames) return FLACAMD_E_CAPACITY;
	hipStream_t st = (hipStream_t)stream;
	HIP_TRY(hipSetDevice(e->device));

	FaCfg c = e->cfg;
	c.pcm_format = (uint32_t)pcm_format;
	c.F = (uint32_t)F64; c.G = c.F * c.nsig;
	c.last_n = (uint32_t)(nsamples - (F64 - 1) * N);
	c.last_fractional = (final_block && c.last_n != N) ? 1u : 0u;
	c.first_frame = first_frame_number; c.prev_assignment = prev_assignment;
	c.multi = 0; c.skip_full = 0;
	return run_batch(e, c, d_pcm, d_out, out_cap, d_frame_bytes, h_total_bytes, st);
}

// The same batch queued behind whatever the stream holds, nothing waited for: the host runs ahead of the device and
// the next batch's first kernel follows this batch's last without the round trip of a synchronisation in between.
// Status word and total go to a page-locked record per outstanding batch; flacamd_encoder_wait reads them.
extern "C" int flacamd_encode_device_queue(flacamd_encoder *e, const void *d_pcm, int pcm_format, uint64_t nsamples,
	int final_block, uint32_t first_frame_number, uint32_t prev_assignment,
	void *d_out, size_t out_cap, uint32_t *d_frame_bytes, void *stream)
{
	if(!e || !d_pcm || !d_out || nsamples == 0) return FLACAMD_E_INVALID_ARGUMENT;
	if(pcm_format != FLACAMD_PCM_S32 && pcm_format != FLACAMD_PCM_S16) return FLACAMD_E_INVALID_ARGUMENT;
	if(pcm_format == FLACAMD_PCM_S16 && e->p.bits_per_sample > 16) return FLACAMD_E_INVALID_ARGUMENT;
	const uint32_t N = e->cfg.N;
	const uint64_t F64 = (nsamples + N - 1) / N;
	if(!final_block && nsamples % N != 0) return FLACAMD_E_INVALID_ARGUMENT;
	if(F64 > e->max_frames) return FLACAMD_E_CAPACITY;
	hipStream_t st = (hipStream_t)stream;
	if(e->q_n >= FLACAMD_QUEUE_DEPTH) return FLACAMD_E_CAPACITY;
	if(e->q_n > 0 && st != e->q_stream) return FLACAMD_E_INVALID_ARGUMENT;   // the batches share the work arrays: one stream orders them
	HIP_TRY(hipSetDevice(e->device));
	if(!e->q_rec) HIP_TRY(hipHostMalloc((void**)&e->q_rec, sizeof(uint64_t) * 2u * FLACAMD_QUEUE_DEPTH, hipHostMallocDefault));

	FaCfg c = e->cfg;
	c.pcm_format = (uint32_t)pcm_format;
	c.F = (uint32_t)F64; c.G = c.F * c.nsig;
	c.last_n = (uint32_t)(nsamples - (F64 - 1) * N);
	c.last_fractional = (final_block && c.last_n != N) ? 1u : 0u;
	c.first_frame = first_frame_number; c.prev_assignment = prev_assignment;
	c.multi = 0; c.skip_full = 0;
	if(e->timing && !e->qev_made) {
		for(uint32_t b = 0; b < FLACAMD_QUEUE_DEPTH; b++) for(int i = 0; i < 7; i++) HIP_TRY(hipEventCreate(&e->qev[b][i]));
		e->qev_made = 1;
	}
	if(e->timing) e->cur_ev = e->qev[e->q_n];
	const int rc = enqueue_batch(e, c, d_pcm, d_out, out_cap, d_frame_bytes, st);
	e->cur_ev = e->ev;
	if(rc != FLACAMD_OK) return rc;
	uint64_t *rec = e->q_rec + 2u * e->q_n;
	rec[0] = 0;
	HIP_TRY(hipMemcpyAsync(&rec[0], e->work.status, 4, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(&rec[1], e->work.offsets + c.F, 8, hipMemcpyDeviceToHost, st));
	e->q_stream = st; e->q_n++;
	return FLACAMD_OK;
}

extern "C" int flacamd_encoder_wait(flacamd_encoder *e, uint64_t *h_total_bytes, uint32_t max_totals, uint32_t *n_batches)
{
	if(!e) return FLACAMD_E_INVALID_ARGUMENT;
	if(n_batches) *n_batches = e->q_n;
	if(e->q_n == 0) return FLACAMD_OK;
	HIP_TRY(hipSetDevice(e->device));
	const uint32_t n = e->q_n;
	e->q_n = 0;
	HIP_TRY(hipStreamSynchronize(e->q_stream));
	if(e->timing && e->qev_made) {   // stage times: the mean over the batches waited for
		for(int i = 0; i < 6; i++) {
			float sum = 0.0f, ms = 0.0f;
			for(uint32_t b = 0; b < n; b++) { if(hipEventElapsedTime(&ms, e->qev[b][i], e->qev[b][i+1]) == hipSuccess) sum += ms; }
			e->last_ms[i] = sum / (float)n;
		}
		e->have_timing = 1;
	}
	int rc = FLACAMD_OK;
	for(uint32_t i = 0; i < n; i++) {
		const uint32_t status = (uint32_t)e->q_rec[2u * i];
		if(h_total_bytes && i < max_totals) h_total_bytes[i] = e->q_rec[2u * i + 1u];
		if(rc == FLACAMD_OK && (status & 1u)) rc = FLACAMD_E_FRAME_OVERFLOW;
		if(rc == FLACAMD_OK && (status & 2u)) rc = FLACAMD_E_CAPACITY;
	}
	return rc;
}

// A frame did not fit its slot.  Slots are sized for a verbatim frame with slack, and the encoder only keeps a subframe
// whose ESTIMATE beats verbatim - but the reference's partition sums wrap at 32 bits where it takes them narrow
// (stream_encoder.c:3808), so on full-scale noise of wide samples an estimate can fall far short of the size and the
// reference then writes a frame several times the verbatim size.  To write the same frame: the pack and gather stages
// once more with slots 16 times as large, through the general pack kernel (which flushes in the middle of a subframe).
static int repack_long(flacamd_encoder *e, const FaCfg &c, const void *d_pcm, void *d_out, size_t out_cap, uint32_t *d_frame_bytes, uint32_t *status, uint64_t *total, hipStream_t st)
{
	size_t big = (size_t)c.slot_bytes * 16u;
	if(big > ((size_t)1 << 24)) big = (size_t)1 << 24;   // a frame size has 24 bits in STREAMINFO (format.c:72)
	if(big <= c.slot_bytes) return FLACAMD_E_FRAME_OVERFLOW;
	void *tmp = 0;
	if(hipMalloc(&tmp, (size_t)c.F * big) != hipSuccess) { (void)hipGetLastError(); return FLACAMD_E_FRAME_OVERFLOW; }
	FaCfg cc = c; FaWork ww = e->work;
	cc.slot_bytes = (uint32_t)big; ww.slots = (uint8_t*)tmp;
	hipError_t he = hipMemsetAsync(e->work.status, 0, 4, st);
	fa_launch_pack_general(d_pcm, &cc, &ww, st, 1);
	fa_launch_gather(&cc, &ww, d_out, out_cap, d_frame_bytes, st);
	if(he == hipSuccess) he = hipGetLastError();
	if(he == hipSuccess) he = hipMemcpyAsync(status, e->work.status, 4, hipMemcpyDeviceToHost, st);
	if(he == hipSuccess) he = hipMemcpyAsync(total, e->work.offsets + c.F, 8, hipMemcpyDeviceToHost, st);
	if(he == hipSuccess) he = hipStreamSynchronize(st);
	(void)hipFree(tmp);
	if(he != hipSuccess) return hip_fail(he, "repack_long");
	return FLACAMD_OK;
}

// 32-bit samples of a stream of at most 16 bits, narrowed: four values per thread
__global__ void __launch_bounds__(256) fa_narrow_kernel(const int4 *__restrict__ src, uint2 *__restrict__ dst, size_t nquads, const int32_t *__restrict__ src1, int16_t *__restrict__ dst1, uint32_t ntail)
{
	const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
	if(i < nquads) {
		const int4 v = src[i];
		dst[i] = make_uint2(((uint32_t)v.x & 0xffffu) | ((uint32_t)v.y << 16), ((uint32_t)v.z & 0xffffu) | ((uint32_t)v.w << 16));
	}
	if(i < ntail) dst1[i] = (int16_t)src1[i];
}

// The stages of one batch on stream st, nothing else: no readback, no synchronisation.
static int enqueue_batch(flacamd_encoder *e, const FaCfg &c_in, const void *d_pcm, void *d_out, size_t out_cap, uint32_t *d_frame_bytes, hipStream_t st)
{
	FaCfg c = c_in;
	e->last_F = c.F; e->last_G = c.G;
	// A stream of at most 16 bits handed over as 32-bit samples (the layout process_interleaved takes): one pass turns the batch
	// into 16-bit samples (0.25 ms per hour of stereo), and every stage then runs as it does on a 16-bit buffer - half the bytes per
	// read, the packed-pair kernels.  (The stream object's pipe narrows on the host for the same reason.)
	// (Single-stream batches: where a multi-stream batch's buffer ends is not known here.)
	if(c.pcm_format == FLACAMD_PCM_S32 && c.bps <= 16u && !c.multi && !(c.debug & 0x1000000u) && ((uintptr_t)d_pcm % 16u) == 0) {
		const size_t total_values = ((size_t)(c.F - 1u) * c.N + c.last_n) * c.channels;
		const size_t need = (total_values * 2u + 15u) & ~(size_t)15u;
		if(e->pcm16_bytes < need) {
			if(e->d_pcm16) { HIP_TRY(hipStreamSynchronize(st)); (void)hipFree(e->d_pcm16); e->d_pcm16 = 0; e->pcm16_bytes = 0; }
			const size_t cap = (size_t)e->max_frames * c.N * c.channels * 2u + 16u;
			HIP_TRY(hipMalloc(&e->d_pcm16, cap > need ? cap : need));
			e->pcm16_bytes = cap > need ? cap : need;
		}
		const size_t nquads = total_values / 4u; const uint32_t ntail = (uint32_t)(total_values - nquads * 4u);
		const size_t threads = nquads > ntail ? nquads : ntail;
		hipLaunchKernelGGL(fa_narrow_kernel, dim3((unsigned)((threads + 255u) / 256u)), dim3(256), 0, st, (const int4*)d_pcm, (uint2*)e->d_pcm16, nquads,
			(const int32_t*)d_pcm + nquads * 4u, (int16_t*)e->d_pcm16 + nquads * 4u, ntail);
		d_pcm = e->d_pcm16; c.pcm_format = FLACAMD_PCM_S16;
	}

	HIP_TRY(hipMemsetAsync(e->work.status, 0, 4, st));
	const bool tm = e->timing != 0;
	if(tm) (void)hipEventRecord(e->cur_ev[0], st);
	// prep and autocorrelation read the same PCM and do not depend on each other (the model stage
	// rescales the sums by the wasted bits): side by side on two streams, joined before the model
	const bool fork = c.max_lpc_order > 0 && !(c.debug & 0x10000u);
	// the fixed predictors' partition sums are a by-product of the preparation kernel's difference pass: the search kernel need not redo it
	c.fixed_psum = (e->work.fixed_psum && fa_prep_fast_ok(d_pcm, &c) && fa_search_fast_ok(d_pcm, &c) && !(c.debug & 0x40000u)) ? 1u : 0u;
	// The autocorrelation kernel is the longer of the two and goes on the caller's stream: it starts the moment the
	// batch before it ends, while whatever goes to the side stream first waits for the fork event to cross queues
	// (0.3 ms late when batches are queued back to back)
	const bool swap = fork && !(c.debug & 0x400000u);
	if(fork) {
		HIP_TRY(hipEventRecord(e->ev_fork, st));
		HIP_TRY(hipStreamWaitEvent(e->side, e->ev_fork, 0));
		fa_launch_autoc(d_pcm, &c, &e->work, swap ? st : e->side);
	}
	// (a short last block has a preparation kernel of its own: behind the autocorrelation kernel, which ends first)
	if(swap) fa_launch_prep(d_pcm, &c, &e->work, e->side, e->side);
	else fa_launch_prep(d_pcm, &c, &e->work, st, fork ? e->side : st);
	if(fork) HIP_TRY(hipEventRecord(e->ev_join, e->side));
	if(tm) (void)hipEventRecord(e->cur_ev[1], st);
	if(fork) HIP_TRY(hipStreamWaitEvent(st, e->ev_join, 0));
	else if(c.max_lpc_order > 0) fa_launch_autoc(d_pcm, &c, &e->work, st);
	if(tm) (void)hipEventRecord(e->cur_ev[2], st);
	fa_launch_model(&c, &e->work, st);
	if(tm) hipEventRecord(e->cur_ev[3], st);
	if(fa_search_rest_frames(d_pcm, &c) && !(c.debug & 0x10000u)) {
		// the short last frame's search beside the others' (it is a handful of workgroups with the chip to themselves otherwise)
		HIP_TRY(hipEventRecord(e->ev_fork, st));
		HIP_TRY(hipStreamWaitEvent(e->side, e->ev_fork, 0));
		fa_launch_search2(d_pcm, &c, &e->work, e->cand_group, e->sum64, st, e->side);
		HIP_TRY(hipEventRecord(e->ev_join, e->side));
		HIP_TRY(hipStreamWaitEvent(st, e->ev_join, 0));
	}
	else fa_launch_search(d_pcm, &c, &e->work, e->cand_group, e->sum64, st);
	if(tm) hipEventRecord(e->cur_ev[4], st);
	// frames straight into the caller's buffer where the register-resident pack kernel takes the batch; else into slots and a copy
	c.direct_out = (fa_pack_direct_ok(d_pcm, &c) && !(c.debug & 3u)) ? 1u : 0u;
	FaWork ww = e->work;
	ww.out = (uint8_t*)d_out; ww.out_cap = out_cap; ww.frame_bytes_out = d_frame_bytes;
	if(c.direct_out) HIP_TRY(hipMemsetAsync(ww.lb, 0, (size_t)c.F * 12u, st));
	if(!(c.debug & 3u)) fa_launch_pack(d_pcm, &c, &ww, st);
	if(tm) hipEventRecord(e->cur_ev[5], st);
	if(c.direct_out) fa_launch_offsets(&c, &ww, st);
	else fa_launch_gather(&c, &e->work, d_out, out_cap, d_frame_bytes, st);
	if(tm) hipEventRecord(e->cur_ev[6], st);
	HIP_TRY(hipGetLastError());
	return FLACAMD_OK;
}

// One batch on stream st; synchronises once at the end to read the status word and the total.
static int run_batch(flacamd_encoder *e, const FaCfg &c, const void *d_pcm, void *d_out, size_t out_cap, uint32_t *d_frame_bytes, uint64_t *h_total_bytes, hipStream_t st)
{
	const int rc = enqueue_batch(e, c, d_pcm, d_out, out_cap, d_frame_bytes, st);
	if(rc != FLACAMD_OK) return rc;
	const bool tm = e->timing != 0;
	uint32_t status = 0; uint64_t total = 0;
	HIP_TRY(hipMemcpyAsync(&status, e->work.status, 4, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(&total, e->work.offsets + c.F, 8, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	if(tm) { for(int i = 0; i < 6; i++) hipEventElapsedTime(&e->last_ms[i], e->ev[i], e->ev[i+1]); e->have_timing = 1; }
	if(status & 1u) { const int rc2 = repack_long(e, c, d_pcm, d_out, out_cap, d_frame_bytes, &status, &total, st); if(rc2 != FLACAMD_OK) return rc2; }
	if(h_total_bytes) *h_total_bytes = total;   // with FLACAMD_E_CAPACITY: the bytes the output buffer needs
	if(status & 1u) return FLACAMD_E_FRAME_OVERFLOW;
	if(status & 2u) return FLACAMD_E_CAPACITY;
	return FLACAMD_OK;
}

// grow-only pinned staging / device table buffers of the multi-stream entry points (no allocation on the steady path)
static int tab_reserve(flacamd_encoder *e, size_t host_bytes, size_t dev_bytes)
{
	if(e->h_tab_bytes < host_bytes) {
		if(e->h_tab) (void)hipHostFree(e->h_tab);
		e->h_tab = 0; e->h_tab_bytes = 0;
		const size_t want = host_bytes + host_bytes / 2;
		HIP_TRY(hipHostMalloc(&e->h_tab, want, hipHostMallocDefault));
		e->h_tab_bytes = want;
	}
	if(e->d_tab_bytes < dev_bytes) {
		if(e->d_tab) (void)hipFree(e->d_tab);
		e->d_tab = 0; e->d_tab_bytes = 0;
		const size_t want = dev_bytes + dev_bytes / 2;
		HIP_TRY(hipMalloc(&e->d_tab, want));
		e->d_tab_bytes = want;
	}
	return FLACAMD_OK;
}

// frames of a multi-stream batch: checks that the streams tile the buffer in whole blocks, returns the frame count
static int streams_frames(const flacamd_encoder *e, uint32_t nstreams, const uint64_t *h_start, const uint64_t *h_len, uint64_t *F_out)
{
	const uint32_t N = e->cfg.N;
	uint64_t F64 = 0;
	for(uint32_t i = 0; i < nstreams; i++) {
		if(h_len[i] == 0 || h_start[i] != F64 * N) return FLACAMD_E_INVALID_ARGUMENT;
		F64 += (h_len[i] + N - 1) / N;
		if(F64 > e->max_frames) return FLACAMD_E_CAPACITY;
	}
	*F_out = F64;
	return FLACAMD_OK;
}

extern "C" int flacamd_encode_streams_device(flacamd_encoder *e, const void *d_pcm, int pcm_format, uint32_t nstreams,
	const uint64_t *h_start, const uint64_t *h_len, void *d_out, size_t out_cap, uint32_t *d_frame_bytes, uint64_t *h_total_bytes, void *stream)
{
	if(!e || !d_pcm || !d_out || nstreams == 0 || !h_start || !h_len) return FLACAMD_E_INVALID_ARGUMENT;
	if(pcm_format != FLACAMD_PCM_S32 && pcm_format != FLACAMD_PCM_S16) return FLACAMD_E_INVALID_ARGUMENT;
	if(pcm_format == FLACAMD_PCM_S16 && e->p.bits_per_sample > 16) return FLACAMD_E_INVALID_ARGUMENT;
	const uint32_t N = e->cfg.N;
	// frame f of the batch starts at sample f * N: the streams must tile the buffer in whole blocks
	uint64_t F64 = 0;
	int rc = streams_frames(e, nstreams, h_start, h_len, &F64);
	if(rc != FLACAMD_OK) return rc;
	hipStream_t st = (hipStream_t)stream;
	HIP_TRY(hipSetDevice(e->device));
	rc = tab_reserve(e, sizeof(uint32_t) * 3u * (size_t)F64, 0);
	if(rc != FLACAMD_OK) return rc;
	uint32_t *tab = (uint32_t*)e->h_tab;   // free again: every call ends with the stream synchronised
	uint32_t nshort = 0;                   // frames shorter than the blocksize (a stream's last), listed for the general search kernel
	{
		uint32_t f = 0;
		for(uint32_t i = 0; i < nstreams; i++) {
			const uint64_t nf = (h_len[i] + N - 1) / N;
			for(uint64_t k = 0; k < nf; k++, f++) {
				tab[f] = (k + 1 == nf) ? (uint32_t)(h_len[i] - k * N) : N;
				tab[F64 + f] = (uint32_t)k;
				if(tab[f] != N) tab[2u * F64 + nshort++] = f;
			}
		}
	}
	HIP_TRY(hipMemcpyAsync(e->work.frame_len, tab, sizeof(uint32_t) * (size_t)F64, hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemcpyAsync(e->work.frame_num, tab + F64, sizeof(uint32_t) * (size_t)F64, hipMemcpyHostToDevice, st));
	if(nshort) HIP_TRY(hipMemcpyAsync(e->work.short_list, tab + 2u * F64, sizeof(uint32_t) * (size_t)nshort, hipMemcpyHostToDevice, st));

	FaCfg c = e->cfg;
	c.pcm_format = (uint32_t)pcm_format;
	c.F = (uint32_t)F64; c.G = c.F * c.nsig;
	c.last_n = N; c.last_fractional = 0; c.first_frame = 0; c.prev_assignment = 0;
	c.multi = 1; c.skip_full = 0; c.nshort = nshort;
	return run_batch(e, c, d_pcm, d_out, out_cap, d_frame_bytes, h_total_bytes, st);
}

// ---- a batch of clips as complete streams ("fLaC", STREAMINFO, VORBIS_COMMENT, frames), assembled on the device
extern "C" void fa_launch_md5(const void *d_pcm, uint32_t pcm_format, uint32_t channels, uint32_t bits_per_sample, const uint64_t *d_starts, const uint64_t *d_ends,
	uint32_t nstreams, uint8_t *d_digests, hipStream_t st);

__constant__ char fa_vendor_string[33] = "reference libFLAC 1.2.1 20070917";   // format.c:56-63

// One wave per clip: smallest and largest frame of the clip, then the 86 bytes a finished stream starts with
// (stream_encoder.c:1034-1106 as written at init and patched by update_metadata_, :2521-2620), written into the room
// the gather stage left in front of the clip's frames.
__global__ void __launch_bounds__(64) fa_clip_heads_kernel(FaCfg c, FaWork w, const uint64_t *__restrict__ starts, const uint64_t *__restrict__ ends,
	const uint32_t *__restrict__ first, const uint32_t *__restrict__ nframes, uint32_t nclips, const uint8_t *__restrict__ digests,
	uint8_t *__restrict__ out, unsigned long long out_cap, uint64_t *__restrict__ stream_off)
{
	__shared__ uint8_t h[FLACAMD_STREAM_HEAD_BYTES];
	const uint32_t clip = blockIdx.x, lane = threadIdx.x;
	const uint32_t f0 = first[clip], nf = nframes[clip];
	uint32_t mn = 0xffffffffu, mx = 0;
	for(uint32_t i = lane; i < nf; i += 64u) { const uint32_t v = w.frame_bytes[f0 + i]; mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
#pragma unroll
	for(int o = 32; o >= 1; o >>= 1) { const uint32_t a = (uint32_t)__shfl_xor((int)mn, o, 64), b = (uint32_t)__shfl_xor((int)mx, o, 64); mn = a < mn ? a : mn; mx = b > mx ? b : mx; }
	const unsigned long long off = w.offsets[f0] + (unsigned long long)FLACAMD_STREAM_HEAD_BYTES * clip;
	if(lane == 0) {
		stream_off[clip] = off;
		if(clip + 1u == nclips) stream_off[nclips] = w.offsets[c.F] + (unsigned long long)FLACAMD_STREAM_HEAD_BYTES * nclips;
	}
	if(off + FLACAMD_STREAM_HEAD_BYTES > out_cap) { if(lane == 0) atomicOr(w.status, 2u); return; }
	if(lane == 0) {
		const unsigned long long total = ends[clip] - starts[clip];
		const unsigned long long packed = ((unsigned long long)c.rate << 44) | ((unsigned long long)(c.channels - 1u) << 41) | ((unsigned long long)(c.bps - 1u) << 36) | (total & 0xFFFFFFFFFull);
		h[0] = 'f'; h[1] = 'L'; h[2] = 'a'; h[3] = 'C';
		h[4] = 0; h[5] = 0; h[6] = 0; h[7] = 34;
		h[8] = (uint8_t)(c.N >> 8); h[9] = (uint8_t)c.N; h[10] = h[8]; h[11] = h[9];
		h[12] = (uint8_t)(mn >> 16); h[13] = (uint8_t)(mn >> 8); h[14] = (uint8_t)mn;
		h[15] = (uint8_t)(mx >> 16); h[16] = (uint8_t)(mx >> 8); h[17] = (uint8_t)mx;
		for(int i = 0; i < 8; i++) h[18 + i] = (uint8_t)(packed >> (8 * (7 - i)));
		for(int i = 0; i < 16; i++) h[26 + i] = digests[(size_t)clip * 16u + i];
		h[42] = 0x84; h[43] = 0; h[44] = 0; h[45] = 8 + 32;
		h[46] = 32; h[47] = 0; h[48] = 0; h[49] = 0;
		for(int i = 0; i < 32; i++) h[50 + i] = (uint8_t)fa_vendor_string[i];
		h[82] = h[83] = h[84] = h[85] = 0;
	}
	__syncthreads();
	for(uint32_t i = lane; i < FLACAMD_STREAM_HEAD_BYTES; i += 64u) out[off + i] = h[i];
}

extern "C" int flacamd_encoder_use_digests(flacamd_encoder *e, const uint8_t *d_digests, void *ready_event)
{
	if(!e) return FLACAMD_E_INVALID_ARGUMENT;
	e->given_digests = d_digests; e->given_ready = d_digests ? ready_event : 0;
	return FLACAMD_OK;
}

extern "C" int flacamd_encode_clips_device(flacamd_encoder *e, const void *d_pcm, int pcm_format, uint32_t nclips,
	const uint64_t *h_start, const uint64_t *h_len, void *d_out, size_t out_cap, uint64_t *h_stream_off, uint64_t *h_total_bytes, void *stream)
{
	if(!e || !d_pcm || !d_out || nclips == 0 || !h_start || !h_len || !h_stream_off) return FLACAMD_E_INVALID_ARGUMENT;
	if(pcm_format != FLACAMD_PCM_S32 && pcm_format != FLACAMD_PCM_S16) return FLACAMD_E_INVALID_ARGUMENT;
	if(pcm_format == FLACAMD_PCM_S16 && e->p.bits_per_sample > 16) return FLACAMD_E_INVALID_ARGUMENT;
	if((uint64_t)nclips * FLACAMD_STREAM_HEAD_BYTES > 0xffffffffull) return FLACAMD_E_CAPACITY;   // the per-frame shift is 32 bits
	const uint32_t N = e->cfg.N;
	uint64_t F64 = 0;
	int rc = streams_frames(e, nclips, h_start, h_len, &F64);
	if(rc != FLACAMD_OK) return rc;
	hipStream_t st = (hipStream_t)stream;
	HIP_TRY(hipSetDevice(e->device));
	if(!e->md5s) {
		// a stream of another priority than the caller's: streams of one priority take the process's few hardware queues in
		// turn, and a hash as long as the longest clip must not sit in front of another encoder's kernels in the same queue
		int least = 0, greatest = 0;
		HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
		HIP_TRY(hipStreamCreateWithPriority(&e->md5s, hipStreamNonBlocking, greatest));
		HIP_TRY(hipEventCreateWithFlags(&e->ev_tab, hipEventDisableTiming));
		HIP_TRY(hipEventCreateWithFlags(&e->ev_md5, hipEventDisableTiming));
	}
	// host staging: frame_len[F] frame_num[F] frame_shift[F] short frames[F] | starts[n] ends[n] (u64) first[n] nframes[n] (u32) | stream_off[n + 1] (coming back)
	const size_t F = (size_t)F64, n = nclips;
	const size_t o_clip = (4u * F * 4u + 7u) & ~(size_t)7u, clip_bytes = n * (8u + 8u + 4u + 4u), o_back = o_clip + ((clip_bytes + 7u) & ~(size_t)7u);
	const size_t d_dig_off = (clip_bytes + 15u) & ~(size_t)15u, d_soff = d_dig_off + n * 16u;
	rc = tab_reserve(e, o_back + (n + 1u) * 8u, d_soff + (n + 1u) * 8u);
	if(rc != FLACAMD_OK) return rc;
	uint8_t *hb = (uint8_t*)e->h_tab;
	uint32_t *flen = (uint32_t*)hb, *fnum = flen + F, *fshift = fnum + F, *fshort = fshift + F;
	uint32_t nshort = 0;   // frames shorter than the blocksize (a clip's last), listed for the general search kernel
	uint64_t *cs = (uint64_t*)(hb + o_clip), *ce = cs + n; uint32_t *cf = (uint32_t*)(ce + n), *cn = cf + n;
	{
		uint32_t f = 0;
		for(uint32_t i = 0; i < nclips; i++) {
			const uint64_t nf = (h_len[i] + N - 1) / N;
			cs[i] = h_start[i]; ce[i] = h_start[i] + h_len[i]; cf[i] = f; cn[i] = (uint32_t)nf;
			for(uint64_t k = 0; k < nf; k++, f++) {
				flen[f] = (k + 1 == nf) ? (uint32_t)(h_len[i] - k * N) : N;
				fnum[f] = (uint32_t)k;
				fshift[f] = (uint32_t)FLACAMD_STREAM_HEAD_BYTES * (i + 1u);
				if(flen[f] != N) fshort[nshort++] = f;
			}
		}
	}
	uint8_t *db = (uint8_t*)e->d_tab;
	const uint64_t *d_cs = (const uint64_t*)db, *d_ce = d_cs + n; const uint32_t *d_cf = (const uint32_t*)(d_ce + n), *d_cn = d_cf + n;
	uint8_t *d_dig = db + d_dig_off; uint64_t *d_so = (uint64_t*)(db + d_soff);
	HIP_TRY(hipMemcpyAsync(e->work.frame_len, flen, F * 4u, hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemcpyAsync(e->work.frame_num, fnum, F * 4u, hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemcpyAsync(e->work.frame_shift, fshift, F * 4u, hipMemcpyHostToDevice, st));
	if(nshort) HIP_TRY(hipMemcpyAsync(e->work.short_list, fshort, (size_t)nshort * 4u, hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemcpyAsync(db, hb + o_clip, clip_bytes, hipMemcpyHostToDevice, st));
	HIP_TRY(hipEventRecord(e->ev_tab, st));
	// the digests: one lane per clip, a serial chain as long as the clip (md5.c:394-420); beside the coding stages -
	// unless the caller had them made already (flacamd_md5_groups_device hashes the clips of many batches in one pass)
	const uint8_t *use_dig = e->given_digests; void *use_ready = e->given_ready;
	e->given_digests = 0; e->given_ready = 0;
	if(!use_dig) {
		HIP_TRY(hipStreamWaitEvent(e->md5s, e->ev_tab, 0));
		fa_launch_md5(d_pcm, (uint32_t)pcm_format, e->cfg.channels, e->cfg.bps, d_cs, d_ce, nclips, d_dig, e->md5s);
		HIP_TRY(hipEventRecord(e->ev_md5, e->md5s));
	}

	FaCfg c = e->cfg;
	c.pcm_format = (uint32_t)pcm_format;
	c.F = (uint32_t)F64; c.G = c.F * c.nsig;
	c.last_n = N; c.last_fractional = 0; c.first_frame = 0; c.prev_assignment = 0;
	c.multi = 2; c.skip_full = 0; c.nshort = nshort;
	uint64_t total = 0;
	rc = run_batch(e, c, d_pcm, d_out, out_cap, 0, &total, st);
	const uint64_t need = total + (uint64_t)FLACAMD_STREAM_HEAD_BYTES * nclips;
	if(h_total_bytes) *h_total_bytes = need;
	if(rc != FLACAMD_OK) { if(!use_dig) (void)hipStreamSynchronize(e->md5s); return rc; }
	if(!use_dig) HIP_TRY(hipStreamWaitEvent(st, e->ev_md5, 0));
	else if(use_ready) HIP_TRY(hipStreamWaitEvent(st, (hipEvent_t)use_ready, 0));
	HIP_TRY(hipMemsetAsync(e->work.status, 0, 4, st));
	hipLaunchKernelGGL(fa_clip_heads_kernel, dim3(nclips), dim3(64), 0, st, c, e->work, d_cs, d_ce, d_cf, d_cn, nclips, use_dig ? use_dig : (const uint8_t*)d_dig, (uint8_t*)d_out, (unsigned long long)out_cap, d_so);
	HIP_TRY(hipGetLastError());
	uint32_t status = 0;
	HIP_TRY(hipMemcpyAsync(hb + o_back, d_so, (n + 1u) * 8u, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(&status, e->work.status, 4, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	if(status & 2u) return FLACAMD_E_CAPACITY;
	memcpy(h_stream_off, hb + o_back, (n + 1u) * 8u);
	return FLACAMD_OK;
}

extern "C" int flacamd_encode_host(flacamd_encoder *e, const void *h_pcm, int pcm_format, uint64_t nsamples,
	int final_block, uint32_t first_frame_number, uint32_t prev_assignment,
	uint8_t *h_out, size_t out_cap, uint32_t *h_frame_bytes, uint64_t *h_total_bytes, uint32_t *h_last_assignment)
{
	if(!e || !h_pcm || !h_out || nsamples == 0) return FLACAMD_E_INVALID_ARGUMENT;
	HIP_TRY(hipSetDevice(e->device));
	const size_t bytes_per = (pcm_format == FLACAMD_PCM_S16 ? 2u : 4u) * (size_t)e->p.channels;
	const size_t pcm_bytes = (size_t)nsamples * bytes_per;
	const uint64_t F = (nsamples + e->cfg.N - 1) / e->cfg.N;
	if(F > e->max_frames) return FLACAMD_E_CAPACITY;
	if(e->pcm_stage_bytes < pcm_bytes) {
		if(e->d_pcm_stage) hipFree(e->d_pcm_stage);
		e->d_pcm_stage = 0; e->pcm_stage_bytes = 0;
		HIP_TRY(hipMalloc(&e->d_pcm_stage, pcm_bytes));
		e->pcm_stage_bytes = pcm_bytes;
	}
	const size_t need_out = (size_t)F * e->cfg.slot_bytes;
	if(e->out_stage_bytes < need_out) {
		if(e->d_out_stage) hipFree(e->d_out_stage);
		if(e->d_fb_stage) hipFree(e->d_fb_stage);
		e->d_out_stage = 0; e->d_fb_stage = 0; e->out_stage_bytes = 0;
		HIP_TRY(hipMalloc((void**)&e->d_out_stage, need_out));
		HIP_TRY(hipMalloc((void**)&e->d_fb_stage, (size_t)e->max_frames * 4u));
		e->out_stage_bytes = need_out;
	}
	HIP_TRY(hipMemcpy(e->d_pcm_stage, h_pcm, pcm_bytes, hipMemcpyHostToDevice));
	uint64_t total = 0;
	int rc = flacamd_encode_device(e, e->d_pcm_stage, pcm_format, nsamples, final_block, first_frame_number, prev_assignment,
		e->d_out_stage, e->out_stage_bytes, e->d_fb_stage, &total, 0);
	if(rc == FLACAMD_E_CAPACITY && total > e->out_stage_bytes) {   // frames beyond the usual bound (see repack_long): a larger staging buffer, once more
		hipFree(e->d_out_stage); e->d_out_stage = 0; e->out_stage_bytes = 0;
		HIP_TRY(hipMalloc((void**)&e->d_out_stage, total));
		e->out_stage_bytes = total;
		rc = flacamd_encode_device(e, e->d_pcm_stage, pcm_format, nsamples, final_block, first_frame_number, prev_assignment,
			e->d_out_stage, e->out_stage_bytes, e->d_fb_stage, &total, 0);
	}
	if(h_total_bytes) *h_total_bytes = total;
	if(rc != FLACAMD_OK) return rc;
	if(total > out_cap) return FLACAMD_E_CAPACITY;   // *h_total_bytes tells the caller how much it takes
	HIP_TRY(hipMemcpy(h_out, e->d_out_stage, total, hipMemcpyDeviceToHost));
	if(h_frame_bytes) HIP_TRY(hipMemcpy(h_frame_bytes, e->d_fb_stage, (size_t)F * 4u, hipMemcpyDeviceToHost));
	if(h_last_assignment) HIP_TRY(hipMemcpy(h_last_assignment, e->work.assignment + (F - 1), 4, hipMemcpyDeviceToHost));
	if(h_total_bytes) *h_total_bytes = total;
	return FLACAMD_OK;
}

// Totals of the estimate spot check (FLACAMD_DEBUG bit 22 at the time the encoder was made; k_pack.hip), optionally
// cleared: subframes looked at, subframes whose estimate differs from the bits written, sum and maximum of the
// differences, bits written.  (The development timestamps share these words and are not taken in the same run.)
extern "C" int flacamd_encoder_spotcheck(flacamd_encoder *e, uint64_t out[5], int reset)
{
	if(!e || !out) return FLACAMD_E_INVALID_ARGUMENT;
	if(!(e->cfg.debug & 0x400000u)) return FLACAMD_E_UNSUPPORTED;
	if(hipSetDevice(e->device) != hipSuccess) return FLACAMD_E_DEVICE;
	if(hipMemcpy(out, e->work.dbg, 5 * sizeof(uint64_t), hipMemcpyDeviceToHost) != hipSuccess) return FLACAMD_E_DEVICE;
	if(reset && hipMemset(e->work.dbg, 0, 5 * sizeof(uint64_t)) != hipSuccess) return FLACAMD_E_DEVICE;
	return FLACAMD_OK;
}

extern "C" long flacamd_encoder_debug_read(flacamd_encoder *e, int what, void *h_dst, size_t cap)
{
	if(!e || !h_dst || e->last_G == 0) return FLACAMD_E_INVALID_ARGUMENT;
	if(hipSetDevice(e->device) != hipSuccess) return FLACAMD_E_DEVICE;
	const size_t G = e->last_G;
	const void *src = 0; size_t bytes = 0;
	switch(what) {
	case 0: src = e->work.autoc; bytes = (size_t)e->cfg.num_apod * e->cfg.lag * G * 4u; break;
	case 1: src = e->work.fixed_err; bytes = 5u * G * 8u; break;
	case 2: break;
	case 3: break;
	case 4: src = e->work.wasted; bytes = G; break;
	case 5: src = e->work.dbg; bytes = (size_t)FA_DBG_WORDS * 8u; break;
	default: return FLACAMD_E_INVALID_ARGUMENT;
	}
	if(what == 2 || what == 3) {
		// strided copy out of the winner records
		FaWinner *tmp = (FaWinner*)malloc(G * sizeof(FaWinner));
		if(!tmp) return FLACAMD_E_OUT_OF_MEMORY;
		if(hipMemcpy(tmp, e->work.winner, G * sizeof(FaWinner), hipMemcpyDeviceToHost) != hipSuccess) { free(tmp); return FLACAMD_E_DEVICE; }
		bytes = what == 2 ? G * 16u : G * 4u;
		if(bytes > cap) { free(tmp); return FLACAMD_E_CAPACITY; }
		for(size_t g = 0; g < G; g++) {
			if(what == 2) memcpy((uint8_t*)h_dst + g * 16u, tmp[g].hdr, 16);
			else memcpy((uint8_t*)h_dst + g * 4u, &tmp[g].hdr[1], 4);
		}
		free(tmp);
		return (long)bytes;
	}
	// note: the layouts above are strided by the G of the LAST batch
	if(bytes > cap) return FLACAMD_E_CAPACITY;
	if(hipMemcpy(h_dst, src, bytes, hipMemcpyDeviceToHost) != hipSuccess) return FLACAMD_E_DEVICE;
	return (long)bytes;
}

// ------------------------------------------------------------------ pipelined encoding from host memory
//
// One stream, PCM arriving in host memory, frames wanted in host memory: batches of frames are kept in flight so
// that the upload of batch k+1, the kernels of batch k and the download of batch k-1 overlap, and consecutive
// batches go to the listed GPUs in turn (a frame depends on nothing but its own samples and its number,
// stream_encoder.c:2866-2923, so frame ranges can be coded anywhere and handed back in order).  A batch border is
// always the start of a loose mid/side decision period (stream_encoder.c:2958-2967), which makes a batch
// independent of the assignment chosen before it.
//
// Each GPU listed ("lane") has one encoder workspace and three streams: uploads, kernels, downloads.  A slot is
// the storage of one batch in flight: pinned host PCM, device PCM, device frames, pinned host frames and sizes.
// The caller fills the pinned PCM of the next slot itself (no second host copy), submits it, and collects batches
// in submission order.

struct PipeMeta { uint32_t status, last_assignment; uint64_t total; };

// Pinned host buffers are expensive to make and, above all, to touch for the first time (the first pass over fresh
// pinned pages runs at a few GB/s), which is what a short stream would spend most of its time on.  The buffers of a
// destroyed pipe therefore go to a small process-wide free list and the next pipe of the same geometry takes them
// over, warm.  At most PIN_CACHE_ENTRIES buffers / PIN_CACHE_BYTES are kept; flacamd_pipe_trim() releases them.
#define PIN_CACHE_ENTRIES 32
#define PIN_CACHE_BYTES ((size_t)1 << 30)
static std::mutex g_pin_lock;
static struct { void *p; size_t bytes; } g_pin[PIN_CACHE_ENTRIES];
static size_t g_pin_total;
// the true size of every buffer handed out: what a buffer is cached under is never the size its caller states
// when giving it back (flacamd_host_free is public, and a wrong size there must not turn into a short buffer later)
static std::unordered_map<void*, size_t> g_pin_live;

static void pin_cache_flush_locked()
{
	for(int i = 0; i < PIN_CACHE_ENTRIES; i++) if(g_pin[i].p) { (void)hipHostFree(g_pin[i].p); g_pin[i].p = 0; g_pin[i].bytes = 0; }
	g_pin_total = 0;
}

static hipError_t pinned_alloc(void **out, size_t bytes)
{
	std::lock_guard<std::mutex> hold(g_pin_lock);
	for(int i = 0; i < PIN_CACHE_ENTRIES; i++) if(g_pin[i].p && g_pin[i].bytes == bytes) {
		*out = g_pin[i].p; g_pin[i].p = 0; g_pin_total -= bytes; g_pin_live[*out] = bytes; return hipSuccess;
	}
	hipError_t he = hipHostMalloc(out, bytes, hipHostMallocPortable);
	if(he != hipSuccess) {   // give the idle buffers back and try once more before reporting failure
		(void)hipGetLastError();
		pin_cache_flush_locked();
		he = hipHostMalloc(out, bytes, hipHostMallocPortable);
	}
	if(he == hipSuccess) g_pin_live[*out] = bytes;
	return he;
}

static void pinned_free(void *p, size_t /*stated_bytes*/)
{
	if(!p) return;
	{
		std::lock_guard<std::mutex> hold(g_pin_lock);
		const auto it = g_pin_live.find(p);
		if(it != g_pin_live.end()) {
			const size_t bytes = it->second;
			g_pin_live.erase(it);
			if(g_pin_total + bytes <= PIN_CACHE_BYTES) for(int i = 0; i < PIN_CACHE_ENTRIES; i++) if(!g_pin[i].p) { g_pin[i].p = p; g_pin[i].bytes = bytes; g_pin_total += bytes; return; }
		}
	}
	(void)hipHostFree(p);   // not cached (cache full, or not one of ours: then hipHostFree reports it and nothing else happens)
}

static void pipe_trim_parked();
extern "C" void flacamd_pipe_trim(void)
{
	pipe_trim_parked();
	std::lock_guard<std::mutex> hold(g_pin_lock);
	pin_cache_flush_locked();
}

struct PipeSlot {
	uint32_t lane;
	void *h_pcm, *d_pcm;
	uint8_t *d_out, *h_out; uint32_t *d_fb, *h_fb;
	PipeMeta *h_meta;
	hipEvent_t ev_in, ev_k, ev_done;
	FaCfg cfg; uint64_t nsamples;
	uint8_t *big; size_t big_cap;   // host copy of a batch holding frames beyond the usual size bound
	int collected;
};

extern "C" void *flacamd_host_alloc(size_t bytes)
{
	void *p = 0;
	if(bytes == 0 || pinned_alloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); return 0; }
	return p;
}
extern "C" void flacamd_host_free(void *p, size_t bytes) { if(p) pinned_free(p, bytes); }

struct PipeLane { int device; flacamd_encoder *enc; hipStream_t hs, cs, ds; };

struct flacamd_pipe {
	flacamd_params p;
	int pcm_format; uint32_t batch_frames, nlanes, nslots;
	size_t pcm_bytes, out_cap;
	PipeLane *lanes; PipeSlot *slots;
	uint64_t head, tail;            // batches submitted / released so far; slot of batch q is q % nslots
	flacamd_params key_params; int key_devices[64]; uint32_t key_batch, key_depth;   // what it was created with
	int complete;
};

extern "C" void flacamd_pcm_narrow(int16_t *dst, const int32_t *src, size_t nvalues);

// A pipe that is destroyed while idle is parked instead (PIPE_PARKED at most) and handed to the next
// flacamd_pipe_create with the same devices, parameters, batch size and depth: a process that codes stream after
// stream then pays for streams, device memory, pinned memory and the encoder workspaces once.
#define PIPE_PARKED 2
static flacamd_pipe *g_parked[PIPE_PARKED];
static void pipe_free(flacamd_pipe *pp);

extern "C" void flacamd_pipe_destroy(flacamd_pipe *pp)
{
	if(!pp) return;
	if(pp->complete && pp->head == pp->tail) {
		std::lock_guard<std::mutex> hold(g_pin_lock);
		for(int i = 0; i < PIPE_PARKED; i++) if(!g_parked[i]) { g_parked[i] = pp; return; }
	}
	pipe_free(pp);
}

static void pipe_free(flacamd_pipe *pp)
{
	for(uint32_t i = 0; pp->slots && i < pp->nslots; i++) {
		PipeSlot &s = pp->slots[i];
		if(pp->lanes && pp->lanes[s.lane].enc) (void)hipSetDevice(pp->lanes[s.lane].device);
		if(s.ev_done) { (void)hipEventSynchronize(s.ev_done); }
		pinned_free(s.h_pcm, pp->pcm_bytes);
		pinned_free(s.h_out, pp->out_cap);
		if(s.h_fb) (void)hipHostFree(s.h_fb);
		if(s.h_meta) (void)hipHostFree(s.h_meta);
		if(s.d_pcm) (void)hipFree(s.d_pcm);
		if(s.d_out) (void)hipFree(s.d_out);
		if(s.d_fb) (void)hipFree(s.d_fb);
		if(s.ev_in) (void)hipEventDestroy(s.ev_in);
		if(s.ev_k) (void)hipEventDestroy(s.ev_k);
		if(s.ev_done) (void)hipEventDestroy(s.ev_done);
		free(s.big);
	}
	for(uint32_t l = 0; pp->lanes && l < pp->nlanes; l++) {
		PipeLane &ln = pp->lanes[l];
		if(!ln.enc) continue;
		(void)hipSetDevice(ln.device);
		if(ln.cs) { (void)hipStreamSynchronize(ln.cs); (void)hipStreamDestroy(ln.cs); }
		if(ln.hs) (void)hipStreamDestroy(ln.hs);
		if(ln.ds) (void)hipStreamDestroy(ln.ds);
		flacamd_encoder_destroy(ln.enc);
	}
	free(pp->slots); free(pp->lanes); free(pp);
}

static int pipe_create_once(flacamd_pipe **out, const int *devices, uint32_t ndevices, const flacamd_params *params, uint32_t batch_frames, uint32_t depth);

extern "C" int flacamd_pipe_create(flacamd_pipe **out, const int *devices, uint32_t ndevices, const flacamd_params *params, uint32_t batch_frames, uint32_t depth)
{
	int rc = pipe_create_once(out, devices, ndevices, params, batch_frames, depth);
	if(rc == FLACAMD_E_OUT_OF_MEMORY) { (void)hipGetLastError(); flacamd_pipe_trim(); rc = pipe_create_once(out, devices, ndevices, params, batch_frames, depth); }   // parked pipes and cached slots given back first
	return rc;
}

static int pipe_create_once(flacamd_pipe **out, const int *devices, uint32_t ndevices, const flacamd_params *params, uint32_t batch_frames, uint32_t depth)
{
	if(!out || !devices || ndevices == 0 || ndevices > 64 || !params || batch_frames == 0) return FLACAMD_E_INVALID_ARGUMENT;
	*out = 0;
	if(depth < 2) depth = 2;
	if(depth > 8) depth = 8;
	{
		std::lock_guard<std::mutex> hold(g_pin_lock);
		for(int i = 0; i < PIPE_PARKED; i++) {
			flacamd_pipe *q = g_parked[i];
			if(q && q->nlanes == ndevices && q->key_batch == batch_frames && q->key_depth == depth && 0 == memcmp(&q->key_params, params, sizeof *params)
				&& 0 == memcmp(q->key_devices, devices, sizeof(int) * ndevices)) {
				// idle (nothing in flight), and as good as new for its next owner: batch 0 goes to slot 0 again, which is what
				// flacamd_pipe_slot_buffer promises and what a caller that reads its own input back (the MD5 thread) relies on
				q->head = q->tail = 0;
				g_parked[i] = 0; *out = q; return FLACAMD_OK;
			}
		}
	}
	flacamd_pipe *pp = (flacamd_pipe*)calloc(1, sizeof *pp);
	if(!pp) return FLACAMD_E_OUT_OF_MEMORY;
	pp->key_params = *params; memcpy(pp->key_devices, devices, sizeof(int) * ndevices); pp->key_batch = batch_frames; pp->key_depth = depth;
	pp->p = *params;
	int rc = flacamd_params_resolve(&pp->p);
	if(rc != FLACAMD_OK) { free(pp); return rc; }
	if(pp->p.channels == 2 && pp->p.do_mid_side && pp->p.loose_mid_side) {   // batch borders on decision period borders
		unsigned lp = (unsigned)((double)pp->p.sample_rate * 0.4 / (double)pp->p.blocksize + 0.5);
		if(lp == 0) lp = 1;
		batch_frames = (batch_frames + lp - 1) / lp * lp;
	}
	pp->pcm_format = pp->p.bits_per_sample <= 16 ? FLACAMD_PCM_S16 : FLACAMD_PCM_S32;
	pp->batch_frames = batch_frames; pp->nlanes = ndevices; pp->nslots = ndevices * depth;
	pp->lanes = (PipeLane*)calloc(pp->nlanes, sizeof(PipeLane));
	pp->slots = (PipeSlot*)calloc(pp->nslots, sizeof(PipeSlot));
	if(!pp->lanes || !pp->slots) { flacamd_pipe_destroy(pp); return FLACAMD_E_OUT_OF_MEMORY; }
	pp->pcm_bytes = (size_t)batch_frames * pp->p.blocksize * pp->p.channels * (pp->pcm_format == FLACAMD_PCM_S16 ? 2u : 4u);
	for(uint32_t l = 0; l < pp->nlanes; l++) {
		PipeLane &ln = pp->lanes[l];
		ln.device = devices[l];
		rc = flacamd_encoder_create(&ln.enc, ln.device, &pp->p, batch_frames);
		if(rc != FLACAMD_OK) { ln.enc = 0; flacamd_pipe_destroy(pp); return rc; }
		hipError_t he = hipStreamCreateWithFlags(&ln.hs, hipStreamNonBlocking);
		if(he == hipSuccess) he = hipStreamCreateWithFlags(&ln.cs, hipStreamNonBlocking);
		if(he == hipSuccess) he = hipStreamCreateWithFlags(&ln.ds, hipStreamNonBlocking);
		if(he != hipSuccess) { rc = hip_fail(he, "hipStreamCreate(pipe)"); flacamd_pipe_destroy(pp); return rc; }
	}
	pp->out_cap = flacamd_encoder_max_output_bytes(pp->lanes[0].enc, batch_frames);
	for(uint32_t i = 0; i < pp->nslots; i++) {
		PipeSlot &s = pp->slots[i];
		s.lane = i % pp->nlanes;
		hipError_t he = hipSetDevice(pp->lanes[s.lane].device);
		if(he == hipSuccess) he = pinned_alloc(&s.h_pcm, pp->pcm_bytes);
		if(he == hipSuccess) he = pinned_alloc((void**)&s.h_out, pp->out_cap);
		if(he == hipSuccess) he = hipHostMalloc((void**)&s.h_fb, (size_t)batch_frames * 4u, hipHostMallocPortable);
		if(he == hipSuccess) he = hipHostMalloc((void**)&s.h_meta, sizeof(PipeMeta), hipHostMallocPortable);
		if(he == hipSuccess) he = hipMalloc(&s.d_pcm, pp->pcm_bytes);
		if(he == hipSuccess) he = hipMalloc((void**)&s.d_out, pp->out_cap);
		if(he == hipSuccess) he = hipMalloc((void**)&s.d_fb, (size_t)batch_frames * 4u);
		if(he == hipSuccess) he = hipEventCreateWithFlags(&s.ev_in, hipEventDisableTiming);
		if(he == hipSuccess) he = hipEventCreateWithFlags(&s.ev_k, hipEventDisableTiming);
		if(he == hipSuccess) he = hipEventCreateWithFlags(&s.ev_done, hipEventDisableTiming);
		if(he != hipSuccess) { rc = hip_fail(he, "pipe slot allocation"); flacamd_pipe_destroy(pp); return rc; }
	}
	pp->complete = 1;
	*out = pp;
	return FLACAMD_OK;
}

static void pipe_trim_parked()
{
	flacamd_pipe *take[PIPE_PARKED];
	{
		std::lock_guard<std::mutex> hold(g_pin_lock);
		for(int i = 0; i < PIPE_PARKED; i++) { take[i] = g_parked[i]; g_parked[i] = 0; }
	}
	for(int i = 0; i < PIPE_PARKED; i++) if(take[i]) pipe_free(take[i]);
}

extern "C" int flacamd_pipe_pcm_format(const flacamd_pipe *pp) { return pp ? pp->pcm_format : FLACAMD_E_INVALID_ARGUMENT; }
extern "C" uint32_t flacamd_pipe_batch_frames(const flacamd_pipe *pp) { return pp ? pp->batch_frames : 0; }
extern "C" uint32_t flacamd_pipe_slots(const flacamd_pipe *pp) { return pp ? pp->nslots : 0; }
extern "C" uint32_t flacamd_pipe_in_flight(const flacamd_pipe *pp) { return pp ? (uint32_t)(pp->head - pp->tail) : 0; }
extern "C" const flacamd_params *flacamd_pipe_params(const flacamd_pipe *pp) { return pp ? &pp->p : 0; }

extern "C" size_t flacamd_pipe_max_batch_bytes(const flacamd_pipe *pp) { return pp ? pp->out_cap : 0; }
extern "C" void *flacamd_pipe_slot_buffer(flacamd_pipe *pp, uint32_t slot) { return pp && slot < pp->nslots ? pp->slots[slot].h_pcm : 0; }

extern "C" void *flacamd_pipe_fill_buffer(flacamd_pipe *pp)
{
	if(!pp || pp->head - pp->tail >= pp->nslots) return 0;
	return pp->slots[pp->head % pp->nslots].h_pcm;
}

extern "C" int flacamd_pipe_submit(flacamd_pipe *pp, uint64_t nsamples, int final_block, uint32_t first_frame_number)
{
	if(!pp || nsamples == 0) return FLACAMD_E_INVALID_ARGUMENT;
	if(pp->head - pp->tail >= pp->nslots) return FLACAMD_E_CAPACITY;
	const uint32_t N = pp->p.blocksize;
	const uint64_t F64 = (nsamples + N - 1) / N;
	if(F64 > pp->batch_frames) return FLACAMD_E_CAPACITY;
	if(!final_block && nsamples % N != 0) return FLACAMD_E_INVALID_ARGUMENT;
	PipeSlot &s = pp->slots[pp->head % pp->nslots];
	PipeLane &ln = pp->lanes[s.lane];
	flacamd_encoder *e = ln.enc;
	HIP_TRY(hipSetDevice(ln.device));
	FaCfg c = e->cfg;
	c.pcm_format = (uint32_t)pp->pcm_format;
	c.F = (uint32_t)F64; c.G = c.F * c.nsig;
	c.last_n = (uint32_t)(nsamples - (F64 - 1) * N);
	c.last_fractional = (final_block && c.last_n != N) ? 1u : 0u;
	c.first_frame = first_frame_number; c.prev_assignment = 0;
	c.multi = 0; c.skip_full = 0;
	s.cfg = c; s.nsamples = nsamples; s.collected = 0;
	const size_t in_bytes = (size_t)nsamples * pp->p.channels * (pp->pcm_format == FLACAMD_PCM_S16 ? 2u : 4u);
	HIP_TRY(hipMemcpyAsync(s.d_pcm, s.h_pcm, in_bytes, hipMemcpyHostToDevice, ln.hs));
	HIP_TRY(hipEventRecord(s.ev_in, ln.hs));
	HIP_TRY(hipStreamWaitEvent(ln.cs, s.ev_in, 0));
	const int rc = enqueue_batch(e, c, s.d_pcm, s.d_out, pp->out_cap, s.d_fb, ln.cs);
	if(rc != FLACAMD_OK) return rc;
	HIP_TRY(hipMemcpyAsync(&s.h_meta->status, e->work.status, 4, hipMemcpyDeviceToHost, ln.cs));
	HIP_TRY(hipMemcpyAsync(&s.h_meta->total, e->work.offsets + c.F, 8, hipMemcpyDeviceToHost, ln.cs));
	HIP_TRY(hipMemcpyAsync(&s.h_meta->last_assignment, e->work.assignment + (c.F - 1), 4, hipMemcpyDeviceToHost, ln.cs));
	HIP_TRY(hipMemcpyAsync(s.h_fb, s.d_fb, (size_t)c.F * 4u, hipMemcpyDeviceToHost, ln.cs));
	HIP_TRY(hipEventRecord(s.ev_k, ln.cs));
	// the frames: their total is only known on the device, so the copy takes the batch's size bound (a little more
	// than the PCM it came from); it runs on the copy engine in the direction the uploads leave idle
	size_t out_bytes = (size_t)c.F * c.slot_bytes;
	if(out_bytes > pp->out_cap) out_bytes = pp->out_cap;
	HIP_TRY(hipStreamWaitEvent(ln.ds, s.ev_k, 0));
	HIP_TRY(hipMemcpyAsync(s.h_out, s.d_out, out_bytes, hipMemcpyDeviceToHost, ln.ds));
	HIP_TRY(hipEventRecord(s.ev_done, ln.ds));
	pp->head++;
	return FLACAMD_OK;
}

// 1 when the oldest batch in flight has finished (collect would not wait), 0 when not, <0 on error
extern "C" int flacamd_pipe_ready(flacamd_pipe *pp)
{
	if(!pp || pp->head == pp->tail) return FLACAMD_E_INVALID_ARGUMENT;
	PipeSlot &s = pp->slots[pp->tail % pp->nslots];
	if(s.collected) return 1;
	const hipError_t he = hipEventQuery(s.ev_done);
	if(he == hipSuccess) return 1;
	if(he == hipErrorNotReady) { (void)hipGetLastError(); return 0; }
	return hip_fail(he, "hipEventQuery");
}

extern "C" int flacamd_pipe_collect(flacamd_pipe *pp, const uint8_t **h_frames, const uint32_t **h_frame_bytes, uint32_t *nframes,
	uint64_t *total_bytes, const void **h_pcm, uint64_t *nsamples, uint32_t *first_frame_number)
{
	if(!pp || pp->head == pp->tail) return FLACAMD_E_INVALID_ARGUMENT;
	PipeSlot &s = pp->slots[pp->tail % pp->nslots];
	PipeLane &ln = pp->lanes[s.lane];
	HIP_TRY(hipSetDevice(ln.device));
	const uint8_t *frames = s.h_out;
	if(!s.collected) {
		HIP_TRY(hipEventSynchronize(s.ev_done));
		if(s.h_meta->status & 1u) {
			// a frame beyond its slot (see repack_long): this batch once more, synchronously, with the buffers it takes.  The
			// batches submitted after it have used the workspace in stream order; wait for them before touching it again.
			HIP_TRY(hipStreamSynchronize(ln.cs));
			uint64_t total = 0; void *tmp = 0;
			int rc = run_batch(ln.enc, s.cfg, s.d_pcm, s.d_out, pp->out_cap, s.d_fb, &total, ln.cs);
			const void *src = s.d_out;
			if(rc == FLACAMD_E_CAPACITY && total > pp->out_cap) {
				hipError_t he = hipMalloc(&tmp, total);
				if(he != hipSuccess) return hip_fail(he, "hipMalloc(long frames)");
				rc = run_batch(ln.enc, s.cfg, s.d_pcm, tmp, total, s.d_fb, &total, ln.cs);
				src = tmp;
			}
			if(rc == FLACAMD_OK && s.big_cap < total) { free(s.big); s.big = (uint8_t*)malloc(total); s.big_cap = s.big ? total : 0; if(!s.big) rc = FLACAMD_E_OUT_OF_MEMORY; }
			hipError_t he = hipSuccess;
			if(rc == FLACAMD_OK) he = hipMemcpy(s.big, src, total, hipMemcpyDeviceToHost);
			if(rc == FLACAMD_OK && he == hipSuccess) he = hipMemcpy(s.h_fb, s.d_fb, (size_t)s.cfg.F * 4u, hipMemcpyDeviceToHost);
			if(rc == FLACAMD_OK && he == hipSuccess) he = hipMemcpy(&s.h_meta->last_assignment, ln.enc->work.assignment + (s.cfg.F - 1), 4, hipMemcpyDeviceToHost);
			if(tmp) (void)hipFree(tmp);
			if(rc != FLACAMD_OK) return rc;
			if(he != hipSuccess) return hip_fail(he, "long frames readback");
			s.h_meta->total = total; s.h_meta->status = 4u;   // frames are in s.big
		}
		else if(s.h_meta->status & 2u) return FLACAMD_E_CAPACITY;
		s.collected = 1;
	}
	if(s.h_meta->status & 4u) frames = s.big;
	if(h_frames) *h_frames = frames;
	if(h_frame_bytes) *h_frame_bytes = s.h_fb;
	if(nframes) *nframes = s.cfg.F;
	if(total_bytes) *total_bytes = s.h_meta->total;
	if(h_pcm) *h_pcm = s.h_pcm;
	if(nsamples) *nsamples = s.nsamples;
	if(first_frame_number) *first_frame_number = s.cfg.first_frame;
	return FLACAMD_OK;
}

extern "C" int flacamd_pipe_release(flacamd_pipe *pp)
{
	if(!pp || pp->head == pp->tail || !pp->slots[pp->tail % pp->nslots].collected) return FLACAMD_E_INVALID_ARGUMENT;
	pp->slots[pp->tail % pp->nslots].collected = 0;
	pp->tail++;
	return FLACAMD_OK;
}

// int32 samples that fit 16 bits -> int16 (the upload of a 16-bit stream then moves half the bytes)
extern "C" void flacamd_pcm_narrow(int16_t *dst, const int32_t *src, size_t nvalues)
{
	size_t i = 0;
#if defined(__SSE2__)
	for(; i + 16 <= nvalues; i += 16) {
		const __m128i a = _mm_loadu_si128((const __m128i*)(src + i)), b = _mm_loadu_si128((const __m128i*)(src + i + 4));
		const __m128i c = _mm_loadu_si128((const __m128i*)(src + i + 8)), d = _mm_loadu_si128((const __m128i*)(src + i + 12));
		// 16-bit truncation, not saturation: what a cast does (shift the low halves up, arithmetic shift back, pack)
		_mm_storeu_si128((__m128i*)(dst + i), _mm_packs_epi32(_mm_srai_epi32(_mm_slli_epi32(a, 16), 16), _mm_srai_epi32(_mm_slli_epi32(b, 16), 16)));
		_mm_storeu_si128((__m128i*)(dst + i + 8), _mm_packs_epi32(_mm_srai_epi32(_mm_slli_epi32(c, 16), 16), _mm_srai_epi32(_mm_slli_epi32(d, 16), 16)));
	}
#endif
	for(; i < nvalues; i++) dst[i] = (int16_t)src[i];
}

// A whole stream (or a run of whole blocks of one) from host memory through the pipe: the frames come out back to
// back in h_out, one size per frame in h_frame_bytes.
extern "C" int flacamd_pipe_encode_host(flacamd_pipe *pp, const void *h_pcm, int pcm_format, uint64_t nsamples, int final_block, uint32_t first_frame_number,
	uint8_t *h_out, size_t out_cap, uint32_t *h_frame_bytes, size_t frame_cap, uint64_t *h_total_bytes, uint32_t *h_nframes)
{
	if(!pp || !h_pcm || !h_out || nsamples == 0 || pp->head != pp->tail) return FLACAMD_E_INVALID_ARGUMENT;
	if(pcm_format != FLACAMD_PCM_S32 && pcm_format != FLACAMD_PCM_S16) return FLACAMD_E_INVALID_ARGUMENT;
	if(pcm_format == FLACAMD_PCM_S16 && pp->pcm_format != FLACAMD_PCM_S16) return FLACAMD_E_INVALID_ARGUMENT;
	const uint32_t N = pp->p.blocksize, ch = pp->p.channels;
	if(!final_block && nsamples % N != 0) return FLACAMD_E_INVALID_ARGUMENT;
	const uint64_t batch = (uint64_t)pp->batch_frames * N;
	uint64_t fed = 0, total = 0; size_t nf = 0; int rc = FLACAMD_OK; int short_out = 0;
	auto drain_one = [&]() -> int {
		const uint8_t *fr; const uint32_t *fb; uint32_t n; uint64_t tb;
		int r = flacamd_pipe_collect(pp, &fr, &fb, &n, &tb, 0, 0, 0);
		if(r != FLACAMD_OK) return r;
		if(total + tb <= out_cap && nf + n <= frame_cap && !short_out) {
			memcpy(h_out + total, fr, tb);
			if(h_frame_bytes) memcpy(h_frame_bytes + nf, fb, (size_t)n * 4u);
		}
		else short_out = 1;
		total += tb; nf += n;
		return flacamd_pipe_release(pp);
	};
	while(fed < nsamples && rc == FLACAMD_OK) {
		void *dst = flacamd_pipe_fill_buffer(pp);
		if(!dst) { rc = drain_one(); continue; }
		const uint64_t n = nsamples - fed < batch ? nsamples - fed : batch;
		const size_t in_wide = (size_t)ch * (pcm_format == FLACAMD_PCM_S16 ? 2u : 4u);
		if(pcm_format == pp->pcm_format) memcpy(dst, (const uint8_t*)h_pcm + fed * in_wide, (size_t)n * in_wide);
		else flacamd_pcm_narrow((int16_t*)dst, (const int32_t*)h_pcm + fed * ch, (size_t)n * ch);
		rc = flacamd_pipe_submit(pp, n, final_block && fed + n == nsamples, first_frame_number + (uint32_t)(fed / N));
		fed += n;
	}
	while(pp->head != pp->tail) {
		const int r = drain_one();
		if(r != FLACAMD_OK) { if(rc == FLACAMD_OK) rc = r; // give the remaining slots back without their output
			while(pp->head != pp->tail) { PipeSlot &s = pp->slots[pp->tail % pp->nslots]; (void)hipSetDevice(pp->lanes[s.lane].device); (void)hipEventSynchronize(s.ev_done); s.collected = 0; pp->tail++; } }
	}
	if(h_total_bytes) *h_total_bytes = total;
	if(h_nframes) *h_nframes = (uint32_t)nf;
	if(rc == FLACAMD_OK && short_out) rc = FLACAMD_E_CAPACITY;
	return rc;
}

// ---- md5 entry point is implemented in k_md5.hip
