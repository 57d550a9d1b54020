// Host-side ceilings of the streaming API: how fast PCM handed over in pageable memory can reach the GPU.
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/host_copy_rates tools/host_copy_rates.hip -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <emmintrin.h>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void conv(const int32_t *s, int16_t *d, size_t n) {
	size_t i = 0;
	for(; i + 8 <= n; i += 8) { __m128i a = _mm_loadu_si128((const __m128i*)(s + i)), b = _mm_loadu_si128((const __m128i*)(s + i + 4)); _mm_storeu_si128((__m128i*)(d + i), _mm_packs_epi32(a, b)); }
	for(; i < n; i++) d[i] = (int16_t)s[i];
}
int main() {
	const size_t wide = 64u << 20;   // 64 M wide stereo samples: 512 MB int32
	const size_t n = wide * 2;
	int32_t *src = (int32_t*)malloc(n * 4);
	for(size_t i = 0; i < n; i++) src[i] = (int32_t)(i * 2654435761u) >> 16;
	void *d = 0; hipMalloc(&d, n * 4);
	void *pin = 0; hipHostMalloc(&pin, n * 4, hipHostMallocDefault);
	hipStream_t st; hipStreamCreate(&st);
	for(int rep = 0; rep < 2; rep++) {
		double t = now(); hipMemcpy(d, src, n * 4, hipMemcpyHostToDevice); double dt = now() - t;
		printf("pageable hipMemcpy H2D int32: %.2f GB/s = %.0f M wide/s\n", n * 4 / dt / 1e9, wide / dt / 1e6);
		t = now(); memcpy(pin, src, n * 4); dt = now() - t;
		printf("memcpy -> pinned, 1 thread: %.2f GB/s = %.0f M wide/s\n", n * 4 / dt / 1e9, wide / dt / 1e6);
		t = now(); hipMemcpyAsync(d, pin, n * 4, hipMemcpyHostToDevice, st); hipStreamSynchronize(st); dt = now() - t;
		printf("pinned H2D int32: %.2f GB/s = %.0f M wide/s\n", n * 4 / dt / 1e9, wide / dt / 1e6);
		t = now(); hipMemcpyAsync(pin, d, n * 2, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st); dt = now() - t;
		printf("pinned D2H: %.2f GB/s\n", n * 2 / dt / 1e9);
		t = now(); conv(src, (int16_t*)pin, n); dt = now() - t;
		printf("int32->int16 into pinned, 1 thread: %.2f GB/s read = %.0f M wide/s\n", n * 4 / dt / 1e9, wide / dt / 1e6);
		for(int nt : {2, 4, 8}) {
			t = now();
			std::vector<std::thread> th;
			for(int k = 0; k < nt; k++) th.emplace_back([=] { size_t a = n / nt * k, b = k + 1 == nt ? n : n / nt * (k + 1); conv(src + a, (int16_t*)pin + a, b - a); });
			for(auto &x : th) x.join();
			dt = now() - t;
			printf("int32->int16 into pinned, %d threads: %.2f GB/s read = %.0f M wide/s\n", nt, n * 4 / dt / 1e9, wide / dt / 1e6);
		}
		// chunked pageable copies as the object would issue them (4 MB chunks)
		t = now();
		for(size_t o = 0; o < n * 4; o += 4u << 20) hipMemcpyAsync((char*)d + o, (char*)src + o, 4u << 20, hipMemcpyHostToDevice, st);
		hipStreamSynchronize(st); dt = now() - t;
		printf("pageable hipMemcpyAsync 4 MB chunks: %.2f GB/s = %.0f M wide/s\n", n * 4 / dt / 1e9, wide / dt / 1e6);
	}
	return 0;
}
