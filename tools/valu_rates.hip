// Micro-benchmark: issue rate of the integer VALU instructions the search kernel could use.
// Each kernel runs 8 independent chains of one instruction, 4096 iterations x 8, on
// 256 CUs x 8 waves/SIMD.  Prints wave-instructions per cycle-equivalent (ns per instr per SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

#define ITERS 2048
#define DEF_KERNEL(NAME, ASM) \
__global__ void __launch_bounds__(256) NAME(int *out, int a, int b) { \
	int x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
	int va = a + threadIdx.x, vb = b ^ threadIdx.x; \
	for(int i = 0; i < ITERS; i++) { \
		asm volatile(ASM(%0) "\n" ASM(%1) "\n" ASM(%2) "\n" ASM(%3) "\n" ASM(%4) "\n" ASM(%5) "\n" ASM(%6) "\n" ASM(%7) \
			: "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(va), "v"(vb)); \
	} \
	out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7; \
}
#define A_MAD24(x) "v_mad_i32_i24 " #x ", %8, %9, " #x
#define A_MUL24(x) "v_mul_i32_i24 " #x ", %8, " #x
#define A_ADD(x) "v_add_u32 " #x ", %8, " #x
#define A_ADD3(x) "v_add3_u32 " #x ", %8, %9, " #x
#define A_MULLO(x) "v_mul_lo_u32 " #x ", %8, " #x
#define A_DOT2(x) "v_dot2_i32_i16 " #x ", %8, %9, " #x
#define A_DOT4(x) "v_dot4_i32_i8 " #x ", %8, %9, " #x
#define A_SAD(x) "v_sad_u32 " #x ", %8, %9, " #x
#define A_MAXI(x) "v_max_i32 " #x ", %8, " #x
#define A_FMA(x) "v_fma_f32 " #x ", %8, %9, " #x
#define A_MADI16(x) "v_mad_i32_i16 " #x ", %8, %9, " #x
#define A_PKMAD(x) "v_pk_mad_i16 " #x ", %8, %9, " #x
#define A_ASHR(x) "v_ashrrev_i32 " #x ", 3, " #x
#define A_MADU24(x) "v_mad_u32_u24 " #x ", %8, %9, " #x
#define A_LSHLADD(x) "v_lshl_add_u32 " #x ", %8, 2, " #x
#define A_MULF(x) "v_mul_f32 " #x ", %8, " #x
#define A_PKFMA(x) "v_pk_fma_f32 " #x ", %8, %9, " #x
#define A_ALIGNBIT(x) "v_alignbit_b32 " #x ", %8, " #x ", 7"
#define A_PERM(x) "v_perm_b32 " #x ", %8, " #x ", %9"
#define A_BFE(x) "v_bfe_i32 " #x ", " #x ", 3, 16"
#define A_MOV(x) "v_mov_b32 " #x ", %8"
#define A_XOR(x) "v_xor_b32 " #x ", %8, " #x
DEF_KERNEL(k_alignbit, A_ALIGNBIT)
DEF_KERNEL(k_perm, A_PERM)
DEF_KERNEL(k_bfe, A_BFE)
DEF_KERNEL(k_mov, A_MOV)
DEF_KERNEL(k_xor, A_XOR)
DEF_KERNEL(k_mad24, A_MAD24)
DEF_KERNEL(k_mul24, A_MUL24)
DEF_KERNEL(k_add, A_ADD)
DEF_KERNEL(k_add3, A_ADD3)
DEF_KERNEL(k_mullo, A_MULLO)
DEF_KERNEL(k_dot2, A_DOT2)
DEF_KERNEL(k_dot4, A_DOT4)
DEF_KERNEL(k_sad, A_SAD)
DEF_KERNEL(k_maxi, A_MAXI)
DEF_KERNEL(k_fma, A_FMA)
DEF_KERNEL(k_madi16, A_MADI16)
DEF_KERNEL(k_pkmad, A_PKMAD)
DEF_KERNEL(k_ashr, A_ASHR)
DEF_KERNEL(k_madu24, A_MADU24)
DEF_KERNEL(k_lshladd, A_LSHLADD)
DEF_KERNEL(k_mulf, A_MULF)

__global__ void __launch_bounds__(256) k_mad64(long long *out, int a, int b) {
	long long x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
	int va = a + threadIdx.x, vb = b ^ threadIdx.x;
	for(int i = 0; i < ITERS; i++) {
		asm volatile("v_mad_i64_i32 %0, vcc, %4, %5, %0\nv_mad_i64_i32 %1, vcc, %4, %5, %1\nv_mad_i64_i32 %2, vcc, %4, %5, %2\nv_mad_i64_i32 %3, vcc, %4, %5, %3\n"
			"v_mad_i64_i32 %0, vcc, %4, %5, %0\nv_mad_i64_i32 %1, vcc, %4, %5, %1\nv_mad_i64_i32 %2, vcc, %4, %5, %2\nv_mad_i64_i32 %3, vcc, %4, %5, %3"
			: "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(va), "v"(vb) : "vcc");
	}
	out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3;
}

// one dependent chain: every instruction takes the result of the one before it
#define DEF_KERNEL_DEP(NAME, ASM) \
__global__ void __launch_bounds__(256) NAME(int *out, int a, int b) { \
	int x0 = threadIdx.x; \
	int va = a + threadIdx.x, vb = b ^ threadIdx.x; \
	for(int i = 0; i < ITERS; i++) { \
		asm volatile(ASM(%0) "\n" ASM(%0) "\n" ASM(%0) "\n" ASM(%0) "\n" ASM(%0) "\n" ASM(%0) "\n" ASM(%0) "\n" ASM(%0) \
			: "+v"(x0) : "v"(va), "v"(vb)); \
	} \
	out[blockIdx.x * 256 + threadIdx.x] = x0; \
}
#define D_MAD24(x) "v_mad_i32_i24 " #x ", %1, %2, " #x
#define D_ADD(x) "v_add_u32 " #x ", %1, " #x
DEF_KERNEL_DEP(k_mad24_dep, D_MAD24)
DEF_KERNEL_DEP(k_add_dep, D_ADD)

// 64-bit and packed-float instructions: four independent chains on register pairs
#define DEF_KERNEL64(NAME, ASM) \
__global__ void __launch_bounds__(256) NAME(long long *out, int a, int b) { \
	long long x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3; \
	long long va = a + threadIdx.x; int vb = (b ^ threadIdx.x) & 15; \
	for(int i = 0; i < ITERS; i++) { \
		asm volatile(ASM(%0) "\n" ASM(%1) "\n" ASM(%2) "\n" ASM(%3) "\n" ASM(%0) "\n" ASM(%1) "\n" ASM(%2) "\n" ASM(%3) \
			: "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(va), "v"(vb)); \
	} \
	out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3; \
}
#define A_ASHR64(x) "v_ashrrev_i64 " #x ", %5, " #x
#define A_LSHLADD64(x) "v_lshl_add_u64 " #x ", " #x ", 0, %4"
#define A_PKMULF(x) "v_pk_mul_f32 " #x ", %4, " #x
#define A_PKADDF(x) "v_pk_add_f32 " #x ", %4, " #x
#define A_PKFMAF(x) "v_pk_fma_f32 " #x ", %4, %4, " #x
#define A_FMA64(x) "v_fma_f64 " #x ", %4, %4, " #x
DEF_KERNEL64(k_ashr64, A_ASHR64)
DEF_KERNEL64(k_lshladd64, A_LSHLADD64)
DEF_KERNEL64(k_pkmulf, A_PKMULF)
DEF_KERNEL64(k_pkaddf, A_PKADDF)
DEF_KERNEL64(k_pkfmaf, A_PKFMAF)
DEF_KERNEL64(k_fma64, A_FMA64)

// The shader clock while every SIMD is busy with the multiply-add loop: the shader clock counter (s_memtime) against the
// constant 100 MHz counter (s_memrealtime) over the same stretch, read by the first lane of every workgroup.  The "cycles"
// column below converts with this measured frequency, not with a nominal one.
__global__ void __launch_bounds__(256) k_clock(int *out, int a, int b, unsigned long long *ticks)
{
	int x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
	int va = a + threadIdx.x, vb = b ^ threadIdx.x;
	const unsigned long long c0 = clock64(), r0 = wall_clock64();
	for(int i = 0; i < ITERS * 4; i++) {
		asm volatile(A_MAD24(%0) "\n" A_MAD24(%1) "\n" A_MAD24(%2) "\n" A_MAD24(%3) "\n" A_MAD24(%4) "\n" A_MAD24(%5) "\n" A_MAD24(%6) "\n" A_MAD24(%7)
			: "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(va), "v"(vb));
	}
	const unsigned long long c1 = clock64(), r1 = wall_clock64();
	out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
	if(threadIdx.x == 0) { ticks[2 * blockIdx.x] = c1 - c0; ticks[2 * blockIdx.x + 1] = r1 - r0; }
}

static double g_ghz = 2.4;

template<typename K, typename T>
static void run(const char *name, K kern, T *buf)
{
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	const int blocks = 256 * 8; // 8 blocks of 4 waves per CU -> 8 waves per SIMD
	hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, buf, 3, 5);
	hipDeviceSynchronize();
	hipEventRecord(e0, 0);
	hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, buf, 3, 5);
	hipEventRecord(e1, 0);
	hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	// wave-instructions per SIMD: 8 waves x ITERS x 8
	const double per_simd = 8.0 * ITERS * 8.0;
	printf("%-12s %8.3f ms  %6.2f ns per wave-instr per SIMD  (= %5.2f cycles at the measured %.3f GHz)\n", name, ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * g_ghz, g_ghz);
}

// the same loops with `waves` waves on every SIMD: what a wave gets when it has the SIMD (nearly) to itself
template<typename K>
static void run_few(const char *name, K kern, int *buf, int waves)
{
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	const int blocks = 256 * waves;
	hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, buf, 3, 5);
	hipDeviceSynchronize();
	hipEventRecord(e0, 0);
	hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, buf, 3, 5);
	hipEventRecord(e1, 0);
	hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	const double per_wave = ITERS * 8.0;
	printf("%-28s %d wave(s) per SIMD  %8.3f ms  %6.2f cycles per instruction of one wave, %5.2f per wave-instr per SIMD\n", name, waves, ms, ms * 1e6 / per_wave * g_ghz, ms * 1e6 / per_wave * g_ghz / waves);
}

int main()
{
	int *buf; hipMalloc(&buf, 256 * 8 * 256 * 8);
	{
		const int blocks = 256 * 8;
		unsigned long long *d_ticks, *h_ticks = (unsigned long long*)malloc(sizeof(unsigned long long) * 2 * blocks);
		hipMalloc(&d_ticks, sizeof(unsigned long long) * 2 * blocks);
		hipLaunchKernelGGL(k_clock, dim3(blocks), dim3(256), 0, 0, buf, 3, 5, d_ticks);
		hipLaunchKernelGGL(k_clock, dim3(blocks), dim3(256), 0, 0, buf, 3, 5, d_ticks);
		hipMemcpy(h_ticks, d_ticks, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
		double lo = 1e30, hi = 0, sum = 0;
		for(int i = 0; i < blocks; i++) {
			const double ghz = (double)h_ticks[2 * i] / ((double)h_ticks[2 * i + 1] / 0.1);   // shader ticks per ns: the constant counter runs at 100 MHz
			lo = ghz < lo ? ghz : lo; hi = ghz > hi ? ghz : hi; sum += ghz;
		}
		g_ghz = sum / blocks;
		printf("shader clock under an all-SIMD integer multiply-add load: mean %.3f GHz (min %.3f, max %.3f over %d workgroups; s_memtime ticks per s_memrealtime tick x 100 MHz)\n", g_ghz, lo, hi, blocks);
		hipFree(d_ticks); free(h_ticks);
	}
	run("mad_i32_i24", k_mad24, buf); run("mul_i32_i24", k_mul24, buf); run("mad_u32_u24", k_madu24, buf); run("add_u32", k_add, buf); run("add3_u32", k_add3, buf);
	run("mul_lo_u32", k_mullo, buf); run("dot2_i32_i16", k_dot2, buf); run("dot4_i32_i8", k_dot4, buf); run("sad_u32", k_sad, buf);
	run("max_i32", k_maxi, buf); run("fma_f32", k_fma, buf); run("mul_f32", k_mulf, buf); run("mad_i32_i16", k_madi16, buf); run("pk_mad_i16", k_pkmad, buf);
	run("ashrrev_i32", k_ashr, buf); run("lshl_add_u32", k_lshladd, buf);
	run("mad_i64_i32", k_mad64, (long long*)buf);
	run("alignbit_b32", k_alignbit, buf); run("perm_b32", k_perm, buf); run("bfe_i32", k_bfe, buf); run("mov_b32", k_mov, buf); run("xor_b32", k_xor, buf);
	run("ashrrev_i64", k_ashr64, (long long*)buf); run("lshl_add_u64", k_lshladd64, (long long*)buf);
	run("pk_mul_f32", k_pkmulf, (long long*)buf); run("pk_add_f32", k_pkaddf, (long long*)buf); run("pk_fma_f32", k_pkfmaf, (long long*)buf); run("fma_f64", k_fma64, (long long*)buf);
	for(int waves = 1; waves <= 4; waves *= 2) {
		run_few("mad_i32_i24, 8 chains", k_mad24, buf, waves); run_few("mad_i32_i24, 1 chain", k_mad24_dep, buf, waves);
		run_few("add_u32, 8 chains", k_add, buf, waves); run_few("add_u32, 1 chain", k_add_dep, buf, waves);
	}
	return 0;
}
