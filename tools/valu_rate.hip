// tools/valu_rate.hip — issue rate of wave64 VALU instructions on gfx950, measured: cycles per instruction of one wave's
// stream of independent operations (8 chains), with 1, 2 and 4 waves per SIMD resident.  Used to price the VALU-bound
// kernels: is a wave64 integer / fp32 instruction 2 or 4 cycles of a SIMD?
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_rate tools/valu_rate.hip && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define BODY(OP) \
	REP8(asm volatile(OP " %0, %0, %8\n\t" OP " %1, %1, %8\n\t" OP " %2, %2, %8\n\t" OP " %3, %3, %8\n\t" \
	                  OP " %4, %4, %8\n\t" OP " %5, %5, %8\n\t" OP " %6, %6, %8\n\t" OP " %7, %7, %8" \
	                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));)

template <int WHICH>
__global__ void k_rate(unsigned long long* out, unsigned* sink, int iters)
{
	unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, k = 3 + (threadIdx.x & 1);
	const unsigned long long t0 = __builtin_readcyclecounter();
	for (int i = 0; i < iters; ++i) {
		if (WHICH == 0) { BODY("v_add_u32") }
		if (WHICH == 1) { BODY("v_and_b32") }
		if (WHICH == 2) { BODY("v_lshlrev_b32") }
		if (WHICH == 3) { BODY("v_mul_f32") }
		if (WHICH == 4) { BODY("v_mul_lo_u32") }
		if (WHICH == 5) { BODY("v_mul_u32_u24") }
		if (WHICH == 6) { BODY("v_max_i32") }
		if (WHICH == 7) { BODY("v_cndmask_b32") }
	}
	const unsigned long long t1 = __builtin_readcyclecounter();
	if ((threadIdx.x & 63) == 0) atomicMax(&out[blockIdx.x], t1 - t0); // the slowest wave of the block (the arbiter favours the oldest)
	sink[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

template <int WHICH>
void run(const char* name)
{
	const int iters = 2000;
	unsigned long long* dOut; unsigned* dSink;
	hipMalloc(&dOut, 512 * 8); hipMalloc(&dSink, 512 * 1024 * 4);
	printf("%-16s", name);
	for (int threads : { 64, 256, 512, 1024, 2048 }) { // 1 wave per CU .. 8 waves per SIMD (2048: two blocks of 1024 per CU)
		const int blocks = threads == 2048 ? 512 : 256, tpb = threads == 2048 ? 1024 : threads;
		hipMemset(dOut, 0, 512 * 8);
		hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
		hipEventRecord(e0, 0);
		hipLaunchKernelGGL(k_rate<WHICH>, dim3(blocks), dim3(tpb), 0, 0, dOut, dSink, iters);
		hipEventRecord(e1, 0);
		hipDeviceSynchronize();
		float ms = 0; hipEventElapsedTime(&ms, e0, e1);
		std::vector<unsigned long long> h(blocks);
		hipMemcpy(h.data(), dOut, blocks * 8, hipMemcpyDeviceToHost);
		double sum = 0; for (auto v : h) sum += (double)v;
		// SIMD throughput from the wall clock: wave-instructions per SIMD / kernel time
		const double perSimd = (double)threads / 64.0 / 4.0 * iters * 64.0;
		printf(" | %4d thr/CU: %5.2f cyc/inst (slowest wave), %5.1f ns per inst per SIMD", threads, sum / blocks / (iters * 64.0), ms * 1e6 / (perSimd > 0 ? perSimd : 1));
	}
	printf("\n");
	hipFree(dOut); hipFree(dSink);
}

int main()
{
	printf("cycles (s_memtime) per wave64 instruction as seen by ONE wave; with w waves per SIMD the SIMD issues w / that many per cycle\n");
	run<0>("v_add_u32"); run<1>("v_and_b32"); run<2>("v_lshlrev_b32"); run<3>("v_mul_f32"); run<4>("v_mul_lo_u32"); run<5>("v_mul_u32_u24"); run<6>("v_max_i32"); run<7>("v_cndmask_b32");
	return 0;
}
