// tools/valu_rate.hip — cost of wave64 instructions on gfx950, measured: nanoseconds of SIMD time per instruction
// (kernel wall time / instructions per SIMD) with 1, 2, 4 and 8 waves per SIMD streaming 8 independent chains of the
// same operation.  Prices the VALU-bound kernels and guides instruction selection: on this chip adds,
// ands and fp32 multiplies issue about twice as fast as shifts, min/max, 24-bit multiplies or conversions.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 -o tools/valu_rate tools/valu_rate.hip && tools/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ONE(OP, A) asm volatile(OP : "+v"(A) : "v"(k), "s"(m64));
#define ROUND(OP) ONE(OP, a0) ONE(OP, a1) ONE(OP, a2) ONE(OP, a3) ONE(OP, a4) ONE(OP, a5) ONE(OP, a6) ONE(OP, a7)
#define BODY(OP) ROUND(OP) ROUND(OP) ROUND(OP) ROUND(OP) ROUND(OP) ROUND(OP) ROUND(OP) ROUND(OP)

#define KERNEL(NAME, OP) \
	__global__ void NAME(unsigned long long* out, unsigned* sink, int iters) \
	{ \
		unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, k = 3 + (threadIdx.x & 1); \
		const unsigned long long m64 = 0x5555555555555555ull; \
		const unsigned long long t0 = __builtin_readcyclecounter(); \
		for (int i = 0; i < iters; ++i) { BODY(OP) } \
		const unsigned long long t1 = __builtin_readcyclecounter(); \
		if ((threadIdx.x & 63) == 0) atomicMax(&out[blockIdx.x], t1 - t0); \
		sink[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7; \
	}

KERNEL(k_add, "v_add_u32 %0, %0, %1")
KERNEL(k_sub, "v_sub_u32 %0, %0, %1")
KERNEL(k_and, "v_and_b32 %0, %0, %1")
KERNEL(k_or, "v_or_b32 %0, %0, %1")
KERNEL(k_xor, "v_xor_b32 %0, %0, %1")
KERNEL(k_mov, "v_mov_b32 %0, %1")
KERNEL(k_lshl, "v_lshlrev_b32 %0, 3, %0")
KERNEL(k_lshr, "v_lshrrev_b32 %0, 3, %0")
KERNEL(k_ashr, "v_ashrrev_i32 %0, 3, %0")
KERNEL(k_bfe, "v_bfe_u32 %0, %0, 3, 5")
KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %0, 2, %1")
KERNEL(k_add_lshl, "v_add_lshl_u32 %0, %0, %1, 2")
KERNEL(k_lshl_or, "v_lshl_or_b32 %0, %0, 2, %1")
KERNEL(k_and_or, "v_and_or_b32 %0, %0, %1, %0")
KERNEL(k_or3, "v_or3_b32 %0, %0, %1, %0")
KERNEL(k_add3, "v_add3_u32 %0, %0, %1, %0")
KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %0")
KERNEL(k_bfi, "v_bfi_b32 %0, %1, %0, %0")
KERNEL(k_max, "v_max_i32 %0, %0, %1")
KERNEL(k_min, "v_min_u32 %0, %0, %1")
KERNEL(k_med3, "v_med3_i32 %0, %0, %1, 7")
KERNEL(k_bcnt, "v_bcnt_u32_b32 %0, %0, %1")
KERNEL(k_mul24, "v_mul_u32_u24 %0, %0, %1")
KERNEL(k_mad24, "v_mad_u32_u24 %0, %0, %1, %0")
KERNEL(k_mullo, "v_mul_lo_u32 %0, %0, %1")
KERNEL(k_mulhi, "v_mul_hi_u32 %0, %0, %1")
KERNEL(k_mulf, "v_mul_f32 %0, %0, %1")
KERNEL(k_addf, "v_add_f32 %0, %0, %1")
KERNEL(k_fma, "v_fma_f32 %0, %0, %1, %0")
KERNEL(k_cvt_fi, "v_cvt_f32_i32 %0, %0")
KERNEL(k_cvt_if, "v_cvt_i32_f32 %0, %0")
KERNEL(k_rcp, "v_rcp_f32 %0, %0")
KERNEL(k_rsq, "v_rsq_f32 %0, %0")
KERNEL(k_sqrt, "v_sqrt_f32 %0, %0")
KERNEL(k_cnd_vcc, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL(k_cnd_sgpr, "v_cndmask_b32 %0, %0, %1, %2")
KERNEL(k_cmp, "v_cmp_lt_u32 vcc, %0, %1\n\tv_add_u32 %0, %0, %1")
KERNEL(k_cmp_cnd, "v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc")
KERNEL(k_cmp_sgpr_cnd, "v_cmp_lt_u32 s[20:21], %0, %1\n\tv_cndmask_b32 %0, %0, %1, s[20:21]")
KERNEL(k_sdwa, "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD")
KERNEL(k_dpp, "v_add_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL(k_readfirst, "v_readfirstlane_b32 s22, %0\n\tv_add_u32 %0, s22, %1")
KERNEL(k_ds_read_u8, "ds_read_u8 %0, %1 offset:64\n\ts_waitcnt lgkmcnt(0)")
KERNEL(k_ds_read_b32, "ds_read_b32 %0, %1 offset:64\n\ts_waitcnt lgkmcnt(0)")
KERNEL(k_ds_bperm, "ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)")

typedef void (*Kern)(unsigned long long*, unsigned*, int);

void run(const char* name, Kern kern, int perBody)
{
	const int iters = 500;
	unsigned long long* dOut; unsigned* dSink;
	(void)hipMalloc(&dOut, 512 * 8); (void)hipMalloc(&dSink, 512 * 1024 * 4);
	printf("%-34s", name);
	for (int threads : { 256, 512, 1024, 2048 }) { // 1, 2, 4 waves per SIMD (one block per CU), 8 (two blocks of 1024)
		const int blocks = threads == 2048 ? 512 : 256, tpb = threads == 2048 ? 1024 : threads;
		(void)hipMemset(dOut, 0, 512 * 8);
		hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
		hipLaunchKernelGGL(kern, dim3(blocks), dim3(tpb), 0, 0, dOut, dSink, 10); // warm
		(void)hipEventRecord(e0, 0);
		hipLaunchKernelGGL(kern, dim3(blocks), dim3(tpb), 0, 0, dOut, dSink, iters);
		(void)hipEventRecord(e1, 0);
		(void)hipDeviceSynchronize();
		float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
		const double perSimd = (double)threads / 64.0 / 4.0 * iters * 64.0 * perBody; // instructions of the measured kind per SIMD
		printf(" | %dw: %5.2f ns", threads / 256, ms * 1e6 / perSimd);
		(void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
	}
	printf("\n");
	(void)hipFree(dOut); (void)hipFree(dSink);
}

int main()
{
	printf("ns of SIMD time per wave64 instruction (wall time / instructions per SIMD) at 1, 2, 4, 8 waves per SIMD; 2.4 GHz: 1 ns = 2.4 cycles\n");
#define R(K) run(#K, K, 1)
	R(k_add); R(k_sub); R(k_and); R(k_or); R(k_xor); R(k_mov); R(k_lshl); R(k_lshr); R(k_ashr); R(k_bfe); R(k_lshl_add); R(k_add_lshl); R(k_lshl_or);
	R(k_and_or); R(k_or3); R(k_add3); R(k_perm); R(k_bfi); R(k_max); R(k_min); R(k_med3); R(k_bcnt); R(k_mul24); R(k_mad24); R(k_mullo); R(k_mulhi);
	R(k_mulf); R(k_addf); R(k_fma); R(k_cvt_fi); R(k_cvt_if); R(k_rcp); R(k_rsq); R(k_sqrt); R(k_cnd_vcc); R(k_cnd_sgpr);
	printf("-- pairs (per PAIR of instructions) --\n");
	R(k_cmp); R(k_cmp_cnd); R(k_cmp_sgpr_cnd); R(k_readfirst);
	printf("-- single --\n");
	R(k_sdwa); R(k_dpp); R(k_ds_read_u8); R(k_ds_read_b32); R(k_ds_bperm);
	return 0;
}
