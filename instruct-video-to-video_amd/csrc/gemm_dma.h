// LDS-DMA helpers shared by the GEMM kernels (gemm.hip, gemm_p8.hip).
#pragma once
#include "common.h"

#define BK 64
#define OOB_OFFSET 0x80000000u  // byte offset beyond every descriptor's num_records (2^31-1): loads return 0

typedef __amdgpu_buffer_rsrc_t srd_t;
__device__ __forceinline__ srd_t make_srd(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7FFFFFFF, 0x00020000);
}
// 16 bytes per lane: LDS[lds_wave_base + lane*16] = mem[srd.base + voff + soff] (zeros when out of range)
__device__ __forceinline__ void dma16(srd_t srd, unsigned voff, int soff, void* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// internal entry of the 8-phase 256x256 kernel (gemm_p8.hip); INSV2V_EUNSUPPORTED when the problem is not eligible
// internal entry of the round-4 8-phase kernel with interleaved half-tile ownership (gemm_q8.hip); same eligibility as gemm_p8
int insv2v_gemm_q8(const insv2v_gemm_desc& d, int variant, hipStream_t s);
// internal entry of the 256 x 320 tile form of the round-4 engine (gemm_r8.hip): LINEAR / CONV3X3, no activation
int insv2v_gemm_r8(const insv2v_gemm_desc& d, int variant, hipStream_t s);
// internal entry of the 4-wave, two-workgroups-per-CU persistent kernel (gemm_w4.hip)
int insv2v_gemm_w4(const insv2v_gemm_desc& d, int variant, hipStream_t s);
