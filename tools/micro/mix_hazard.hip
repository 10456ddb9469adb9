// Does gfx950 need a wait state between v_fma_mixlo_f16 vX and v_fma_mixhi_f16 vX (the 16-bit-destination forwarding rule hipcc enforces for
// instructions it emits itself, tools/isa_hazards.py rule D)?  Every lane splits the same stream of fp32 pairs (fresh v_rcp_f32 results, as
// in rba_reduce_up4_mx_kernel) into h = f16(x), l = f16(x - h) three ways and counts results that differ from the reference form:
//   A  mixlo r ; mixhi r                       back to back (round 3's one-statement asm)
//   B  mixlo r ; s_nop 0 ; mixhi r             the wait state hipcc inserts
//   C  mixlo r0 ; mixlo r1 ; mixhi r0 ; mixhi r1   interleaved (round 4's form)
// reference: l built from two full-register conversions (v_cvt_f16_f32 of fma(h, -1, x), v_pack), no 16-bit destination writes at all.
// Run alone on its SIMD (1 wave per SIMD) and with 8 waves per SIMD (other waves' instructions fall between the pair).
// hipcc --offload-arch=gfx950 -O3 tools/micro/mix_hazard.hip -o /tmp/mix_hazard && /tmp/mix_hazard
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t ref_l(float a, float b, uint32_t hp) {
  const h2 hh = __builtin_bit_cast(h2, hp);
  const float ra = __builtin_fmaf((float)hh.x, -1.0f, a), rb = __builtin_fmaf((float)hh.y, -1.0f, b);
  const h2 ll = {(_Float16)ra, (_Float16)rb};
  return __builtin_bit_cast(uint32_t, ll);
}

template <int KIND>
__global__ __launch_bounds__(256) void k(const float* __restrict__ x, unsigned long long* __restrict__ bad, int iters) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  float s0 = x[t & 4095] + 1.5f, s1 = x[(t + 7) & 4095] + 2.5f, s2 = x[(t + 13) & 4095] + 3.5f, s3 = x[(t + 29) & 4095] + 4.5f;
  unsigned long long nb = 0;
  for (int it = 0; it < iters; ++it) {
    const float a0 = __builtin_amdgcn_rcpf(s0), a1 = __builtin_amdgcn_rcpf(s1), b0 = __builtin_amdgcn_rcpf(s2), b1 = __builtin_amdgcn_rcpf(s3);
    const h2 va = {(_Float16)a0, (_Float16)a1}, vb = {(_Float16)b0, (_Float16)b1};
    const uint32_t pa = __builtin_bit_cast(uint32_t, va), pb = __builtin_bit_cast(uint32_t, vb);
    uint32_t ra, rb;
    if (KIND == 0) {
      asm volatile("s_nop 0\n\tv_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, %1, %2, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\ts_nop 0"
                   : "=&v"(ra) : "v"(pa), "v"(-1.0f), "v"(a0), "v"(a1));
      asm volatile("s_nop 0\n\tv_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, %1, %2, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\ts_nop 0"
                   : "=&v"(rb) : "v"(pb), "v"(-1.0f), "v"(b0), "v"(b1));
    } else if (KIND == 1) {
      asm volatile("s_nop 0\n\tv_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]\n\ts_nop 0\n\tv_fma_mixhi_f16 %0, %1, %2, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\ts_nop 0"
                   : "=&v"(ra) : "v"(pa), "v"(-1.0f), "v"(a0), "v"(a1));
      asm volatile("s_nop 0\n\tv_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]\n\ts_nop 0\n\tv_fma_mixhi_f16 %0, %1, %2, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\ts_nop 0"
                   : "=&v"(rb) : "v"(pb), "v"(-1.0f), "v"(b0), "v"(b1));
    } else {
      asm volatile("s_nop 0\n\t"
                   "v_fma_mixlo_f16 %0, %2, %4, %5 op_sel_hi:[1,0,0]\n\t"
                   "v_fma_mixlo_f16 %1, %3, %4, %7 op_sel_hi:[1,0,0]\n\t"
                   "v_fma_mixhi_f16 %0, %2, %4, %6 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                   "v_fma_mixhi_f16 %1, %3, %4, %8 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                   "s_nop 0"
                   : "=&v"(ra), "=&v"(rb) : "v"(pa), "v"(pb), "v"(-1.0f), "v"(a0), "v"(a1), "v"(b0), "v"(b1));
    }
    nb += (ra != ref_l(a0, a1, pa)) + (rb != ref_l(b0, b1, pb));
    s0 = s0 * 1.0001f + 0.37f; s1 = s1 * 0.9999f + 0.11f; s2 = s2 * 1.0002f + 0.53f; s3 = s3 * 0.9998f + 0.71f;   // stays in (1, ~1e4): rcp in f16's normal range
    if (s0 > 9000.f) s0 = 1.25f;
    if (s2 > 9000.f) s2 = 1.75f;
  }
  if (nb) atomicAdd(bad, nb);
}

template <int KIND>
void run(const char* name, const float* x, unsigned long long* bad, int blocks, int iters) {
  hipMemset(bad, 0, 8);
  hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(256), 0, 0, x, bad, iters);
  hipDeviceSynchronize();
  unsigned long long h = 0;
  hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
  printf("  %-44s blocks %5d  pairs %12.0f  results that differ from the reference: %llu\n", name, blocks, 2.0 * blocks * 256.0 * iters, h);
}

int main() {
  float* x;
  unsigned long long* bad;
  hipMalloc(&x, 4096 * 4);
  hipMalloc(&bad, 8);
  float hx[4096];
  for (int i = 0; i < 4096; ++i) hx[i] = (float)((i * 2654435761u) >> 8) / 16777216.0f;
  hipMemcpy(x, hx, sizeof(hx), hipMemcpyHostToDevice);
  const int iters = 20000;
  for (int blocks : {256, 2048}) {                                   // one wave per SIMD; eight waves per SIMD
    printf("%s\n", blocks == 256 ? "one workgroup per CU (a wave alone on its SIMD):" : "eight workgroups per CU (8 waves per SIMD):");
    run<0>("A mixlo r; mixhi r (back to back)", x, bad, blocks, iters);
    run<1>("B mixlo r; s_nop 0; mixhi r", x, bad, blocks, iters);
    run<2>("C mixlo r0; mixlo r1; mixhi r0; mixhi r1", x, bad, blocks, iters);
  }
  return 0;
}
