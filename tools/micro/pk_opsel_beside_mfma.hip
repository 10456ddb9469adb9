// Round 5, GroupNorm-fold failures, third micro-probe (profiles/r05_gnfold_select.txt).  tools/gnf_asm_probe.py showed: the zero LOW products of
// `v_pk_mul_f32 vD, vA, vB op_sel:[0,1]` disappear when the kernel's MFMAs are removed -- also in the PROLOGUE of a workgroup, which has issued no MFMA yet: the matrix
// instructions of the OTHER wave of the SIMD are what it takes.  Here one half of every workgroup's waves (one per SIMD) runs v_mfma_f32_32x32x16_f16 back to back for the
// whole launch while the other half (the SIMDs' other waves) does nothing but packed multiplies in the form under test on known operands, and counts wrong products per lane quarter.
// hipcc --offload-arch=gfx950 -O3 tools/micro/pk_opsel_beside_mfma.hip -o tools/micro/bin/pk_opsel_beside_mfma && tools/micro/bin/pk_opsel_beside_mfma
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// FORM 0: op_sel:[0,1] (LOW product = A.lo * B.hi: cross select on source 1)   1: no op_sel   2: op_sel:[1,0] (cross select on source 0)   3: two v_mul_f32
// MFMA 0: the other waves idle (s_sleep)   1: they run MFMAs
template <int FORM, int MFMA>
__global__ __launch_bounds__(512) void k(unsigned long long* __restrict__ bad, int iters, float* sinkp) {
  const int tid = threadIdx.x, wave = tid >> 6, quarter = (tid & 63) >> 4;
  if (wave >= 4) {                                                                  // waves 4-7: the second wave of SIMD 0-3
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j)
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    f16x8 fa, fb;
    for (int e = 0; e < 8; ++e) {
      fa[e] = (_Float16)(0.001f * (tid + e));
      fb[e] = (_Float16)(0.002f * (tid - e));
    }
    float va = 1.0f + tid, vb = 0.5f;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 pa = {1.0f + tid, 2.0f}, pb = {0.5f, 0.25f};
    for (int it = 0; it < iters; ++it) {
      if (MFMA == 1)
        asm volatile(".rept 8\n\tv_mfma_f32_32x32x16_f16 %0, %4, %5, %0\n\tv_mfma_f32_32x32x16_f16 %1, %4, %5, %1\n\tv_mfma_f32_32x32x16_f16 %2, %4, %5, %2\n\t"
                     "v_mfma_f32_32x32x16_f16 %3, %4, %5, %3\n\t.endr"
                     : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) : "v"(fa), "v"(fb));
      else if (MFMA == 2)                                                             // MFMAs with plain VALU between them (the GEMM loop's shape: gnf_asm_probe's aggressor experiments)
        asm volatile(".rept 8\n\tv_mfma_f32_32x32x16_f16 %0, %4, %5, %0\n\tv_add_f32 %6, %6, %7\n\tv_mul_f32 %7, %7, %6\n\tv_mfma_f32_32x32x16_f16 %1, %4, %5, %1\n\tv_add_f32 %6, %6, %7\n\t"
                     "v_mfma_f32_32x32x16_f16 %2, %4, %5, %2\n\tv_fma_f32 %7, %6, %7, %6\n\tv_max_f32 %6, %6, %7\n\tv_mfma_f32_32x32x16_f16 %3, %4, %5, %3\n\tv_add_f32 %7, %6, %7\n\t.endr"
                     : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) : "v"(fa), "v"(fb), "v"(va), "v"(vb));
      else if (MFMA == 3)                                                             // ... with PACKED VALU between them
        asm volatile(".rept 8\n\tv_mfma_f32_32x32x16_f16 %0, %4, %5, %0\n\tv_pk_add_f32 %6, %6, %7\n\tv_pk_mul_f32 %7, %7, %6\n\tv_mfma_f32_32x32x16_f16 %1, %4, %5, %1\n\tv_pk_add_f32 %6, %6, %7\n\t"
                     "v_mfma_f32_32x32x16_f16 %2, %4, %5, %2\n\tv_pk_fma_f32 %7, %6, %7, %6\n\tv_mfma_f32_32x32x16_f16 %3, %4, %5, %3\n\tv_pk_add_f32 %7, %6, %7\n\t.endr"
                     : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) : "v"(fa), "v"(fb), "v"(pa), "v"(pb));
      else
        asm volatile("s_sleep 16");
    }
    float sink = 0.f;
    for (int j = 0; j < 4; ++j) sink += acc[j][0] + acc[j][9];
    if (sink == 1234.5f) sinkp[0] = sink;
    return;
  }
  unsigned long long bad_lo = 0, bad_hi = 0;
  for (int it = 0; it < iters; ++it) {
    const uint32_t s = (uint32_t)(it * 2654435761u) ^ (uint32_t)(tid * 40503u + blockIdx.x * 977u);
    const float x0 = 1.0f + (float)(s & 1023) * 0.001f, x1 = 2.0f + (float)((s >> 10) & 1023) * 0.001f;
    const float y0 = 0.5f + (float)((s >> 20) & 255) * 0.01f, y1 = 3.0f + (float)((s >> 5) & 511) * 0.002f;
    float lo, hi;
#define SETUP "v_mov_b32 v130, %2\n\tv_mov_b32 v131, %3\n\tv_mov_b32 v182, %4\n\tv_mov_b32 v183, %5\n\ts_nop 4\n\t"
#define TAIL "s_nop 7\n\ts_nop 7\n\tv_mov_b32 %0, v166\n\tv_mov_b32 %1, v167"
#define OPS : "=v"(lo), "=v"(hi) : "v"(x0), "v"(x1), "v"(y0), "v"(y1) : "v130", "v131", "v166", "v167", "v182", "v183"
    if (FORM == 0) asm volatile(SETUP "v_pk_mul_f32 v[166:167], v[130:131], v[182:183] op_sel:[0,1]\n\t" TAIL OPS);
    if (FORM == 1) asm volatile(SETUP "v_pk_mul_f32 v[166:167], v[130:131], v[182:183]\n\t" TAIL OPS);
    if (FORM == 2) asm volatile(SETUP "v_pk_mul_f32 v[166:167], v[182:183], v[130:131] op_sel:[1,0]\n\t" TAIL OPS);
    if (FORM == 3) asm volatile(SETUP "v_mul_f32 v166, v130, v183\n\tv_mul_f32 v167, v131, v183\n\t" TAIL OPS);
    const float want_lo = FORM == 1 ? x0 * y0 : x0 * y1, want_hi = x1 * y1;
    bad_lo += lo != want_lo;
    bad_hi += hi != want_hi;
  }
  if (bad_lo) atomicAdd(bad + quarter, bad_lo);
  if (bad_hi) atomicAdd(bad + 4 + quarter, bad_hi);
}

template <int FORM, int MFMA>
void run(unsigned long long* bad, float* sink, int blocks, int iters) {
  (void)hipMemset(bad, 0, 64);
  hipLaunchKernelGGL((k<FORM, MFMA>), dim3(blocks), dim3(512), 0, 0, bad, iters, sink);
  (void)hipDeviceSynchronize();
  unsigned long long h[8];
  (void)hipMemcpy(h, bad, 64, hipMemcpyDeviceToHost);
  static const char* names[4] = {"v_pk_mul_f32 op_sel:[0,1] (source 1 cross)", "v_pk_mul_f32 (no op_sel)                 ", "v_pk_mul_f32 op_sel:[1,0] (source 0 cross)", "2 x v_mul_f32                            "};
  printf("  %s  other wave of the SIMD: %s  %d workgroups: %.3g products | wrong LOW by lane quarter: %llu %llu %llu %llu | wrong HIGH: %llu %llu %llu %llu\n", names[FORM],
         MFMA == 1 ? "MFMAs back to back      " : MFMA == 2 ? "MFMAs + plain VALU      " : MFMA == 3 ? "MFMAs + packed VALU     " : "idle                    ", blocks, (double)blocks * 256 * iters, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
}

int main() {
  unsigned long long* bad;
  float* sink;
  (void)hipMalloc(&bad, 64);
  (void)hipMalloc(&sink, 64);
  const int iters = 20000;
  for (int blocks : {256, 512}) {
    run<0, 0>(bad, sink, blocks, iters); run<0, 1>(bad, sink, blocks, iters); run<0, 2>(bad, sink, blocks, iters); run<0, 3>(bad, sink, blocks, iters); run<2, 2>(bad, sink, blocks, iters); run<3, 3>(bad, sink, blocks, iters); run<1, 1>(bad, sink, blocks, iters); run<2, 1>(bad, sink, blocks, iters); run<3, 1>(bad, sink, blocks, iters);
  }
  return 0;
}
