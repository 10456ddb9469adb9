// Round 5, GroupNorm-fold failures (profiles/r05_gnfold_select.txt), second micro-probe.  What the failing builds of split_linear_h3l_kernel<GNF> have in common and the
// passing ones do not: `v_pk_mul_f32 vD[0:1], vA[0:1], vB[0:1] op_sel:[0,1]` -- the LOW product takes the HIGH register of a pair that a global_load_dwordx2 delivered
// (mean, rstd), issued right behind the s_waitcnt that covers that load, with matrix instructions of the previous loop iteration still in the pipe and further loads
// outstanding.  The wrong values were always the LOW products of lanes 48-63.  Here: MF x 4 v_mfma_f32_32x32x16_f16, one global_load_dwordx2 into v[182:183] followed by
// EXTRA more loads (left outstanding), s_waitcnt vmcnt(EXTRA), then the packed multiply in the form under test; the products are compared with plain v_mul_f32 of values
// fetched again later.  Counted per lane quarter.
// RESULT (profiles/r05_pk_opsel_erratum.txt): with SPLIT = 2 (the SIMDs' second waves issue MFMAs with plain VALU between them) the LOW result of the packed instruction is exactly 0 in
// lanes 48-63 whenever its op_sel takes the HIGH register of SOURCE 1 (v_pk_mul_f32 and v_pk_add_f32; 1e5 ... 1e7 of 5e8 per launch); every other select form is exact.
// The registers are the failing kernel's own (destination v[166:167], source 0 v[130:131], source 1 v[182:183]): the kernel then allocates > 184 registers, two waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 tools/micro/pk_opsel_after_load.hip -o tools/micro/bin/pk_opsel_after_load && tools/micro/bin/pk_opsel_after_load
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// FORM 0: v_pk_mul_f32 d, x, y op_sel:[0,1]      (low = x.lo * y.hi, high = x.hi * y.hi)     <- the failing builds
// FORM 1: v_pk_mul_f32 d, x, y                    (low = x.lo * y.lo, high = x.hi * y.hi)
// FORM 2: v_mov t, y.hi ; v_pk_mul_f32 d, t, x op_sel_hi:[0,1]   (the passing builds)
template <int FORM, int MF, int EXTRA, int SPLIT = 0>
__global__ __launch_bounds__(512) void k(const float* __restrict__ buf, int nbuf, unsigned long long* __restrict__ bad, int iters) {
  const int tid = threadIdx.x, quarter = (tid & 63) >> 4;
  unsigned long long bad_lo = 0, bad_hi = 0;
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  f16x8 fa, fb;
  for (int e = 0; e < 8; ++e) {
    fa[e] = (_Float16)(0.001f * (tid + e));
    fb[e] = (_Float16)(0.002f * (tid - e));
  }
  f32x4 sinkv = {0.f, 0.f, 0.f, 0.f};
  if (SPLIT && tid >= 256) {                                                         // SPLIT: waves 4-7 (the SIMDs' second waves) only run MFMAs, for the whole launch
    float va = 1.0f + tid, vb = 0.5f;
    for (int it = 0; it < iters * 4; ++it)
      if (SPLIT == 2)                                                                // MFMAs with plain VALU between them: what the aggressor experiments of tools/gnf_asm_probe.py say it takes
        asm volatile(".rept 8\n\tv_mfma_f32_32x32x16_f16 %0, %4, %5, %0\n\tv_add_f32 %6, %6, %7\n\tv_mul_f32 %7, %7, %6\n\tv_mfma_f32_32x32x16_f16 %1, %4, %5, %1\n\tv_add_f32 %6, %6, %7\n\t"
                     "v_mfma_f32_32x32x16_f16 %2, %4, %5, %2\n\tv_fma_f32 %7, %6, %7, %6\n\tv_max_f32 %6, %6, %7\n\tv_mfma_f32_32x32x16_f16 %3, %4, %5, %3\n\tv_add_f32 %7, %6, %7\n\t.endr"
                     : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) : "v"(fa), "v"(fb), "v"(va), "v"(vb));
      else
      asm volatile(".rept 8\n\tv_mfma_f32_32x32x16_f16 %0, %4, %5, %0\n\tv_mfma_f32_32x32x16_f16 %1, %4, %5, %1\n\tv_mfma_f32_32x32x16_f16 %2, %4, %5, %2\n\t"
                   "v_mfma_f32_32x32x16_f16 %3, %4, %5, %3\n\t.endr"
                   : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) : "v"(fa), "v"(fb));
    float sk = 0.f;
    for (int j = 0; j < 4; ++j) sk += acc[j][0] + acc[j][7];
    if (sk == 1234.5f) atomicAdd(bad, 1ull);
    return;
  }
  for (int it = 0; it < iters; ++it) {
    const uint32_t s = (uint32_t)(it * 2654435761u) ^ (uint32_t)(tid * 40503u + blockIdx.x * 977u);
    const float* src = buf + 2 * (s % (uint32_t)(nbuf / 2 - 8));                       // 8-byte aligned pair (y.lo, y.hi), different lines per lane: the returns straggle
    const float* oth = buf + 4 * ((s >> 3) % (uint32_t)(nbuf / 4 - 64));
    const float x0 = 1.0f + (float)(s & 1023) * 0.001f, x1 = 2.0f + (float)((s >> 10) & 1023) * 0.001f;
    float lo, hi;
    f32x4 e0, e1, e2, e3;
    if (MF)
      asm volatile(".rept %c6\n\tv_mfma_f32_32x32x16_f16 %0, %4, %5, %0\n\tv_mfma_f32_32x32x16_f16 %1, %4, %5, %1\n\tv_mfma_f32_32x32x16_f16 %2, %4, %5, %2\n\t"
                   "v_mfma_f32_32x32x16_f16 %3, %4, %5, %3\n\t.endr"
                   : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) : "v"(fa), "v"(fb), "n"(MF));
#define LOADS                                                                                                                                   \
  "v_mov_b32 v130, %6\n\tv_mov_b32 v131, %7\n\t"                                                                                               \
  "global_load_dwordx2 v[182:183], %8, off\n\t"                                                                                                \
  ".if %c10 >= 1\n\tglobal_load_dwordx4 %2, %9, off\n\t.endif\n\t"                                                                             \
  ".if %c10 >= 2\n\tglobal_load_dwordx4 %3, %9, off offset:64\n\t.endif\n\t"                                                                   \
  ".if %c10 >= 3\n\tglobal_load_dwordx4 %4, %9, off offset:128\n\t.endif\n\t"                                                                  \
  ".if %c10 >= 4\n\tglobal_load_dwordx4 %5, %9, off offset:192\n\t.endif\n\t"                                                                  \
  "s_waitcnt vmcnt(%c10)\n\t"
#define TAIL "s_nop 7\n\ts_nop 7\n\tv_mov_b32 %0, v166\n\tv_mov_b32 %1, v167\n\ts_waitcnt vmcnt(0)"
#define OPS : "=v"(lo), "=v"(hi), "=&v"(e0), "=&v"(e1), "=&v"(e2), "=&v"(e3) : "v"(x0), "v"(x1), "v"(src), "v"(oth), "n"(EXTRA) : "memory", "v166", "v167", "v130", "v131", "v182", "v183", "v220"
    if (FORM == 0) asm volatile(LOADS "v_pk_mul_f32 v[166:167], v[130:131], v[182:183] op_sel:[0,1]\n\t" TAIL OPS);
    if (FORM == 1) asm volatile(LOADS "v_pk_mul_f32 v[166:167], v[130:131], v[182:183]\n\t" TAIL OPS);
    if (FORM == 2) asm volatile(LOADS "v_mov_b32 v220, v183\n\tv_pk_mul_f32 v[166:167], v[220:221], v[130:131] op_sel_hi:[0,1]\n\t" TAIL OPS);
    // the other select forms the library's kernels contain, in the same harness (x = v[130:131] from v_mov, y = v[182:183] from the load)
    if (FORM == 3) asm volatile(LOADS "v_pk_mul_f32 v[166:167], v[182:183], v[130:131] op_sel:[1,0]\n\t" TAIL OPS);                       // cross select on source 0
    if (FORM == 4) asm volatile(LOADS "v_pk_mul_f32 v[166:167], v[130:131], v[182:183] op_sel_hi:[1,0]\n\t" TAIL OPS);                    // HIGH product takes the LOW register of source 1 (broadcast)
    if (FORM == 5) asm volatile(LOADS "v_pk_mul_f32 v[166:167], v[182:183], v[130:131] op_sel_hi:[0,1]\n\t" TAIL OPS);                    // ... of source 0
    if (FORM == 6) asm volatile(LOADS "v_pk_fma_f32 v[166:167], v[130:131], v[130:131], v[182:183] op_sel:[0,0,1] op_sel_hi:[1,1,0]\n\t" TAIL OPS);   // source 2 swapped
    if (FORM == 7) asm volatile(LOADS "v_pk_add_f32 v[166:167], v[130:131], v[182:183] op_sel:[0,1] op_sel_hi:[1,0]\n\t" TAIL OPS);        // source 1 swapped (the horizontal-add form)
    if (FORM == 8) asm volatile(LOADS "v_pk_mul_f32 v[166:167], v[182:183], v[130:131] op_sel:[0,1]\n\t" TAIL OPS);                       // source 1 cross, the LOADED pair as source 0
    if (FORM == 10) asm volatile(LOADS "v_pk_fma_f32 v[166:167], v[130:131], v[182:183], v[130:131] op_sel:[0,1,0]\n\t" TAIL OPS);                // source 1 cross in a fused multiply-add
    if (FORM == 9) asm volatile(LOADS "v_mul_f32 v166, v130, v183\n\tv_mul_f32 v167, v131, v183\n\t" TAIL OPS);
    const float y0 = src[0], y1 = src[1];
    float want_lo = x0 * y1, want_hi = x1 * y1;                                        // FORM 0, 2, 3, 9
    if (FORM == 1) want_lo = x0 * y0;
    if (FORM == 4) { want_lo = x0 * y0; want_hi = x1 * y0; }
    if (FORM == 5) { want_lo = y0 * x0; want_hi = y0 * x1; }
    if (FORM == 6) { want_lo = fmaf(x0, x0, y1); want_hi = fmaf(x1, x1, y0); }
    if (FORM == 7) { want_lo = x0 + y1; want_hi = x1 + y0; }
    if (FORM == 8) { want_lo = y0 * x1; want_hi = y1 * x1; }
    if (FORM == 10) { want_lo = fmaf(x0, y1, x0); want_hi = fmaf(x1, y1, x1); }
    if (lo != want_lo && bad_lo == 0) {                                                // first wrong product of this thread: what did it get?
      float* smp = reinterpret_cast<float*>(bad + 8);
      smp[0] = lo; smp[1] = x0; smp[2] = y0; smp[3] = y1; smp[4] = hi; smp[5] = x1;
    }
    bad_lo += lo != want_lo;
    bad_hi += hi != want_hi;
    if (EXTRA >= 1) sinkv += e0;
    if (EXTRA >= 2) sinkv += e1;
    if (EXTRA >= 3) sinkv += e2;
    if (EXTRA >= 4) sinkv += e3;
  }
  float sink = sinkv.x + sinkv.y + sinkv.z + sinkv.w;
  for (int j = 0; j < 4; ++j) sink += acc[j][0] + acc[j][7];
  if (sink == 1234.5f) bad_lo += 1;
  if (bad_lo) atomicAdd(bad + quarter, bad_lo);
  if (bad_hi) atomicAdd(bad + 4 + quarter, bad_hi);
}

static float* g_buf;
static const int NBUF = 1 << 24;                                                      // 64 MB of floats: most loads miss the caches

template <int FORM, int MF, int EXTRA, int THREADS, int SPLIT = 0>
void run(unsigned long long* bad, int blocks, int iters) {
  (void)hipMemset(bad, 0, 128);
  hipLaunchKernelGGL((k<FORM, MF, EXTRA, SPLIT>), dim3(blocks), dim3(THREADS), 0, 0, g_buf, NBUF, bad, iters);
  (void)hipDeviceSynchronize();
  unsigned long long h[16];
  (void)hipMemcpy(h, bad, 128, hipMemcpyDeviceToHost);
  static const char* names[11] = {"pk_mul x, y op_sel:[0,1]    ", "pk_mul (no op_sel)         ", "v_mov hi; pk_mul op_sel_hi ", "pk_mul y, x op_sel:[1,0]    ", "pk_mul x, y op_sel_hi:[1,0] ", "pk_mul y, x op_sel_hi:[0,1] ",
                                  "pk_fma x, x, y src2 swapped ", "pk_add x, y src1 swapped    ", "pk_mul y, x op_sel:[0,1]    ", "2 x v_mul_f32              ", "pk_fma x, y, x op_sel:[0,1,0]"};
  printf("  %s%s  %d x 4 MFMA in flight, %d loads left outstanding, %d waves / workgroup, %d workgroups: %.3g products | wrong LOW by lane quarter: %llu %llu %llu %llu | wrong HIGH: %llu %llu %llu %llu\n",
         SPLIT == 2 ? "[other wave of every SIMD: MFMAs + VALU] " : SPLIT ? "[other wave of every SIMD: MFMAs only] " : "", names[FORM], MF, EXTRA, THREADS / 64, blocks, (double)blocks * THREADS * iters, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
  if (h[0] + h[1] + h[2] + h[3]) {
    const float* smp = reinterpret_cast<const float*>(h + 8);
    printf("      a wrong LOW product: got %.9g for x.lo %.9g, y = (%.9g, %.9g): x.lo * y.hi = %.9g, x.lo * y.lo = %.9g; its HIGH product %.9g (x.hi %.9g)\n", smp[0], smp[1], smp[2], smp[3],
           smp[1] * smp[3], smp[1] * smp[2], smp[4], smp[5]);
  }
}

__global__ void fill(float* b, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) b[i] = 0.5f + (float)((i * 2654435761u) >> 20) * 0.001f;
}

int main() {
  unsigned long long* bad;
  (void)hipMalloc(&bad, 128);
  (void)hipMalloc(&g_buf, (size_t)NBUF * 4);
  hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, g_buf, NBUF);
  const int iters = 4000;
  for (int rep = 0; rep < 2; ++rep) {
    const int blocks = 256;
    // the configuration that reproduces it: the SIMDs' second waves issue MFMAs with plain VALU between them, the first waves load the pair and multiply with 2 or 4 more loads outstanding
    run<0, 0, 2, 512, 2>(bad, blocks, iters); run<0, 0, 4, 512, 2>(bad, blocks, iters); run<0, 0, 0, 512, 2>(bad, blocks, iters);
    run<1, 0, 4, 512, 2>(bad, blocks, iters); run<2, 0, 4, 512, 2>(bad, blocks, iters); run<9, 0, 4, 512, 2>(bad, blocks, iters);          // controls
    run<3, 0, 4, 512, 2>(bad, blocks, iters); run<4, 0, 4, 512, 2>(bad, blocks, iters); run<5, 0, 4, 512, 2>(bad, blocks, iters); run<6, 0, 4, 512, 2>(bad, blocks, iters);
    run<7, 0, 4, 512, 2>(bad, blocks, iters); run<8, 0, 4, 512, 2>(bad, blocks, iters); run<10, 0, 4, 512, 2>(bad, blocks, iters); run<10, 0, 2, 512, 2>(bad, blocks, iters);
    run<3, 0, 2, 512, 2>(bad, blocks, iters); run<4, 0, 2, 512, 2>(bad, blocks, iters); run<6, 0, 2, 512, 2>(bad, blocks, iters); run<7, 0, 2, 512, 2>(bad, blocks, iters); run<8, 0, 2, 512, 2>(bad, blocks, iters);
    run<0, 0, 4, 512, 1>(bad, blocks, iters); run<0, 0, 4, 512, 0>(bad, blocks, iters);                                                     // aggressor: MFMAs only / the same code in every wave
  }
  return 0;
}
