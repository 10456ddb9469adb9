// Round 5: the cause of the GroupNorm-fold failures (profiles/r05_gnfold_select.txt).  In every failing build of split_linear_h3l_kernel<GNF> the compiler had placed a PACKED fp32
// instruction (v_pk_mul_f32 / v_pk_add_f32) whose DESTINATION is the data register pair of a ds_write_b128 issued a few instructions earlier (the fourth of four back-to-back
// stores); in every passing build those registers were first rewritten by plain VALU.  tools/micro/ds_write_war.hip showed that a plain v_mov rewriting a store's data registers is
// interlocked.  Here: K back-to-back ds_write_b128, then -- after N unrelated plain VALU instructions -- `v_pk_mul_f32 v[100:101], x, y` (PK = 1) or two `v_mul_f32` (PK = 0) into
// the data registers of the last store.  Counted per quarter of the wave (lanes 0-15 ... 48-63): stores that reached LDS with the wrong data, and products that are not x * y afterwards.
// hipcc --offload-arch=gfx950 -O3 tools/micro/pk_after_ds_write.hip -o tools/micro/bin/pk_after_ds_write && tools/micro/bin/pk_after_ds_write
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define PROBE_SETUP                                                                                                                                            \
  "v_mov_b32 v100, %4\n\tv_mov_b32 v101, %5\n\tv_mov_b32 v102, %6\n\tv_mov_b32 v103, %7\n\tv_mov_b32 v104, %8\n\tv_mov_b32 v105, %9\n\t"                        \
  "v_mov_b32 v106, %10\n\tv_mov_b32 v107, %11\n\tv_mov_b32 v108, %4\n\tv_mov_b32 v109, %5\n\tv_mov_b32 v110, %6\n\tv_mov_b32 v111, %7\n\ts_nop 7\n\t"                \
  ".if %c13 >= 4\n\tds_write_b128 %3, v[108:111] offset:8192\n\tds_write_b128 %3, v[108:111] offset:16384\n\tds_write_b128 %3, v[108:111] offset:24576\n\t.endif\n\t"   \
  ".if %c13 >= 8\n\tds_write_b128 %3, v[108:111] offset:32768\n\tds_write_b128 %3, v[108:111] offset:40960\n\tds_write_b128 %3, v[108:111] offset:49152\n\t"            \
  "ds_write_b128 %3, v[108:111] offset:57344\n\t.endif\n\t"                                                                                                  \
  "ds_write_b128 %3, v[100:103]\n\t"                                                                                                                           \
  ".rept %c12\n\tv_xor_b32 %0, %0, %1\n\t.endr\n\t"
#define PROBE_TAIL "s_nop 7\n\ts_nop 7\n\tv_mov_b32 %2, v100"
#define PROBE_OPERANDS                                                                                                                                         \
  : "+v"(j0), "+v"(j1), "=v"(p0)                                                                                                                               \
  : "v"(a0), "v"(pat[0]), "v"(pat[1]), "v"(pat[2]), "v"(pat[3]), "v"(x0), "v"(x1), "v"(y0), "v"(y1), "n"(N), "n"(K)                                       \
  : "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113"

template <int K, int N, int PK, int MF = 0>
__global__ __launch_bounds__(512) void k(unsigned long long* __restrict__ bad, int iters) {
  __shared__ __attribute__((aligned(16))) u32x4 lds[512 * 8];
  const int tid = threadIdx.x, quarter = (tid & 63) >> 4;
  unsigned long long bad_store = 0, bad_prod = 0;
  const uint32_t a0 = (uint32_t)(uintptr_t)(lds) + tid * 16;
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  f16x8 fa, fb;
  for (int e = 0; e < 8; ++e) {
    fa[e] = (_Float16)(0.001f * (tid + e));
    fb[e] = (_Float16)(0.002f * (tid - e));
  }
  for (int it = 0; it < iters; ++it) {
    // MF matrix instructions in flight when the stores and the packed instruction issue (the loop back edge of the GEMM: five v_mfma_f32_32x32x16_f16 back to back)
    if (MF)
      asm volatile(".rept %c6\n\tv_mfma_f32_32x32x16_f16 %0, %4, %5, %0\n\tv_mfma_f32_32x32x16_f16 %1, %4, %5, %1\n\tv_mfma_f32_32x32x16_f16 %2, %4, %5, %2\n\t"
                   "v_mfma_f32_32x32x16_f16 %3, %4, %5, %3\n\t.endr"
                   : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) : "v"(fa), "v"(fb), "n"(MF));
    const uint32_t s = (uint32_t)(it * 2654435761u) ^ (uint32_t)(tid * 40503u + blockIdx.x);
    u32x4 pat = {s, s ^ 0x11111111u, s ^ 0x22222222u, s ^ 0x33333333u};
    uint32_t j0 = s, j1 = s + 7;
    const float x0 = 1.0f + (float)(s & 1023) * 0.001f, x1 = 2.0f + (float)((s >> 10) & 1023) * 0.001f, y0 = 0.5f + (float)((s >> 20) & 255) * 0.01f, y1 = 3.0f;
    float p0;
    if (PK == 2) asm volatile(PROBE_SETUP "v_pk_mul_f32 v[112:113], v[104:105], v[106:107] op_sel:[0,1]\n\ts_nop 7\n\ts_nop 7\n\tv_mov_b32 %2, v112" PROBE_OPERANDS);
    else if (PK == 3) asm volatile(PROBE_SETUP "v_pk_mul_f32 v[112:113], v[104:105], v[106:107]\n\ts_nop 7\n\ts_nop 7\n\tv_mov_b32 %2, v112" PROBE_OPERANDS);
    else if (PK) asm volatile(PROBE_SETUP "v_pk_mul_f32 v[100:101], v[104:105], v[106:107]\n\t" PROBE_TAIL PROBE_OPERANDS);
    else asm volatile(PROBE_SETUP "v_mul_f32 v100, v104, v106\n\tv_mul_f32 v101, v105, v107\n\t" PROBE_TAIL PROBE_OPERANDS);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    u32x4 got;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(got) : "v"(a0) : "memory");
    bad_store += (got[0] != s) | (got[1] != (s ^ 0x11111111u)) | (got[2] != (s ^ 0x22222222u)) | (got[3] != (s ^ 0x33333333u));
    bad_prod += (p0 != (PK == 2 ? x0 * y1 : x0 * y0));
    if (j0 == 0x12345u && j1 == 0x54321u) bad_store += 1;
  }
  float sink = 0.f;
  for (int j = 0; j < 4; ++j) sink += acc[j][0] + acc[j][7];
  if (sink == 1234.5f) bad_store += 1;
  if (bad_store) atomicAdd(bad + quarter, bad_store);
  if (bad_prod) atomicAdd(bad + 4 + quarter, bad_prod);
}

template <int K, int N, int PK, int MF = 0, int THREADS = 256>
void run(unsigned long long* bad, int blocks, int iters) {
  hipMemset(bad, 0, 64);
  hipLaunchKernelGGL((k<K, N, PK, MF>), dim3(blocks), dim3(THREADS), 0, 0, bad, iters);
  hipDeviceSynchronize();
  unsigned long long h[8];
  hipMemcpy(h, bad, 64, hipMemcpyDeviceToHost);
  printf("  %d MFMA x 4 in flight, %d waves / workgroup, %d store(s) queued, %s into the last store's data registers after %2d VALU: %.3g stores | wrong in LDS by lane quarter: %llu %llu %llu %llu | wrong low products: %llu %llu %llu %llu\n", MF, THREADS / 64, K,
         PK == 2 ? "v_pk_mul op_sel:[0,1] (other dest)" : PK == 3 ? "v_pk_mul no op_sel (other dest)  " : PK ? "v_pk_mul_f32" : "2 x v_mul_f32 ", N, (double)blocks * THREADS * iters, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
}

int main() {
  unsigned long long* bad;
  hipMalloc(&bad, 64);
  const int iters = 20000;
  for (int blocks : {256, 1024}) {
    printf("%s\n", blocks == 256 ? "256 workgroups:" : "1024 workgroups (two per CU: 64 KiB of LDS each):");
    run<1, 0, 2>(bad, blocks, iters); run<4, 0, 2>(bad, blocks, iters); run<8, 0, 2>(bad, blocks, iters); run<4, 1, 2>(bad, blocks, iters); run<4, 2, 2>(bad, blocks, iters); run<4, 8, 2>(bad, blocks, iters);
    run<4, 0, 3>(bad, blocks, iters); run<8, 0, 3>(bad, blocks, iters);
    run<4, 0, 2, 2>(bad, blocks, iters / 4); run<4, 0, 2, 2, 512>(bad, blocks / 2, iters / 4); run<8, 0, 2, 2, 512>(bad, blocks / 2, iters / 4); run<4, 0, 3, 2, 512>(bad, blocks / 2, iters / 4);
    run<4, 0, 1>(bad, blocks, iters); run<4, 0, 0>(bad, blocks, iters);
  }
  return 0;
}
