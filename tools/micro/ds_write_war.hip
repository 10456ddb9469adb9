// Round 5: how long after `ds_write_b128 addr, v[a:a+3]` has ISSUED does gfx950 still read v[a:a+3]?  hipcc guarantees two wait states before the next writer of a
// wide store's data registers; the GroupNorm fold of split_linear_h3l_kernel (four ds_write_b128 of weights, then four of activation rows, their data
// registers reused a few instructions later) wrote wrong rows for lanes 48-63 of random tiles (tools/gnfold_probe.py).  Here every wave stores a known
// pattern with K back-to-back ds_write_b128 (K = 1, 4, 8: a queue in front of the last store), overwrites the LAST store's data registers after N
// unrelated VALU instructions, reads the slot back and counts 16-byte values that are not the pattern.  Run with 1 and with 8 workgroups per CU.
// hipcc --offload-arch=gfx950 -O3 tools/micro/ds_write_war.hip -o tools/micro/bin/ds_write_war && tools/micro/bin/ds_write_war
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int K, int N, int ADDR>
__global__ __launch_bounds__(256) void k(unsigned long long* __restrict__ bad, int iters) {
  __shared__ __attribute__((aligned(16))) u32x4 lds[256 * 8];
  const int tid = threadIdx.x;
  unsigned long long nb = 0;
  const uint32_t a0 = (uint32_t)(uintptr_t)(lds) + tid * 16;
  for (int it = 0; it < iters; ++it) {
    const uint32_t s = (uint32_t)(it * 2654435761u) ^ (uint32_t)(tid * 40503u + blockIdx.x);
    u32x4 pat = {s, s ^ 0x11111111u, s ^ 0x22222222u, s ^ 0x33333333u};
    u32x4 fill = {~s, s + 1, s + 2, s + 3};
    uint32_t j0 = s, j1 = s + 7;
    // K - 1 stores of `fill` to other slots (queue), then the store under test of `pat` to slot 0
#define FILL(o) asm volatile("ds_write_b128 %0, %1 offset:" #o ::"v"(a0), "v"(fill) : "memory")
    if (K >= 8) { FILL(4096); FILL(8192); FILL(12288); FILL(16384); }
    if (K >= 4) { FILL(20480); FILL(24576); FILL(28672); }
#undef FILL
    // the store under test uses fixed registers v[100:103] (named in the clobber list) so that the rewrite can address ONE register of the quad
    if (ADDR)     // ADDR: the rewrite hits the store's ADDRESS register (v104) instead: it then points 64 KiB - 16 away (slot of the neighbour workgroup size... inside LDS: lds + 0x7000)
      asm volatile("v_mov_b32 v100, %3\n\tv_mov_b32 v101, %4\n\tv_mov_b32 v102, %5\n\tv_mov_b32 v103, %6\n\tv_mov_b32 v104, %2\n\ts_nop 4\n\t"
                   "ds_write_b128 v104, v[100:103]\n\t"
                   ".rept %c7\n\tv_xor_b32 %0, %0, %1\n\t.endr\n\t"
                   "v_add_u32 v104, 0x7000, v104"
                   : "+v"(j0), "+v"(j1) : "v"(a0), "v"(pat[0]), "v"(pat[1]), "v"(pat[2]), "v"(pat[3]), "n"(N) : "memory", "v100", "v101", "v102", "v103", "v104");
    else
    asm volatile("v_mov_b32 v100, %3\n\tv_mov_b32 v101, %4\n\tv_mov_b32 v102, %5\n\tv_mov_b32 v103, %6\n\ts_nop 4\n\t"
                 "ds_write_b128 %2, v[100:103]\n\t"
                 ".rept %c7\n\tv_xor_b32 %0, %0, %1\n\t.endr\n\t"
                 "v_mov_b32 v100, 0xdeadbeef\n\tv_mov_b32 v103, 0xdeadbeef"
                 : "+v"(j0), "+v"(j1) : "v"(a0), "v"(pat[0]), "v"(pat[1]), "v"(pat[2]), "v"(pat[3]), "n"(N) : "memory", "v100", "v101", "v102", "v103");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    u32x4 got;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(got) : "v"(a0) : "memory");
    nb += (got[0] != s) | (got[1] != (s ^ 0x11111111u)) | (got[2] != (s ^ 0x22222222u)) | (got[3] != (s ^ 0x33333333u));
    if (j0 == 0x12345u && j1 == 0x54321u) nb += 1;
  }
  if (nb) atomicAdd(bad, nb);
}

template <int K, int N, int ADDR = 0>
void run(unsigned long long* bad, int blocks, int iters) {
  hipMemset(bad, 0, 8);
  hipLaunchKernelGGL((k<K, N, ADDR>), dim3(blocks), dim3(256), 0, 0, bad, iters);
  hipDeviceSynchronize();
  unsigned long long h = 0;
  hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
  printf("  %d store(s) queued, %s register rewritten after %2d VALU: blocks %5d  stores %12.0f  corrupted: %llu\n", K, ADDR ? "ADDRESS" : "data", N, blocks, (double)blocks * 256.0 * iters, h);
}

int main() {
  unsigned long long* bad;
  hipMalloc(&bad, 8);
  const int iters = 20000;
  for (int blocks : {256, 1024}) {
    printf("%s\n", blocks == 256 ? "one workgroup per CU:" : "four workgroups per CU (LDS limit: 32 KiB each):");
    run<1, 0>(bad, blocks, iters); run<1, 2>(bad, blocks, iters); run<1, 8>(bad, blocks, iters);
    run<4, 0>(bad, blocks, iters); run<4, 2>(bad, blocks, iters); run<4, 8>(bad, blocks, iters); run<4, 32>(bad, blocks, iters);
    run<8, 0>(bad, blocks, iters); run<8, 2>(bad, blocks, iters); run<8, 8>(bad, blocks, iters); run<8, 32>(bad, blocks, iters); run<8, 64>(bad, blocks, iters);
    run<1, 0, 1>(bad, blocks, iters); run<1, 2, 1>(bad, blocks, iters); run<4, 0, 1>(bad, blocks, iters); run<4, 2, 1>(bad, blocks, iters); run<8, 0, 1>(bad, blocks, iters);
    run<8, 2, 1>(bad, blocks, iters); run<8, 8, 1>(bad, blocks, iters); run<8, 32, 1>(bad, blocks, iters);
  }
  return 0;
}
