// EXPERIMENT (tools only, librba_tune.so; measured in profiles/r04_k6_ws.txt and NOT adopted -- see the result at the end of this comment).
// K6, weights-stationary form of the f16x3 Linear (round 4, late) -- for the Linears whose K fits one CU's LDS: K <= 512 at 64 output columns
// (reference: Mlp.fc1 and WindowAttention.proj of backbone/swin.py:35-41, 131-171 at Swin stages 1-3).
//
// Why.  What a stage-3 fc1 launch of the pipelined kernel loses (profiles/r04_k6_fc1_timeline.txt) is not in its k loop (MFMAs issue 85 % of the time) but
// around it: every tile pays a prologue and a GELU + split epilogue during which the matrix pipe of the CU idles, because all waves of a CU walk the same
// barrier-synchronised weight ring and so reach their epilogues TOGETHER (2 x (1.3 + 5.9) of 55.8 us), and a launch has only two tiles per CU to amortise a
// de-phasing ramp over.  Here nothing synchronises the waves after a one-off prologue:
//   * a workgroup = one CU, sixteen (or twelve) waves; it owns ONE 64-column panel of the weight for the whole launch: the packed (h, l) image of the panel,
//     K x 64 x 4 bytes (128 KB at K = 512), is copied to LDS once;
//   * the rows are cut into 32-row wave tiles; a wave takes the next tile of its workgroup's row group from an LDS counter, streams the tile's split
//     A operand straight into registers (the producer's fragment image: four contiguous 1 KiB wave loads per 32-wide block, one block ahead), reads the weight
//     fragments from LDS and runs the block's 12 MFMAs; no barrier, no staging stores in the loop; 64 accumulator registers (32 x 64, main + low);
//   * the waves of a SIMD start their first tiles `stagger` apart, so that from then on one of them is in its epilogue (VALU, stores) while the others keep
//     the matrix pipe busy; this file is built WITHOUT packed fp32 (build.py: packed fp32 executes on the matrix pipe's datapath and would stall the other
//     waves' MFMAs -- profiles/r03_mfma_valu_overlap.txt -- plain VALU runs in their shadow);
//   * workgroup -> (row group, panel) keeps the CUs of one XCD on the same rows: their A operand (2 MB per 1 024 rows at K = 512) is served by that XCD's L2.
// Same products in the same order per accumulator as split_linear_h3p_kernel: fp32 outputs bit-identical (checked on every launch form; the split-image output differs
// in 6e-5 of its words by an equivalent (h + 1 ulp, l - 2048 ulp) representation: the un-packed build rounds one fp32 of the GELU differently).
//
// RESULT (tools/k6_ws_ab.py, MI355X, product launch forms on the same operands): stage-3 fc1 (8192 x 2048 x 512, GELU + split out) 62.4 us pipelined -> 71.3 us here;
// 8192 x 1024 x 512 fp32 out 29.4 -> 33.5; stage-3 proj 23.3 -> 22.9; stage-2 fc1 (K = 256) 78.9 -> 70.1 warm, 80.6 -> 78.9 cold; stage-1 proj 36.2 -> 34.8 / 44.0 -> 40.2.
// The stagger (0 ... 6 us between the waves of a SIMD, by HW_ID) changes nothing: the waves are NOT waiting for each other's epilogues.  What bounds the form is the
// operand stream: a 64-column panel re-reads the activations N / 64 times -- 537 MB per stage-3 fc1 launch through L2 -> L1 at 7.5-8 TB/s, the same fabric rate every
// LDS-DMA formulation of round 2 ended at -- twice the pipelined kernel's A traffic, and the weight traffic it saves was the smaller share.  A 128-column panel
// (half the traffic) needs 256 KB of LDS at K = 512.  Kept for the record; the product keeps the pipelined kernel.
#define RBA_H3_HELPERS_ONLY
#include "../split_linear_h3.h"

// tools / tests: 0 = never, 1 = by rule (ws_applies), 2 = wherever legal
extern "C" __attribute__((visibility("default"))) int rba_k6_ws = 1;
// ticks of the 100 MHz clock between the first tiles of the waves that share a SIMD
extern "C" __attribute__((visibility("default"))) int rba_k6_ws_stagger = 300;

namespace {

enum { WS_F32 = 0, WS_RES = 1, WS_SPLIT = 2 };

template <int MODE, int ACT, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void split_linear_ws_kernel(const char* __restrict__ A, const u32x4_t* __restrict__ Wp,
                                                                      const float* __restrict__ bias, const float* R, void* out, int M, int N, int K,
                                                                      int P, int TPG, int stagger) {
  extern __shared__ __attribute__((aligned(16))) u32x4_t wl[];                     // [K / 32][g][plane][64 rows][2 slots] x 16 B, then the tile counter
  constexpr bool FOUT = MODE == WS_SPLIT;
  const int NB = K >> 5, S16 = K >> 4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  // workgroup -> (panel, row group): consecutive workgroup ids go to consecutive XCDs, so id % 8 is the XCD; the gridDim.x / 8 workgroups of an XCD
  // take P panels x (gridDim.x / 8 / P) row groups of one contiguous range of rows
  const int id = blockIdx.x, xcd = id & 7, c = id >> 3;
  const int gpx = (int)(gridDim.x >> 3) / P;                                       // row groups per XCD
  const int panel = c % P, rg = xcd * gpx + c / P;
  const int MT32 = (M + 31) >> 5;
  const int t_lo = rg * TPG;
  int* counter = reinterpret_cast<int*>(wl + NB * 512);

  int* simd_of = counter + 1;                                                      // [WAVES]
  {  // the panel's weight image -> LDS: per (block, g, plane) one contiguous 2 KB run of the packed tile; every load in flight before the first store
    const u32x4_t* src = Wp + (int64_t)(panel >> 1) * S16 * 512 + (panel & 1) * 128;
    constexpr int UMAX = (16 * 512 + 64 * WAVES - 1) / (64 * WAVES);                // K <= 512
    u32x4_t tmp[UMAX];
#pragma unroll
    for (int i = 0; i < UMAX; ++i) {
      const int u = tid + i * 64 * WAVES;
      const int uc = u < NB * 512 ? u : 0;
      const int b = uc >> 9, g = (uc >> 8) & 1, pl = (uc >> 7) & 1, r2 = uc & 127;
      tmp[i] = src[(2 * b + g) * 512 + pl * 256 + r2];
    }
#pragma unroll
    for (int i = 0; i < UMAX; ++i) {
      const int u = tid + i * 64 * WAVES;
      if (u < NB * 512) wl[u] = tmp[i];
    }
    if (tid == 0) *counter = 0;
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if (lane == 0) simd_of[wave] = (int)((hw >> 4) & 3);                           // HW_ID[5:4] = the SIMD this wave runs on
  }
  __syncthreads();
  if (stagger > 0) {                                                               // the k-th wave of a SIMD starts k * stagger ticks late
    const int mine = simd_of[wave];
    int k = 0;
    for (int w2 = 0; w2 < wave; ++w2) k += simd_of[w2] == mine ? 1 : 0;
    k = __builtin_amdgcn_readfirstlane(k);
    if (k > 0) {
      const unsigned long long t0 = wall_clock64(), d = (unsigned long long)stagger * (unsigned)k;
      while (wall_clock64() - t0 < d) __builtin_amdgcn_s_sleep(8);
    }
  }

  const int fb = l31 * 2 + (lh ^ ((l31 >> 3) & 1));
  auto mm = [](const f16x8_t a, const f16x8_t b, const f32x16_t c) {
    return FOUT ? __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c, 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  };

  for (;;) {
    int t = 0;
    if (lane == 0) t = atomicAdd(counter, 1);
    t = __builtin_amdgcn_readfirstlane(t);
    const int wt = t_lo + t;
    if (t >= TPG || wt >= MT32) break;
    const char* fbase = A + ((int64_t)wt * NB) * 4096 + lane * 16;
    f32x16_t accm[2], accl[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) accm[j][r] = accl[j][r] = 0.f;
    f16x8_t pa[3][4];                                                              // [block % 3][h g0, l g0, h g1, l g1]: loads two blocks ahead
    auto aload = [&](int b, f16x8_t (&d)[4]) {
      const char* s = fbase + (b < NB ? b : NB - 1) * 4096;
#pragma unroll
      for (int q = 0; q < 4; ++q) d[q] = *reinterpret_cast<const f16x8_t*>(s + q * 1024);
    };
    auto block = [&](int b, const f16x8_t (&a)[4]) {
      const u32x4_t* img = wl + b * 512 + fb;
      f16x8_t bh0[2], bl0[2], bh1[2], bl1[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        bh0[j] = __builtin_bit_cast(f16x8_t, img[64 * j]);
        bl0[j] = __builtin_bit_cast(f16x8_t, img[128 + 64 * j]);
        bh1[j] = __builtin_bit_cast(f16x8_t, img[256 + 64 * j]);
        bl1[j] = __builtin_bit_cast(f16x8_t, img[384 + 64 * j]);
      }
      // per accumulator the order of split_linear_h3p_kernel (main: h0 h0, h1 h1; low: h0 l0, l0 h0, h1 l1, l1 h1); the two column tiles alternate
#pragma unroll
      for (int j = 0; j < 2; ++j) accm[j] = mm(a[0], bh0[j], accm[j]);
#pragma unroll
      for (int j = 0; j < 2; ++j) accl[j] = mm(a[0], bl0[j], accl[j]);
#pragma unroll
      for (int j = 0; j < 2; ++j) accm[j] = mm(a[2], bh1[j], accm[j]);
#pragma unroll
      for (int j = 0; j < 2; ++j) accl[j] = mm(a[1], bh0[j], accl[j]);
#pragma unroll
      for (int j = 0; j < 2; ++j) accl[j] = mm(a[2], bl1[j], accl[j]);
#pragma unroll
      for (int j = 0; j < 2; ++j) accl[j] = mm(a[3], bh1[j], accl[j]);
    };
    aload(0, pa[0]);
    aload(1, pa[1]);
    int b = 0;
    for (; b + 3 <= NB; b += 3) {
      aload(b + 2, pa[2]);
      block(b, pa[0]);
      aload(b + 3, pa[0]);
      block(b + 1, pa[1]);
      aload(b + 4, pa[1]);
      block(b + 2, pa[2]);
    }
    if (b < NB) {                                                                  // NB % 3 = 1 or 2 (uniform)
      block(b, pa[0]);
      if (b + 1 < NB) block(b + 1, pa[1]);
    }
    const int m0 = wt * 32, n0 = panel * 64;
    if (MODE == WS_SPLIT) h3_epilogue_split<ACT, 2>(accm, accl, bias, out, M, N, m0, n0, 0, l31, lh);
    else h3_epilogue<ACT, 2, 0, MODE == WS_RES>(accm, accl, bias, reinterpret_cast<float*>(out), R, M, N, m0, n0, 32, 64, 0, l31, lh);
  }
}

template <int MODE, int ACT>
int ws_launch(const void* xf, const u32x4_t* wp, const float* bias, const float* res, void* out, int64_t M, int N, int K, hipStream_t st) {
  constexpr int WAVES = 16;
  const int P = N / 64, WG = 256;
  const int G = WG / P, MT32 = (int)((M + 31) >> 5), TPG = (MT32 + G - 1) / G;
  const size_t shm = (size_t)(K >> 5) * 8192 + 16 + 4 * WAVES;
  auto kern = split_linear_ws_kernel<MODE, ACT, WAVES>;
  static bool attr = false;                                                        // 128 KB of dynamic LDS: above the default limit
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256) != hipSuccess) return -1;
    attr = true;
  }
  hipLaunchKernelGGL(kern, dim3(WG), dim3(64 * WAVES), shm, st, reinterpret_cast<const char*>(xf), wp, bias, res, out, (int)M, N, K, P, TPG,
                     rba_k6_ws_stagger);
  return 0;
}

}  // namespace

// Where the form applies: split-image A operand, K a multiple of 64 up to 512, N / 64 panels dividing the 32 CUs of an XCD, and enough 32-row wave tiles that
// every CU's waves have work (at least 8 per CU); by rule (rba_k6_ws == 1) only where the pipelined kernel runs at least one full round of 128 x 128 tiles.
extern "C" bool rba_ws_applies(int64_t M, int N, int K) {
  if (rba_k6_ws == 0 || (K & 63) || K < 64 || K > 512 || (N & 63)) return false;
  const int P = N / 64;
  if (P < 1 || P > 32 || (32 % P)) return false;
  const int64_t MT32 = (M + 31) >> 5;
  if (rba_k6_ws == 2) return MT32 >= 256 / P;
  return MT32 * P >= 256 * 8 && ((M + 127) / 128) * ((N + 127) / 128) >= 256;
}

// mode 0: fp32 rows out (act 0 / 1 / 2), 1: out = (residual + x W^T) + bias, 2: GELU(x W^T + bias) written as the next Linear's split image
extern "C" int rba_ws_launch(int mode, int act, const void* x_frag, const void* wp, const float* bias, const float* res,
                                                                  void* out, int64_t M, int N, int K, void* stream) {
  const u32x4_t* w = reinterpret_cast<const u32x4_t*>(wp);
  hipStream_t st = (hipStream_t)stream;
  rba_begin();
  if (mode == 2) return ws_launch<WS_SPLIT, 1>(x_frag, w, bias, nullptr, out, M, N, K, st);
  if (mode == 1) return ws_launch<WS_RES, 0>(x_frag, w, bias, res, out, M, N, K, st);
  if (act == 1) return ws_launch<WS_F32, 1>(x_frag, w, bias, nullptr, out, M, N, K, st);
  if (act == 2) return ws_launch<WS_F32, 2>(x_frag, w, bias, nullptr, out, M, N, K, st);
  return ws_launch<WS_F32, 0>(x_frag, w, bias, nullptr, out, M, N, K, st);
}
