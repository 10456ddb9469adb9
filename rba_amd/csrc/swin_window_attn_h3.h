// K5, f16x3 form (head_dim 32): the same workgroup-per-(window, head) kernel as swin_window_attn_mfma_kernel with both contractions
// on v_mfma_f32_16x16x32_f16 -- three f16 MFMAs per fp32 product (split_linear_h3.h) instead of exact-fp32 MFMAs at 1/16 of the rate:
//   S^T = K . Q^T   one 32-deep step per key tile: A = K fragment (8 contiguous d of one key: ONE ds_read_b128 per piece),
//                   B = Q^T (registers, split once per strip); 3 MFMAs of 16 cycles instead of 8 of 32
//   O   = P . V     32 keys per step (two key tiles): A = P, which the lane already holds (the S^T accumulator layout IS the A layout,
//                   as in the fp32 kernel), B = V fetched from its row-major [key][d] LDS image with the TRANSPOSING read
//                   ds_read_b64_tr_b16 (a 16-lane group reads 4 keys x 16 d and every lane receives its d column)
// Pieces: K, Q and P's first factor use the unscaled residual l = f16(x - h) into ONE accumulator (a softmax only sees absolute
// errors of the scores: <= 32 |q| 2^-25 from f16's subnormal floor); the P . V product uses scaled residuals (x 2^11) and a second
// accumulator like the Linear kernel, so the output keeps 22 bits whatever V's magnitude.  |q|, |k|, |v| < 65504.
// Measured (tools/k5_sweep.py): matrix work is ~40 % of the fp32-MFMA kernel (ablation builds in DESIGN.md section 7).
// LDS: four f16 planes [NP keys][32 d] with 64-byte rows: K (h, l) with the 16-byte chunk c of key k at c ^ P[(k >> 2) & 3],
// P = {0, 2, 3, 1} (conflict-free ds_read_b128 for the row-per-lane fragment); V (h, l) with the 32-byte half d / 16 at
// (d / 16) ^ ((k >> 2) & 1) (conflict-free transposing reads).
#pragma once

// Timing build (tools only, csrc/tune/k5_timing.hip defines K5H_TIMING before including this header): every wave's lane 0 stamps the
// 100 MHz wall clock at its phase boundaries into dbg[(workgroup * WAVES + wave) * 12 + i].  The product build compiles none of it.
#ifdef K5H_TIMING
#define K5H_DBG_PARAM , unsigned long long* __restrict__ dbg
#define K5H_STAMP(i)                                                                                                              \
  do {                                                                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                                            \
    if ((threadIdx.x & 63) == 0)                                                                                                  \
      dbg[((((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * WAVES + (threadIdx.x >> 6)) * 12 + (i)] = wall_clock64(); \
    __builtin_amdgcn_sched_barrier(0);                                                                                            \
  } while (0)
#define K5H_WAIT_VM() __builtin_amdgcn_s_waitcnt(0x0070 | 0x0F00 | 0xC000 * 0)
#define K5H_SINK(x) asm volatile("" ::"s"(__builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, (x)))))
#elif defined(K5H_ABLATE)
// Ablation build (tools only, csrc/tune/k5_wpe_ab.hip): bit 0 = no bias-fragment loads, bit 1 = K / V / Q rows all read from the qkv bias
// (no gather traffic), bit 2 = no stores.  Wrong results by construction; launch time only.
#define K5H_DBG_PARAM , int ablate
#define K5H_STAMP(i)
#define K5H_WAIT_VM()
#define K5H_SINK(x)
#else
#define K5H_DBG_PARAM
#define K5H_STAMP(i)
#define K5H_WAIT_VM()
#define K5H_SINK(x)
#endif

namespace {

typedef _Float16 k5h_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 k5h_f16x2 __attribute__((ext_vector_type(2)));
typedef __fp16 k5h_h4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef uint32_t k5h_u32x4 __attribute__((ext_vector_type(4)));

// (a, b) -> packed f16 h and the packed f16 residual l = f16((x - h) * SC), SC = 1 (unscaled) or 2048
template <bool SCALED>
__device__ __forceinline__ void k5h_split2(float a, float b, uint32_t& h, uint32_t& l) {
  const k5h_f16x2 hv = {(_Float16)a, (_Float16)b};
  h = __builtin_bit_cast(uint32_t, hv);
  const float m = SCALED ? -2048.0f : -1.0f;
  const float a2 = SCALED ? a * 2048.0f : a, b2 = SCALED ? b * 2048.0f : b;
  uint32_t r;
  asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(m), "v"(a2));
  asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(h), "v"(m), "v"(b2));
  l = r;
}

template <int NT, int WAVES, bool FRAG, bool SOUT = false, int WPE = 5>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(WPE, 8))) void swin_window_attn_h3_kernel(
    const float* __restrict__ qkv, const float* __restrict__ qkv_bias, const float* __restrict__ bias, float* __restrict__ out,
    int H, int W, int Hp, int Wp, int nH, int ws, int shift, float scale K5H_DBG_PARAM) {
  K5H_STAMP(0);
#ifdef K5H_ABLATE
  // bits 3..: a late start for every second round of 256 workgroups, (ablate >> 3) * 0.5 us (do two co-resident workgroups run phase-locked?)
  if (((ablate >> 3) & 0xff) && ((((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) >> 8) & 1)) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)(((ablate >> 3) & 0xff) * 50)) __builtin_amdgcn_s_sleep(4);
  }
#endif
  constexpr int HD = 32, NP = NT * 16, PL = NP * 64;                         // bytes per f16 plane
  // NT = 9 is only ever launched for 12 x 12 windows (144 tokens = 9 full key tiles): a compile-time window size turns the token -> (row,
  // column) divisions of the gather into multiply-shifts and removes the "key beyond the window" tests of the softmax (round 3: K5 is
  // bound by the instructions it issues -- ~9 waves per SIMD per stage-3 launch at ~600 VALU instructions each -- not by the matrix pipe)
  if (NT == 9) ws = 12;
  extern __shared__ __attribute__((aligned(16))) unsigned char k5h_lds[];
  unsigned char* Kh = k5h_lds;                                                  // + PL: Kl; + 2 PL: Vh; + 3 PL: Vl
  int* tok = reinterpret_cast<int*>(k5h_lds + 4 * PL);
  int* rid = tok + NP;
  const int N = NT == 9 ? 144 : ws * ws;
  int wx = blockIdx.x, wy = blockIdx.y, hz = blockIdx.z;
#ifdef K5H_ABLATE
  if (ablate & 0x4000) {   // heads vary fastest over the dispatch order: linear id -> (image, window row, window column, head)
    const int L = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    const int hh = L % nH, wl = L / nH;
    wx = wl % (int)gridDim.x;
    wy = (wl / (int)gridDim.x) % (int)gridDim.y;
    hz = (wl / (int)(gridDim.x * gridDim.y)) * nH + hh;
  }
#endif
  // the shift mask (swin.py:413-440) only separates tokens inside the LAST row / column of windows: everywhere else all 144 tokens share
  // region 0 and the 36 compare / add pairs per lane are skipped (workgroup-uniform branch)
  const bool need_mask = shift > 0 && (wy == (int)gridDim.y - 1 || wx == (int)gridDim.x - 1);
  const int h = hz % nH, b = hz / nH;
  const int C = nH * HD;
  const int64_t tok_stride = 3 * (int64_t)C;
  const float* qkv_b = qkv + (int64_t)b * H * W * tok_stride;
  const float* qb = qkv_bias + h * HD;

  // ---- gather K, V of the window's N tokens into the f16 planes.  All global loads of a thread are issued before the first is
  // consumed (a rolled loop pays the memory latency once per trip), and the Q fragment of the wave's first strip is requested
  // here too, so that it arrives under the gather instead of in front of the first MFMA.
  constexpr int NIT = (NP * (HD / 4) + 64 * WAVES - 1) / (64 * WAVES);
  float4 kk4[NIT], vv4[NIT];
  int tkv[NIT], rgv[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = threadIdx.x + it * 64 * WAVES;
    const int t = i >> 3, d4 = i & 7;
    kk4[it] = vv4[it] = make_float4(0.f, 0.f, 0.f, 0.f);
    tkv[it] = -2;
    rgv[it] = -1;
    if (i < NP * (HD / 4) && t < N) {
      const int r = wy * ws + t / ws, c = wx * ws + t % ws;
      int rr = r + shift, cc = c + shift;
      rr = rr >= Hp ? rr - Hp : rr;
      cc = cc >= Wp ? cc - Wp : cc;
#ifdef K5H_ABLATE
      if (ablate & 2) rr = H;
#endif
      if (rr < H && cc < W) {
        tkv[it] = rr * W + cc;
        const float* p = qkv_b + (int64_t)tkv[it] * tok_stride + h * HD + d4 * 4;
        kk4[it] = *reinterpret_cast<const float4*>(p + C);
        vv4[it] = *reinterpret_cast<const float4*>(p + 2 * C);
      } else {
        tkv[it] = -1;
        kk4[it] = *reinterpret_cast<const float4*>(qb + C + d4 * 4);
        vv4[it] = *reinterpret_cast<const float4*>(qb + 2 * C + d4 * 4);
      }
      const int hid = r < Hp - ws ? 0 : (r < Hp - shift ? 1 : 2);
      const int wid = c < Wp - ws ? 0 : (c < Wp - shift ? 1 : 2);
      rgv[it] = hid * 3 + wid;
    }
  }
  // Q rows of this wave's first strip (token index by the same arithmetic as above: tok[] is not written yet)
  float4 q_a = make_float4(0.f, 0.f, 0.f, 0.f), q_b = q_a;
  {
    const int qt0 = (threadIdx.x >> 6) * 16 + (threadIdx.x & 15), kq = (threadIdx.x & 63) >> 4;
    if ((threadIdx.x >> 6) < NT && qt0 < N) {
      const int r = wy * ws + qt0 / ws, c = wx * ws + qt0 % ws;
      int rr = r + shift, cc = c + shift;
      rr = rr >= Hp ? rr - Hp : rr;
      cc = cc >= Wp ? cc - Wp : cc;
#ifdef K5H_ABLATE
      if (ablate & 2) rr = H;
#endif
      const float4* p = (rr < H && cc < W) ? reinterpret_cast<const float4*>(qkv_b + (int64_t)(rr * W + cc) * tok_stride + h * HD + kq * 8)
                                           : reinterpret_cast<const float4*>(qb + kq * 8);
      q_a = p[0];
      q_b = p[1];
    }
  }
  K5H_STAMP(1);                                                                 // all global loads of the gather issued
  K5H_WAIT_VM();
  K5H_STAMP(2);                                                                 // ... and arrived
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = threadIdx.x + it * 64 * WAVES;
    if (i >= NP * (HD / 4)) break;
    const int t = i >> 3, d4 = i & 7;
    uint32_t h0, l0, h1, l1;
    const int tq = (t >> 2) & 3;
    const int perm = (0x1320 >> (4 * tq)) & 3;                                  // P = {0, 2, 3, 1}
    k5h_split2<false>(kk4[it].x, kk4[it].y, h0, l0);
    k5h_split2<false>(kk4[it].z, kk4[it].w, h1, l1);
    const int ko = t * 64 + (((d4 >> 1) ^ perm) * 16) + (d4 & 1) * 8;
    *reinterpret_cast<uint2*>(Kh + ko) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(Kh + PL + ko) = make_uint2(l0, l1);
    k5h_split2<true>(vv4[it].x, vv4[it].y, h0, l0);
    k5h_split2<true>(vv4[it].z, vv4[it].w, h1, l1);
    const int vo = t * 64 + (((d4 >> 2) ^ (tq & 1)) * 32) + (d4 & 3) * 8;
    *reinterpret_cast<uint2*>(Kh + 2 * PL + vo) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(Kh + 3 * PL + vo) = make_uint2(l0, l1);
    if (d4 == 0) { tok[t] = tkv[it]; rid[t] = rgv[it]; }
  }
  K5H_STAMP(3);                                                                 // split + LDS writes issued
  __syncthreads();
  K5H_STAMP(4);                                                                 // barrier passed

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, kk = lane >> 4;
  const bool vec_bias = (N & 3) == 0;
  // K fragment of key tile c: key c * 16 + l15, chunk kk -> byte offset (+ c * 1024)
  const int kfrag = l15 * 64 + ((kk ^ ((0x1320 >> (4 * ((l15 >> 2) & 3))) & 3)) * 16);
  // V transposing read of key tile c, d tile dt: rows c * 16 + 4 kk + (l15 >> 2) -> (+ c * 1024 + (dt ^ (kk & 1)) * 32)
  const int vfrag = (4 * kk + (l15 >> 2)) * 64 + (l15 & 3) * 8;

  for (int strip = wave; strip < NT; strip += WAVES) {
    const int qt = strip * 16 + l15;
    const int qtok = tok[qt];
    k5h_f16x8 qh, ql;
    {
      float4 a = q_a, c4 = q_b;                                                  // first strip: requested before the gather
      if (strip != wave) {
        a = c4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (qtok >= 0) {
          const float4* p = reinterpret_cast<const float4*>(qkv_b + (int64_t)qtok * tok_stride + h * HD + kk * 8);
          a = p[0]; c4 = p[1];
        } else if (qtok == -1) {
          const float4* p = reinterpret_cast<const float4*>(qb + kk * 8);
          a = p[0]; c4 = p[1];
        }
      }
      k5h_u32x4 hq, lq;
      uint32_t x0, x1;
      k5h_split2<false>(a.x * scale, a.y * scale, x0, x1); hq[0] = x0; lq[0] = x1;
      k5h_split2<false>(a.z * scale, a.w * scale, x0, x1); hq[1] = x0; lq[1] = x1;
      k5h_split2<false>(c4.x * scale, c4.y * scale, x0, x1); hq[2] = x0; lq[2] = x1;
      k5h_split2<false>(c4.z * scale, c4.w * scale, x0, x1); hq[3] = x0; lq[3] = x1;
      qh = __builtin_bit_cast(k5h_f16x8, hq);
      ql = __builtin_bit_cast(k5h_f16x8, lq);
    }
    // ---- S^T tiles: lane holds S[key = c*16 + 4*kk + r][query = qt].  The relative-position bias is the INITIAL value of the
    // accumulator: its loads are issued ahead of the MFMAs that consume them instead of in front of the softmax that waits for them.
    const float* brow = bias + ((int64_t)h * N + (qt < N ? qt : 0)) * N;
    // FRAG: bias pre-permuted to [nH][strip][c][lane][4] (rba_swin_bias_fragments_f32): 1 KiB coalesced per load
    const float4* bfrag = reinterpret_cast<const float4*>(bias) + (((int64_t)h * NT + strip) * NT) * 64 + lane;
    f32x4_t S[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c) {
      const int k0i = c * 16 + kk * 4;
      f32x4_t a0 = {0.f, 0.f, 0.f, 0.f};
      if (FRAG) {
#ifdef K5H_ABLATE
        if (!(ablate & 1))
#endif
        {
          const float4 t4 = bfrag[c * 64];
          a0 = (f32x4_t){t4.x, t4.y, t4.z, t4.w};
        }
      } else if (qt < N) {
        if (vec_bias && k0i + 3 < N) {
          const float4 t4 = *reinterpret_cast<const float4*>(brow + k0i);
          a0 = (f32x4_t){t4.x, t4.y, t4.z, t4.w};
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (k0i + r < N) a0[r] = brow[k0i + r];
        }
      }
      S[c] = a0;
    }
#ifdef K5H_TIMING
    K5H_WAIT_VM();
    K5H_SINK((float)qh[0]);
    K5H_STAMP(5);                                                               // Q split, bias fragments arrived
#endif
    // three sweeps over the key tiles instead of three dependent MFMAs per tile: consecutive MFMAs then write DIFFERENT accumulators (no
    // dependent-issue stalls), every accumulator still receives its products in the order kh.qh, kh.ql, kl.qh (bit-identical)
    {
      k5h_f16x8 kf[NT];
#pragma unroll
      for (int c = 0; c < NT; ++c) kf[c] = __builtin_bit_cast(k5h_f16x8, *reinterpret_cast<const k5h_u32x4*>(Kh + c * 1024 + kfrag));
#pragma unroll
      for (int c = 0; c < NT; ++c) S[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[c], qh, S[c], 0, 0, 0);
#pragma unroll
      for (int c = 0; c < NT; ++c) S[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[c], ql, S[c], 0, 0, 0);
#pragma unroll
      for (int c = 0; c < NT; ++c) kf[c] = __builtin_bit_cast(k5h_f16x8, *reinterpret_cast<const k5h_u32x4*>(Kh + PL + c * 1024 + kfrag));
#pragma unroll
      for (int c = 0; c < NT; ++c) S[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[c], qh, S[c], 0, 0, 0);
    }
    K5H_SINK(S[NT - 1][3]);
    K5H_STAMP(6);                                                               // Q K^T done
    // ---- shift mask, padding keys; row max
    float m = -INFINITY;
    if (need_mask) {
      const int myrid = rid[qt];
#pragma unroll
      for (int c = 0; c < NT; ++c) {
        const int k0i = c * 16 + kk * 4;
        const int4 kr4 = *reinterpret_cast<const int4*>(rid + k0i);
        const int krid[4] = {kr4.x, kr4.y, kr4.z, kr4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (krid[r] != myrid) S[c][r] += -100.0f;
      }
    }
#pragma unroll
    for (int c = 0; c < NT; ++c) {
      const int k0i = c * 16 + kk * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = S[c][r];
        if (NT != 9 && k0i + r >= N) v = -INFINITY;
        S[c][r] = v;
        m = fmaxf(m, v);
      }
    }
    m = fmaxf(m, __shfl_xor(m, 16, RBA_WAVE));
    m = fmaxf(m, __shfl_xor(m, 32, RBA_WAVE));
    // exp(s - m) = exp2(s log2(e) - m log2(e)): one packed fma per PAIR of scores + v_exp_f32 each; the row sum in packed pairs too (this
    // kernel is bound by the vector instructions it issues, and its MFMAs are short dependent chains with nothing to hide under)
    const float mneg = -m * 1.44269504088896340736f;
    f32x2 ls2 = {0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NT; ++c)
#pragma unroll
      for (int r = 0; r < 4; r += 2) {
        const f32x2 t = (f32x2){S[c][r], S[c][r + 1]} * 1.44269504088896340736f + mneg;
        const f32x2 pp = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
        S[c][r] = pp.x;
        S[c][r + 1] = pp.y;
        ls2 += pp;
      }
    float lsum = ls2.x + ls2.y;
    lsum += __shfl_xor(lsum, 16, RBA_WAVE);
    lsum += __shfl_xor(lsum, 32, RBA_WAVE);
    K5H_SINK(lsum);
    K5H_STAMP(7);                                                               // softmax done
    // ---- O = P . V: 32 keys (two key tiles) per step, two 16-wide d tiles, main + low accumulators
    f32x4_t Om[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, Ol[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int c0 = 0; c0 < NT; c0 += 2) {
      const bool two = c0 + 1 < NT;
      const int c1 = two ? c0 + 1 : c0;                                       // odd NT: the missing tile's P is zero, its V rows are c0's
      k5h_u32x4 ph, pl;
      uint32_t x0, x1;
      k5h_split2<true>(S[c0][0], S[c0][1], x0, x1); ph[0] = x0; pl[0] = x1;
      k5h_split2<true>(S[c0][2], S[c0][3], x0, x1); ph[1] = x0; pl[1] = x1;
      if (two) {
        k5h_split2<true>(S[c1][0], S[c1][1], x0, x1); ph[2] = x0; pl[2] = x1;
        k5h_split2<true>(S[c1][2], S[c1][3], x0, x1); ph[3] = x0; pl[3] = x1;
      } else {
        ph[2] = ph[3] = pl[2] = pl[3] = 0u;
      }
      const k5h_f16x8 pa = __builtin_bit_cast(k5h_f16x8, ph), pb = __builtin_bit_cast(k5h_f16x8, pl);
      k5h_f16x8 vh[2], vl[2];
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const int half = (dt ^ (kk & 1)) * 32;
        const unsigned char* v0 = Kh + 2 * PL + c0 * 1024 + vfrag + half;
        const unsigned char* v1 = Kh + 2 * PL + c1 * 1024 + vfrag + half;
        const k5h_h4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) k5h_h4*)(v0));
        const k5h_h4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) k5h_h4*)(v1));
        const k5h_h4 r2 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) k5h_h4*)(v0 + PL));
        const k5h_h4 r3 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) k5h_h4*)(v1 + PL));
        const uint2 u0 = __builtin_bit_cast(uint2, r0), u1 = __builtin_bit_cast(uint2, r1);
        const uint2 u2 = __builtin_bit_cast(uint2, r2), u3 = __builtin_bit_cast(uint2, r3);
        vh[dt] = __builtin_bit_cast(k5h_f16x8, (k5h_u32x4){u0.x, u0.y, u1.x, u1.y});
        vl[dt] = __builtin_bit_cast(k5h_f16x8, (k5h_u32x4){u2.x, u2.y, u3.x, u3.y});
      }
      // O^T = V^T P^T (V as the A operand, P as B: the same register contents, swapped): the lane ends up with four consecutive
      // head-dim channels of ONE query instead of one channel of four queries -- 16-byte stores, and the pairing the split output needs.
      // Issue order alternates the two head-dim halves so no MFMA waits on the one before it; per accumulator the order of the
      // products is unchanged (vh.ph | vl.ph, vh.pl)
      Om[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh[0], pa, Om[0], 0, 0, 0);
      Om[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh[1], pa, Om[1], 0, 0, 0);
      Ol[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl[0], pa, Ol[0], 0, 0, 0);
      Ol[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl[1], pa, Ol[1], 0, 0, 0);
      Ol[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh[0], pb, Ol[0], 0, 0, 0);
      Ol[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh[1], pb, Ol[1], 0, 0, 0);
    }
    K5H_SINK(Ol[1][3]);
    K5H_STAMP(8);                                                               // P V done
    // ---- scatter: lane holds O[query = strip*16 + l15][d = 16 dt + 4 kk + r]
    const float inv = 1.0f / lsum;
    const int t = tok[strip * 16 + l15];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      f32x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = fmaf(Ol[dt][r], 0.00048828125f, Om[dt][r]) * inv;
      if (SOUT) {
        // the proj Linear's split A operand (split_linear_h3.h, "PRE"): channel c = 32 h + 16 dt + 4 kk + r -> block h, piece
        // (g = kk / 2, h|l), k-half dt; the 8-channel piece is completed by the lane of the neighbouring kk (lane ^ 16):
        // v_permlane16_swap(h, l) leaves an even-kk lane with (its h, the partner's h) and an odd-kk lane with (the partner's l, its l)
        uint32_t h0, l0, h1, l1;
        rba_split_f16x2(o.x, o.y, h0, l0);
        rba_split_f16x2(o.z, o.w, h1, l1);
        const auto p0 = __builtin_amdgcn_permlane16_swap(h0, l0, false, false), p1 = __builtin_amdgcn_permlane16_swap(h1, l1, false, false);
        const rba_u32x4 piece = {p0[0], p1[0], p0[1], p1[1]};
#ifdef K5H_ABLATE
        if (!(ablate & 4) || piece[0] == 0x12345678u)
#endif
        if (t >= 0) {
          const int64_t row = (int64_t)b * H * W + t;
          char* dst = reinterpret_cast<char*>(out) + ((row >> 5) * nH + h) * 4096 + ((kk >> 1) * 2 + (kk & 1)) * 1024 +
                      (dt * 32 + (int)(row & 31)) * 16;
          *reinterpret_cast<rba_u32x4*>(dst) = piece;
        }
      } else if (t >= 0) {
        *reinterpret_cast<f32x4*>(out + ((int64_t)b * H * W + t) * C + h * HD + 16 * dt + 4 * kk) = o;
      }
    }
  }
  K5H_STAMP(9);                                                                 // stores issued
#ifdef K5H_TIMING
  if ((threadIdx.x & 63) == 0) {
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    dbg[((((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * WAVES + (threadIdx.x >> 6)) * 12 + 10] = hw | ((unsigned long long)xcc << 32);
  }
#endif
}

}  // namespace
