// K5, f16x3 form, PERSISTENT workgroups with the next (window, head)'s gather in flight (round 4, last session) -- built, bit-identical, measured,
// NOT adopted (tools only: csrc/tune/k5_persist.hip, tools/k5_persist_ab.py, profiles/r04_k5_persist.txt): isolated launches 0.798 -> 0.769 ms per image warm
// and 1.122 -> 1.118 ms cold (the prefetch does not shorten the cold launch: with the operands evicted it is the HBM fetch of the qkv tensor, not its
// latency), and in the network 140.9 / 140.0 -> 138.2 / 138.0 images/s (three streams; one 154-VGPR workgroup pinned to every CU for the whole launch keeps
// the other streams' kernels out), single stream unchanged.
// Same arithmetic, in the same order, as swin_window_attn_h3_kernel<9, 9, true, SOUT> (swin_window_attn_h3.h): bit-identical outputs.
// What changes is WHEN the global loads are issued.  The one-shot kernel's workgroup issues its K / V / Q gather, waits for it, splits it into
// the LDS planes, and only then starts the work that bounds the launch (vector / transcendental / MFMA issue: 18.4 of 25.1 us with warm
// operands) -- and in the network the operands are cold: the qkv tensor was just written by the qkv GEMM (50 MB at stage 3) and the same launch
// takes 36 us, 12 of them the gather, which a second resident workgroup does not hide (docs/kernels/K5.md).  Here one 9-wave workgroup per CU
// walks a contiguous range of (image, head, window row, window column) items: the rows of item i + 1 are requested into registers before the
// strip of item i is computed and are split into LDS after it (two barriers per item, one set of planes), the bias fragments of an item
// are requested before its LDS writes.  ~125 VGPRs: one workgroup per CU, which is all the on-chip work needs (it is issue-bound at either
// occupancy).  12 x 12 windows, head_dim 32, fragment-ordered bias only.
#pragma once

namespace {

struct K5pItem {
  const float* qkv_b;   // the item's image
  int h, wx, wy;
  int64_t row0;         // b * H * W
};

template <bool SOUT>
__global__ __launch_bounds__(576) __attribute__((amdgpu_waves_per_eu(3, 3))) void swin_window_attn_h3_persist_kernel(
    const float* __restrict__ qkv, const float* __restrict__ qkv_bias, const float* __restrict__ bias, float* __restrict__ out, int H, int W,
    int Hp, int Wp, int nH, int shift, float scale, int nwx, int nwy, int items) {
  constexpr int NT = 9, WAVES = 9, HD = 32, NP = NT * 16, PL = NP * 64, ws = 12, N = 144;
  extern __shared__ __attribute__((aligned(16))) unsigned char k5p_lds[];
  unsigned char* Kh = k5p_lds;                                                  // + PL: Kl; + 2 PL: Vh; + 3 PL: Vl
  int* tok = reinterpret_cast<int*>(k5p_lds + 4 * PL);
  int* rid = tok + NP;
  const int first = (int)((int64_t)blockIdx.x * items / gridDim.x), last = (int)((int64_t)(blockIdx.x + 1) * items / gridDim.x);
  if (first >= last) return;
  const int C = nH * HD;
  const int64_t tok_stride = 3 * (int64_t)C;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, kk = lane >> 4;
  const int kfrag = l15 * 64 + ((kk ^ ((0x1320 >> (4 * ((l15 >> 2) & 3))) & 3)) * 16);
  const int vfrag = (4 * kk + (l15 >> 2)) * 64 + (l15 & 3) * 8;
  // gather geometry of this thread, the same for every item: two 16-byte pieces (tokens t0 and t0 + 72 of the window, piece d4) of K and of V
  constexpr int NIT = 2;
  const int t0 = threadIdx.x >> 3, d4 = threadIdx.x & 7;
  const int tr0 = t0 / ws, tc0 = t0 - tr0 * ws;                                 // token t0 + 72 sits six window rows below
  const int qt = wave * 16 + l15;                                               // this lane's query of the wave's strip
  const int qr0 = qt / ws, qc0 = qt - qr0 * ws;

  auto decode = [&](int item) {
    K5pItem it;
    it.wx = item % nwx;
    const int t = item / nwx;
    it.wy = t % nwy;
    const int z = t / nwy;
    it.h = z % nH;
    const int b = z / nH;
    it.row0 = (int64_t)b * H * W;
    it.qkv_b = qkv + it.row0 * tok_stride;
    return it;
  };
  float4 kk4[NIT], vv4[NIT], q_a, q_b;
  int tkv[NIT], rgv[NIT];
  // request the K / V pieces and the Q row of `item` (registers; consumed one iteration later)
  auto request = [&](const K5pItem& it) {
    const float* qb = qkv_bias + it.h * HD;
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
      const int r = it.wy * ws + tr0 + 6 * j, c = it.wx * ws + tc0;
      int rr = r + shift, cc = c + shift;
      rr = rr >= Hp ? rr - Hp : rr;
      cc = cc >= Wp ? cc - Wp : cc;
      if (rr < H && cc < W) {
        tkv[j] = rr * W + cc;
        const float* p = it.qkv_b + (int64_t)tkv[j] * tok_stride + it.h * HD + d4 * 4;
        kk4[j] = *reinterpret_cast<const float4*>(p + C);
        vv4[j] = *reinterpret_cast<const float4*>(p + 2 * C);
      } else {
        tkv[j] = -1;
        kk4[j] = *reinterpret_cast<const float4*>(qb + C + d4 * 4);
        vv4[j] = *reinterpret_cast<const float4*>(qb + 2 * C + d4 * 4);
      }
      const int hid = r < Hp - ws ? 0 : (r < Hp - shift ? 1 : 2);
      const int wid = c < Wp - ws ? 0 : (c < Wp - shift ? 1 : 2);
      rgv[j] = hid * 3 + wid;
    }
    {
      const int r = it.wy * ws + qr0, c = it.wx * ws + qc0;
      int rr = r + shift, cc = c + shift;
      rr = rr >= Hp ? rr - Hp : rr;
      cc = cc >= Wp ? cc - Wp : cc;
      const float4* p = (rr < H && cc < W) ? reinterpret_cast<const float4*>(it.qkv_b + (int64_t)(rr * W + cc) * tok_stride + it.h * HD + kk * 8)
                                           : reinterpret_cast<const float4*>(qb + kk * 8);
      q_a = p[0];
      q_b = p[1];
    }
  };

  K5pItem cur = decode(first);
  request(cur);
  for (int item = first; item < last; ++item) {
    // ---- bias fragments of this item: the initial value of the score accumulators, requested ahead of the LDS writes and the barrier
    const float4* bfrag = reinterpret_cast<const float4*>(bias) + (((int64_t)cur.h * NT + wave) * NT) * 64 + lane;
    f32x4_t S[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c) {
      const float4 t4 = bfrag[c * 64];
      S[c] = (f32x4_t){t4.x, t4.y, t4.z, t4.w};
    }
    if (item != first) __syncthreads();                                       // every wave has finished reading the previous item's planes
    // ---- split the requested rows into the f16 planes (layout: swin_window_attn_h3.h)
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
      const int t = t0 + 72 * j;
      uint32_t h0, l0, h1, l1;
      const int tq = (t >> 2) & 3;
      const int perm = (0x1320 >> (4 * tq)) & 3;                                // P = {0, 2, 3, 1}
      k5h_split2<false>(kk4[j].x, kk4[j].y, h0, l0);
      k5h_split2<false>(kk4[j].z, kk4[j].w, h1, l1);
      const int ko = t * 64 + (((d4 >> 1) ^ perm) * 16) + (d4 & 1) * 8;
      *reinterpret_cast<uint2*>(Kh + ko) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(Kh + PL + ko) = make_uint2(l0, l1);
      k5h_split2<true>(vv4[j].x, vv4[j].y, h0, l0);
      k5h_split2<true>(vv4[j].z, vv4[j].w, h1, l1);
      const int vo = t * 64 + (((d4 >> 2) ^ (tq & 1)) * 32) + (d4 & 3) * 8;
      *reinterpret_cast<uint2*>(Kh + 2 * PL + vo) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(Kh + 3 * PL + vo) = make_uint2(l0, l1);
      if (d4 == 0) { tok[t] = tkv[j]; rid[t] = rgv[j]; }
    }
    k5h_f16x8 qh, ql;
    {
      k5h_u32x4 hq, lq;
      uint32_t x0, x1;
      k5h_split2<false>(q_a.x * scale, q_a.y * scale, x0, x1); hq[0] = x0; lq[0] = x1;
      k5h_split2<false>(q_a.z * scale, q_a.w * scale, x0, x1); hq[1] = x0; lq[1] = x1;
      k5h_split2<false>(q_b.x * scale, q_b.y * scale, x0, x1); hq[2] = x0; lq[2] = x1;
      k5h_split2<false>(q_b.z * scale, q_b.w * scale, x0, x1); hq[3] = x0; lq[3] = x1;
      qh = __builtin_bit_cast(k5h_f16x8, hq);
      ql = __builtin_bit_cast(k5h_f16x8, lq);
    }
    const bool need_mask = shift > 0 && (cur.wy == nwy - 1 || cur.wx == nwx - 1);
    const int h = cur.h;
    const int64_t row0 = cur.row0;
    __syncthreads();
    // ---- the next item's rows go on their way now; they are consumed after this item's strip
    if (item + 1 < last) {
      cur = decode(item + 1);
      request(cur);
    }
    // ---- S^T = K . Q^T (three sweeps: kh.qh, kh.ql, kl.qh per accumulator, in this order)
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
    for (int c = 0; c < NT; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) m = fmaxf(m, S[c][r]);
    m = fmaxf(m, __shfl_xor(m, 16, RBA_WAVE));
    m = fmaxf(m, __shfl_xor(m, 32, RBA_WAVE));
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
    // ---- O = P . V
    f32x4_t Om[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, Ol[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int c0 = 0; c0 < NT; c0 += 2) {
      const bool two = c0 + 1 < NT;
      const int c1 = two ? c0 + 1 : c0;
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
      Om[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh[0], pa, Om[0], 0, 0, 0);
      Om[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh[1], pa, Om[1], 0, 0, 0);
      Ol[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl[0], pa, Ol[0], 0, 0, 0);
      Ol[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl[1], pa, Ol[1], 0, 0, 0);
      Ol[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh[0], pb, Ol[0], 0, 0, 0);
      Ol[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh[1], pb, Ol[1], 0, 0, 0);
    }
    // ---- scatter: lane holds O[query = qt][d = 16 dt + 4 kk + r]
    const float inv = 1.0f / lsum;
    const int t = tok[qt];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      f32x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = fmaf(Ol[dt][r], 0.00048828125f, Om[dt][r]) * inv;
      if (SOUT) {
        uint32_t h0, l0, h1, l1;
        rba_split_f16x2(o.x, o.y, h0, l0);
        rba_split_f16x2(o.z, o.w, h1, l1);
        const auto p0 = __builtin_amdgcn_permlane16_swap(h0, l0, false, false), p1 = __builtin_amdgcn_permlane16_swap(h1, l1, false, false);
        const rba_u32x4 piece = {p0[0], p1[0], p0[1], p1[1]};
        if (t >= 0) {
          const int64_t row = row0 + t;
          char* dst = reinterpret_cast<char*>(out) + ((row >> 5) * nH + h) * 4096 + ((kk >> 1) * 2 + (kk & 1)) * 1024 +
                      (dt * 32 + (int)(row & 31)) * 16;
          *reinterpret_cast<rba_u32x4*>(dst) = piece;
        }
      } else if (t >= 0) {
        *reinterpret_cast<f32x4*>(out + (row0 + t) * C + h * HD + 16 * dt + 4 * kk) = o;
      }
    }
  }
}

// one workgroup per CU (the kernel's register budget allows no more), fewer when there are fewer items
template <bool SOUT>
int launch_h3_persist(const float* qkv, const float* qkv_bias, const float* bias_frag, float* out, int B, int H, int W, int Hp, int Wp, int nH,
                      int shift, float scale, hipStream_t st) {
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    n_cu = n;
  }
  const int nwx = Wp / 12, nwy = Hp / 12;
  const int64_t items64 = (int64_t)nwx * nwy * B * nH;
  if (items64 > 0x7fffffff) return (int)hipErrorInvalidValue;
  const int items = (int)items64;
  const size_t shm = (size_t)(4 * 9 * 16 * 64) + (size_t)(2 * 9 * 16) * sizeof(int);
  const dim3 grid((unsigned)(items < n_cu ? items : n_cu)), block(576);
  hipLaunchKernelGGL((swin_window_attn_h3_persist_kernel<SOUT>), grid, block, shm, st, qkv, qkv_bias, bias_frag, out, H, W, Hp, Wp, nH, shift, scale,
                     nwx, nwy, items);
  return rba_launch_status();
}

}  // namespace
