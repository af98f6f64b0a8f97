// Token-tile staging shared by the two qkv -> (q, k, v) producers (bf16: elementwise.hip, fp8: attention_fp8.hip).
//
// The [B*L, 3D] qkv rows of T consecutive tokens are one contiguous range: a workgroup copies it into LDS with
// LDS-DMA (16 B per lane, no VGPR round trip), rotates q and k IN PLACE there (RoPE partners are hd/2 columns apart:
// 72 bytes at head_dim 72, so a 16-byte output chunk needs operands from two different 16-byte input chunks --
// 8-byte LDS accesses, which cost nothing, instead of 8-byte global accesses, which run at 0.54-0.70 x the 16-byte
// rate), and then every lane turns one 16-byte chunk of a head row into one 16-byte store: a (tensor, head) leaves the
// workgroup as a run of T complete head rows.
#pragma once
#include "common.h"

namespace ropestage {

// quotient and remainder by a runtime divisor d <= 2^16 for u < 2^24 (exact: one float estimate + one correction)
struct Div {
  unsigned d;
  float r;
  __device__ __forceinline__ explicit Div(unsigned dd) : d(dd), r(1.0f / (float)dd) {}
  __device__ __forceinline__ unsigned div(unsigned u, unsigned& rem) const {
    unsigned q = (unsigned)((float)u * r);
    int m = (int)(u - q * d);
    if (m < 0) { --q; m += (int)d; }
    else if (m >= (int)d) { ++q; m -= (int)d; }
    rem = (unsigned)m;
    return q;
  }
};

__host__ __device__ constexpr int lds_bytes(int T, int D) { return (T * 6 * D + 1023) / 1024 * 1024; }

// phase A: issue the copy of `nt` token rows (row_b bytes each) starting at `src` into smem
__device__ __forceinline__ void issue_rows(const bf16_t* src, char* smem, int nt, int row_b, int wave, int lane) {
  const unsigned bytes = (unsigned)(nt * row_b);
  const srd_t rs = make_srd(src, bytes);
  const unsigned base = lds_addr_of(smem);
  for (unsigned off = (unsigned)wave * 1024u; off < bytes; off += 4096u) lds_dma16(rs, base + off, off + (unsigned)lane * 16u);
}

// phase B: RoPE of the q and k parts of the staged rows, in place (fp32 math, bf16 result -- model.py:266-275).
// One unit = 4 columns d..d+3 of the first half of a head and their partners d + hd/2.
template <int HD>
__device__ __forceinline__ void rotate_rows(char* smem, const float* cosb, const float* sinb, long tok0, int nt, int L,
                                            int H, int row_b, int tid) {
  constexpr int HQ = HD >> 3, HALF = HD >> 1;
  const Div by2h((unsigned)(2 * H));
  const int nunits = nt * 2 * H * HQ;
  const int l0 = (int)(tok0 % L);
  for (int u = tid; u < nunits; u += 256) {
    const int i = u % HQ;
    unsigned th;
    const unsigned t = by2h.div((unsigned)(u / HQ), th);  // th = tensor * H + head: column offset th * HD
    int l = l0 + (int)t;
    if (l >= L) l -= L;
    bf16_t* p = reinterpret_cast<bf16_t*>(smem + t * row_b) + th * HD + 4 * i;
    const f32x4 c4 = *reinterpret_cast<const f32x4*>(cosb + (long)l * HALF + 4 * i);
    const f32x4 s4 = *reinterpret_cast<const f32x4*>(sinb + (long)l * HALF + 4 * i);
    const u32x2 lo = *reinterpret_cast<const u32x2*>(p);
    const u32x2 hi = *reinterpret_cast<const u32x2*>(p + HALF);
    const float x1[4] = {bflo(lo[0]), bfhi(lo[0]), bflo(lo[1]), bfhi(lo[1])};
    const float x2[4] = {bflo(hi[0]), bfhi(hi[0]), bflo(hi[1]), bfhi(hi[1])};
    float y1[4], y2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      y1[e] = x1[e] * c4[e] + x2[e] * s4[e];
      y2[e] = x2[e] * c4[e] - x1[e] * s4[e];
    }
    *reinterpret_cast<u32x2*>(p) = u32x2{pack_bf2(y1[0], y1[1]), pack_bf2(y1[2], y1[3])};
    *reinterpret_cast<u32x2*>(p + HALF) = u32x2{pack_bf2(y2[0], y2[1]), pack_bf2(y2[2], y2[3])};
  }
}

// residual-V mix of 8 columns with the reference's rounding points (bf16 tensors: lam*v rounds, (1-lam) rounds,
// (1-lam)*v0 rounds, the sum rounds; model.py:131-134)
__device__ __forceinline__ u32x4 mix_v(u32x4 vr, u32x4 v0r, float lam, float oml) {
  u32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float a0 = bf2f(f2bf(lam * bflo(vr[e]))) + bf2f(f2bf(oml * bflo(v0r[e])));
    const float a1 = bf2f(f2bf(lam * bfhi(vr[e]))) + bf2f(f2bf(oml * bfhi(v0r[e])));
    o[e] = pack_bf2(a0, a1);
  }
  return o;
}

}  // namespace ropestage
