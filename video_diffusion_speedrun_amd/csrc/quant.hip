// OCP fp8 quantisation for the fp8 GEMM path (BASELINE config 5; the reference has no fp8 path, so the recipe is
// this build's: per-tensor scaling, e4m3fn for activations / weights, e5m2 for gradients, saturating casts).
//
//   vds_absmax       amax = max |x| over a bf16 matrix (atomic max on the float bits; caller zeroes amax)
//   vds_quant_fp8    q = sat(x * fmax / amax) as fp8, written row-major [M,K] and, optionally, transposed [K,M]
//                    (the k-contiguous operand of the input- / weight-gradient products, which the fp8 GEMM runs
//                    as NT); dequantisation factor amax / fmax written to *dq_out.
//
// HBM-bound: 2 B/element read, 1 (+1) B/element written.  Tile 128 x 128 per 256-thread workgroup: 16-byte
// loads, 8-byte row-major stores, and the transposed copy through LDS with 4x4 byte transposes (v_perm_b32) so
// that every global store instruction still writes 128 contiguous bytes per row.
#include "common.h"
#include "prof.h"
#include "../../include/vds.h"

namespace {

constexpr int QT = 128;       // tile edge
constexpr int QLD = QT + 4;   // LDS row stride in bytes (33 dwords: 2-way conflicts at most on the column reads)

__device__ __forceinline__ void unpack8(const u32x4& u, float (&f)[8]) {
#pragma unroll
  for (int e = 0; e < 4; ++e) { f[2 * e] = bflo(u[e]); f[2 * e + 1] = bfhi(u[e]); }
}

// one 128 x 128 tile per workgroup, 8 independent 16-byte loads per thread (same indexing as quant_kernel)
__global__ __launch_bounds__(256) void absmax_kernel(const bf16_t* x, long ldx, int M, int K, float* amax) {
  const int tid = threadIdx.x;
  const int m0 = blockIdx.y * QT, k0 = blockIdx.x * QT;
  u32x4 raw[8];
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int idx = tid + 256 * it;
    const int m = m0 + (idx >> 4), k = k0 + 8 * (idx & 15);
    raw[it] = u32x4{0u, 0u, 0u, 0u};
    if (m < M && k < K) raw[it] = *reinterpret_cast<const u32x4*>(x + (long)m * ldx + k);
  }
  unsigned mx = 0u;  // max over |bf16| bit patterns (monotone for non-negative floats)
#pragma unroll
  for (int it = 0; it < 8; ++it)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      mx = max(mx, (raw[it][e] << 16) & 0x7fffffffu);
      mx = max(mx, raw[it][e] & 0x7fff0000u);
    }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
  // same-address atomics serialise (~10 ns each): one per workgroup, and only when it can raise the running max
  __shared__ unsigned red[4];
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  if (tid == 0) {
    mx = max(max(red[0], red[1]), max(red[2], red[3]));
    unsigned* a = reinterpret_cast<unsigned*>(amax);
    if (mx > __atomic_load_n(a, __ATOMIC_RELAXED)) atomicMax(a, mx);
  }
}

template <int FMT, bool TRANS>
__global__ __launch_bounds__(256) void quant_kernel(const bf16_t* x, long ldx, int M, int K, const float* amax,
                                                    unsigned char* q, long ldq, unsigned char* qt, long ldt,
                                                    float* dq_out, float* amax_out) {
  __shared__ __attribute__((aligned(16))) unsigned char tile[TRANS ? QT * QLD : 16];
  __shared__ unsigned red[4];
  unsigned mx = 0u;  // amax_out: max |x| of this tile (bf16 bit patterns), for the next step's scale
  const int tid = threadIdx.x;
  const int m0 = blockIdx.y * QT, k0 = blockIdx.x * QT;
  const float fmax = fp8_fmax(FMT);
  const float am = *amax;
  const float scale = am > 0.f ? fmax / am : 1.0f;
  if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0 && dq_out) *dq_out = am > 0.f ? am / fmax : 1.0f;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int idx = tid + 256 * it;
    const int r = idx >> 4, c = idx & 15;
    const int m = m0 + r, k = k0 + 8 * c;
    u32x2 w = {0u, 0u};
    if (m < M && k < K) {  // K is a multiple of 8: chunks are all-in or all-out
      float v[8];
      const u32x4 raw = *reinterpret_cast<const u32x4*>(x + (long)m * ldx + k);
      unpack8(raw, v);
      if (amax_out)
#pragma unroll
        for (int e = 0; e < 4; ++e) mx = max(mx, max((raw[e] << 16) & 0x7fffffffu, raw[e] & 0x7fff0000u));
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fminf(fmaxf(v[e] * scale, -fmax), fmax);
      w[0] = fp8_cvt4<FMT>(v[0], v[1], v[2], v[3]);
      w[1] = fp8_cvt4<FMT>(v[4], v[5], v[6], v[7]);
      if (q) *reinterpret_cast<u32x2*>(q + (long)m * ldq + k) = w;
    }
    if constexpr (TRANS) *reinterpret_cast<u32x2*>(tile + r * QLD + 8 * c) = w;
  }
  if (amax_out) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
    if ((tid & 63) == 0) red[tid >> 6] = mx;
  }
  if (TRANS || amax_out) __syncthreads();
  if (amax_out && tid == 0) {
    mx = max(max(red[0], red[1]), max(red[2], red[3]));
    unsigned* a = reinterpret_cast<unsigned*>(amax_out);
    if (mx > __atomic_load_n(a, __ATOMIC_RELAXED)) atomicMax(a, mx);
  }
  if constexpr (TRANS) {
    const int mq = tid & 31;  // 4 consecutive m per lane: a wave's 32 lanes cover the tile's 128 m of one k row
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int kq = (tid >> 5) + 8 * it;
      unsigned r[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) r[i] = *reinterpret_cast<const unsigned*>(tile + (4 * mq + i) * QLD + 4 * kq);
      // 4x4 byte transpose: r[i] byte j = element (m 4mq+i, k 4kq+j)  ->  c[j] byte i
      const unsigned t0 = __builtin_amdgcn_perm(r[1], r[0], 0x05010400u), t1 = __builtin_amdgcn_perm(r[1], r[0], 0x07030602u);
      const unsigned t2 = __builtin_amdgcn_perm(r[3], r[2], 0x05010400u), t3 = __builtin_amdgcn_perm(r[3], r[2], 0x07030602u);
      const unsigned c[4] = {__builtin_amdgcn_perm(t2, t0, 0x05040100u), __builtin_amdgcn_perm(t2, t0, 0x07060302u),
                             __builtin_amdgcn_perm(t3, t1, 0x05040100u), __builtin_amdgcn_perm(t3, t1, 0x07060302u)};
      const int m = m0 + 4 * mq;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = k0 + 4 * kq + j;
        if (k >= K || m >= M) continue;
        unsigned char* dst = qt + (long)k * ldt + m;
        if (m + 4 <= M) *reinterpret_cast<unsigned*>(dst) = c[j];
        else
          for (int i = 0; i < M - m; ++i) dst[i] = (unsigned char)(c[j] >> (8 * i));
      }
    }
  }
}

// qt[k, m] = q[m, k] for an fp8 (any 1-byte) matrix: the k-contiguous copy of an operand whose row-major copy was
// emitted by its producer.  Tile 128 x 128, 16-byte loads (8 per row), transposed through LDS with the 4x4 byte
// transposes of quant_kernel: every store instruction writes 128 contiguous bytes per row.  1 B read + 1 B written.
__global__ __launch_bounds__(256) void transpose_u8_kernel(const unsigned char* q, long ldq, int M, int K,
                                                           unsigned char* qt, long ldt) {
  __shared__ __attribute__((aligned(16))) unsigned char tile[QT * (QT + 16)];
  constexpr int TLD = QT + 16;  // 16-byte aligned rows (b128 stores), 36 dwords: column reads 4-way conflicted at most
  const int tid = threadIdx.x;
  const int m0 = blockIdx.y * QT, k0 = blockIdx.x * QT;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int idx = tid + 256 * it;
    const int r = idx >> 3, c = idx & 7;
    const int m = m0 + r, k = k0 + 16 * c;
    u32x4 w = {0u, 0u, 0u, 0u};
    if (m < M && k < K) w = *reinterpret_cast<const u32x4*>(q + (long)m * ldq + k);  // K % 16 == 0
    *reinterpret_cast<u32x4*>(tile + r * TLD + 16 * c) = w;
  }
  __syncthreads();
  const int mq = tid & 31;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int kq = (tid >> 5) + 8 * it;
    unsigned r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = *reinterpret_cast<const unsigned*>(tile + (4 * mq + i) * TLD + 4 * kq);
    const unsigned t0 = __builtin_amdgcn_perm(r[1], r[0], 0x05010400u), t1 = __builtin_amdgcn_perm(r[1], r[0], 0x07030602u);
    const unsigned t2 = __builtin_amdgcn_perm(r[3], r[2], 0x05010400u), t3 = __builtin_amdgcn_perm(r[3], r[2], 0x07030602u);
    const unsigned c[4] = {__builtin_amdgcn_perm(t2, t0, 0x05040100u), __builtin_amdgcn_perm(t2, t0, 0x07060302u),
                           __builtin_amdgcn_perm(t3, t1, 0x05040100u), __builtin_amdgcn_perm(t3, t1, 0x07060302u)};
    const int m = m0 + 4 * mq;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k0 + 4 * kq + j;
      if (k >= K || m >= M) continue;
      unsigned char* dst = qt + (long)k * ldt + m;
      if (m + 4 <= M) *reinterpret_cast<unsigned*>(dst) = c[j];
      else
        for (int i = 0; i < M - m; ++i) dst[i] = (unsigned char)(c[j] >> (8 * i));
    }
  }
}

}  // namespace

extern "C" int vds_transpose_fp8(const void* q, int64_t ldq, int32_t M, int32_t K, void* qt, int64_t ldt,
                                 vds_stream_t stream) {
  if (!q || !qt || M < 1 || K < 16 || (K & 15) || (ldq & 15) || (ldt & 3)) return VDS_ERR_ARG;
  const dim3 grid((K + QT - 1) / QT, (M + QT - 1) / QT);
  vdsprof::Scope ps(VDS_PROF_FP8_QUANT, (hipStream_t)stream, 0.0, 2.0 * M * K);
  hipLaunchKernelGGL(transpose_u8_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const unsigned char*)q, (long)ldq, M,
                     K, (unsigned char*)qt, (long)ldt);
  return hipGetLastError() == hipSuccess ? VDS_OK : VDS_ERR_LAUNCH;
}

extern "C" int vds_absmax(const void* x, int64_t ldx, int32_t M, int32_t K, float* amax, vds_stream_t stream) {
  if (!x || !amax || M < 1 || K < 8 || (K & 7) || (ldx & 7)) return VDS_ERR_ARG;
  const dim3 grid((K + QT - 1) / QT, (M + QT - 1) / QT);
  vdsprof::Scope ps(VDS_PROF_FP8_QUANT, (hipStream_t)stream, 0.0, 2.0 * M * K);
  hipLaunchKernelGGL(absmax_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (long)ldx, M, K, amax);
  return hipGetLastError() == hipSuccess ? VDS_OK : VDS_ERR_LAUNCH;
}

extern "C" int vds_quant_fp8(const void* x, int64_t ldx, int32_t M, int32_t K, int32_t fmt, const float* amax,
                             void* q, int64_t ldq, void* qt, int64_t ldt, float* dq_out, float* amax_out,
                             vds_stream_t stream) {
  if (!x || !amax || (!q && !qt) || M < 1 || K < 8 || (K & 7) || (ldx & 7) || (fmt != 0 && fmt != 1)) return VDS_ERR_ARG;
  if ((q && (ldq & 7)) || (qt && (ldt & 3))) return VDS_ERR_ARG;
  const dim3 grid((K + QT - 1) / QT, (M + QT - 1) / QT);
  hipStream_t s = (hipStream_t)stream;
  vdsprof::Scope ps(VDS_PROF_FP8_QUANT, s, 0.0, (2.0 + (q ? 1.0 : 0.0) + (qt ? 1.0 : 0.0)) * M * K);
#define GOQ(F, T)                                                                                                  \
  if (fmt == F && (qt != nullptr) == T) {                                                                          \
    hipLaunchKernelGGL((quant_kernel<F, T>), grid, dim3(256), 0, s, (const bf16_t*)x, (long)ldx, M, K, amax,      \
                       (unsigned char*)q, (long)ldq, (unsigned char*)qt, (long)ldt, dq_out, amax_out);             \
    return hipGetLastError() == hipSuccess ? VDS_OK : VDS_ERR_LAUNCH;                                              \
  }
  GOQ(0, false) GOQ(0, true) GOQ(1, false) GOQ(1, true)
#undef GOQ
  return VDS_ERR_UNSUPPORTED;
}
