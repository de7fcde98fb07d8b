// nn.Linear of the discriminator head (srgan/discriminator.py:65-67): batch 16 against a
// [1024][18432] fp32 weight (75.5 MB).  Every pass streams the weight exactly once, so all
// three kernels are HBM-bound; the arithmetic rides on v_mfma_f32_16x16x4_f32 with the batch
// as the 16-wide M (forward / data gradient) or as the contraction (weight gradient), which
// keeps the VALU free for address generation.  Each lane loads 16 contiguous bytes, a row's
// 64 or 256 bytes are contiguous across the lanes of one instruction.
//
// MFMA 16x16x4 maps: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D: col=l&15, row=4*(l>>4)+reg.
#include "srx_common.h"
#include <algorithm>

namespace {

// y_partial[z][b][j] = sum_{k in split z} x[b][k] w[j][k]
// block = 4 waves, one 16-wide j tile; the waves interleave 64-wide k slices.  NB blocks of 16 batch rows share every
// weight fragment (round 5: a batch of 32 -- the discriminator's real + fake pair -- streamed the 75.5 MB weight once per
// 16-row block: two launches of ~20 us where one does)
template <int NB>
__global__ __launch_bounds__(256) void linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         float* __restrict__ part, int B, int K, int J, int b0,
                                                         int kper) {
  __shared__ f32x4 red[4][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, h = lane >> 4;
  const int j0 = blockIdx.x * 16;
  const int kbeg = blockIdx.y * kper, kend = min(K, kbeg + kper);
  const int j = j0 + r;
  const bool jok = j < J;
  const float* wr = w + (size_t)(jok ? j : 0) * K;
  const float* xr[NB];
  bool bok[NB];
  f32x4 acc[NB];
#pragma unroll
  for (int n = 0; n < NB; ++n) {
    const int b = b0 + 16 * n + r;
    bok[n] = b < B;
    xr[n] = x + (size_t)(bok[n] ? b : 0) * K;
    acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  for (int kb = kbeg + wave * 64; kb < kend; kb += 256) {
    f32x4 xa[NB][4], wa[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int k = kb + 16 * t + 4 * h;
      const bool ok = k < kend;
      const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + (ok ? k : 0));
      wa[t] = (ok && jok) ? wv : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(xr[n] + (ok ? k : 0));
        xa[n][t] = (ok && bok[n]) ? xv : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[n][t][e], wa[t][e], acc[n], 0, 0, 0);
  }
#pragma unroll
  for (int n = 0; n < NB; ++n) {
    if (n) __syncthreads();
    red[wave][lane] = acc[n];
    __syncthreads();
    if (wave == 0) {
      f32x4 s = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
      // D: col = j0 + (lane&15), rows b0 + 16 n + 4*(lane>>4) + reg
      const int jj = j0 + r;
      if (jj < J) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int bb = b0 + 16 * n + 4 * h + e;
          if (bb < B) part[((size_t)blockIdx.y * B + bb) * J + jj] = s[e];
        }
      }
    }
  }
}

__global__ void linear_fwd_final_kernel(const float* __restrict__ part, int nsplit, int B, int J,
                                        const float* __restrict__ bias, int act, float slope, float* __restrict__ y) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * J) return;
  float s = 0.f;
  for (int z = 0; z < nsplit; z += 8) {  // eight partials per trip, loads first (same order of additions)
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = part[(size_t)min(z + u, nsplit - 1) * B * J + i];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (z + u < nsplit) s += v[u];
  }
  if (bias) s += bias[i % J];
  if (act == SRX_ACT_RELU) s = fmaxf(s, 0.f);
  else if (act == SRX_ACT_LRELU) s = s > 0.f ? s : s * slope;
  y[i] = s;
}

// dx_partial[z][b][k] = sum_{j in split z} dy[b][j] w[j][k];  wave = 64 k columns (4 accumulators per block of 16 rows);
// NB blocks of 16 batch rows share every weight fragment (see linear_fwd_kernel)
template <int NB>
__global__ __launch_bounds__(256) void linear_bwd_data_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                              float* __restrict__ part, int B, int K, int J, int b0,
                                                              int jper) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, h = lane >> 4;
  const int n0 = (blockIdx.x * 4 + wave) * 64;
  if (n0 >= K) return;
  const int jbeg = blockIdx.y * jper, jend = min(J, jbeg + jper);
  const int kcol = n0 + 4 * r;
  const bool kok = kcol < K;
  bool bok[NB];
  const float* dyr[NB];
  f32x4 acc[NB][4];
#pragma unroll
  for (int n = 0; n < NB; ++n) {
    const int b = b0 + 16 * n + r;
    bok[n] = b < B;
    dyr[n] = dy + (size_t)(bok[n] ? b : 0) * J;
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[n][e] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  for (int jb = jbeg; jb < jend; jb += 16) {
    float av[NB][4];
    f32x4 wv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int j = jb + 4 * t + h;
      const bool jok = j < jend;
      const f32x4 v = *reinterpret_cast<const f32x4*>(w + (size_t)(jok ? j : 0) * K + (kok ? kcol : 0));
      wv[t] = (jok && kok) ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        const float a = dyr[n][jok ? j : 0];
        av[n][t] = (jok && bok[n]) ? a : 0.f;
      }
    }
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[n][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[n][t], wv[t][e], acc[n][e], 0, 0, 0);
  }
  // acc[n][e][reg]: row b0 + 16 n + 4h + reg, column n0 + 4*r + e  -> one float4 per row
  if (kok) {
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int bb = b0 + 16 * n + 4 * h + g;
        if (bb < B) {
          f32x4 o = {acc[n][0][g], acc[n][1][g], acc[n][2][g], acc[n][3][g]};
          *reinterpret_cast<f32x4*>(part + ((size_t)blockIdx.y * B + bb) * K + kcol) = o;
        }
      }
  }
}

__global__ void sum_slabs_kernel(const float* __restrict__ part, int nsplit, int64_t n, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i * 4 >= n) return;
  f32x4 s = *reinterpret_cast<const f32x4*>(part + i * 4);
  for (int z = 1; z < nsplit; z += 4) {  // four slabs per trip, loads first (same order of additions)
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(part + (size_t)min(z + u, nsplit - 1) * n + i * 4);
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (z + u < nsplit) s += v[u];
  }
  *reinterpret_cast<f32x4*>(out + i * 4) = s;
}

// dw[j][k] = sum_b dy[b][j] x[b][k];  wave = 16 j rows x 64 k columns
__global__ __launch_bounds__(256) void linear_bwd_weight_kernel(const float* __restrict__ x,
                                                                const float* __restrict__ dy, float* __restrict__ dw,
                                                                int B, int K, int J, int accumulate) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, h = lane >> 4;
  const int n0 = (blockIdx.x * 4 + wave) * 64;
  if (n0 >= K) return;
  const int j0 = blockIdx.y * 16;
  const int kcol = n0 + 4 * r;
  const bool kok = kcol < K;
  const int j = j0 + r;
  const bool jok = j < J;
  f32x4 acc[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) acc[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int bb = 0; bb < B; bb += 4) {
    const int b = bb + h;
    const bool bok = b < B;
    const float a = dy[(size_t)(bok ? b : 0) * J + (jok ? j : 0)];
    const float av = (bok && jok) ? a : 0.f;
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)(bok ? b : 0) * K + (kok ? kcol : 0));
    const f32x4 xv = (bok && kok) ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xv[e], acc[e], 0, 0, 0);
  }
  if (kok) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int jj = j0 + 4 * h + g;
      if (jj < J) {
        f32x4 o = {acc[0][g], acc[1][g], acc[2][g], acc[3][g]};
        f32x4* dst = reinterpret_cast<f32x4*>(dw + (size_t)jj * K + kcol);
        if (accumulate) o += *dst;
        *dst = o;
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------------
// Round 6: the same three passes with NOTHING but loads and MFMAs in their loops.  On gfx950 the f32 MFMA runs on the vector ALUs
// (tools/probe/mfma_valu.hip): the kernels above select zeros for out-of-range rows / split ends with ~70 v_cndmask per 32 MFMAs and
// wait for each trip's loads before the next trip requests its own -- at batch 32 the 75.5 MB weight stream needs 60 % of the fp32
// matrix peak to keep up with HBM, and those kernels ran at 2 TB/s.  Here every operand comes through a raw buffer descriptor (rows
// past the tensor read 0: no selects; the k / j position is the instruction's scalar offset: no address arithmetic), the loads of
// trip i + 1 are requested before the MFMAs of trip i, and the work decomposition -- splits, wave interleave, order of the sums --
// is the one above, so the results are bit-identical.  Preconditions (checked on the host, otherwise the kernels above run): K a
// multiple of 64 (no partial 64-wide slice), J a multiple of 16 for the two backward passes, every tensor below 4 GiB.
// ---------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float srx_bload1(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}

template <int NB>
__global__ __launch_bounds__(256) void linear_fwd_fast_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              float* __restrict__ part, int B, int K, int J, int b0, int kper) {
  __shared__ f32x4 red[4][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = srx_uniform(tid >> 6);
  const int r = lane & 15, h = lane >> 4;
  const int j0 = blockIdx.x * 16;
  const int kbeg = blockIdx.y * kper, kend = min(K, kbeg + kper);
  const __amdgpu_buffer_rsrc_t rw = srx_rsrc(w, (unsigned)((size_t)J * K * 4)), rx = srx_rsrc(x, (unsigned)((size_t)B * K * 4));
  const unsigned wv = 4u * ((unsigned)(j0 + r) * (unsigned)K + 4u * h);  // (a row past J lies past the tensor: reads 0)
  unsigned xv[NB];
  f32x4 acc[NB];
#pragma unroll
  for (int n = 0; n < NB; ++n) {
    xv[n] = 4u * ((unsigned)(b0 + 16 * n + r) * (unsigned)K + 4u * h);
    acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  f32x4 wa[2][4], xa[2][NB][4];
  auto load = [&](int set, int kb) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const unsigned so = (unsigned)srx_uniform((kb + 16 * t) * 4);
      wa[set][t] = srx_bload(rw, wv, so);
#pragma unroll
      for (int n = 0; n < NB; ++n) xa[set][n][t] = srx_bload(rx, xv[n], so);
    }
  };
  auto mma = [&](int set) {
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[set][n][t][e], wa[set][t][e], acc[n], 0, 0, 0);
  };
  int kb = kbeg + wave * 64;  // (the waves interleave 64-wide slices, as above)
  if (kb < kend) {
    load(0, kb);
    while (true) {
      if (kb + 256 < kend) load(1, kb + 256);
      mma(0);
      kb += 256;
      if (kb >= kend) break;
      if (kb + 256 < kend) load(0, kb + 256);
      mma(1);
      kb += 256;
      if (kb >= kend) break;
    }
  }
#pragma unroll
  for (int n = 0; n < NB; ++n) {
    if (n) __syncthreads();
    red[wave][lane] = acc[n];
    __syncthreads();
    if (wave == 0) {
      f32x4 s = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
      const int jj = j0 + r;
      if (jj < J) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int bb = b0 + 16 * n + 4 * h + e;
          if (bb < B) part[((size_t)blockIdx.y * B + bb) * J + jj] = s[e];
        }
      }
    }
  }
}

template <int NB>
__global__ __launch_bounds__(256) void linear_bwd_data_fast_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                                   float* __restrict__ part, int B, int K, int J, int b0, int jper) {
  const int tid = threadIdx.x, lane = tid & 63, wave = srx_uniform(tid >> 6);
  const int r = lane & 15, h = lane >> 4;
  const int n0 = (blockIdx.x * 4 + wave) * 64;
  if (n0 >= K) return;  // (K is a multiple of 64: a wave's columns are all in range or none is)
  const int jbeg = blockIdx.y * jper, jend = min(J, jbeg + jper);
  const int kcol = n0 + 4 * r;
  const __amdgpu_buffer_rsrc_t rw = srx_rsrc(w, (unsigned)((size_t)J * K * 4)), rd = srx_rsrc(dy, (unsigned)((size_t)B * J * 4));
  const unsigned wv = 4u * ((unsigned)h * (unsigned)K + (unsigned)kcol);
  unsigned dv[NB];
  f32x4 acc[NB][4];
#pragma unroll
  for (int n = 0; n < NB; ++n) {
    dv[n] = 4u * ((unsigned)(b0 + 16 * n + r) * (unsigned)J + (unsigned)h);  // (a row past B lies past the tensor: reads 0)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[n][e] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  f32x4 wf[2][4];
  float av[2][NB][4];
  auto load = [&](int set, int jb) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      wf[set][t] = srx_bload(rw, wv, (unsigned)srx_uniform((jb + 4 * t) * K * 4));
#pragma unroll
      for (int n = 0; n < NB; ++n) av[set][n][t] = srx_bload1(rd, dv[n], (unsigned)srx_uniform((jb + 4 * t) * 4));
    }
  };
  auto mma = [&](int set) {
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[n][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[set][n][t], wf[set][t][e], acc[n][e], 0, 0, 0);
  };
  int jb = jbeg;
  if (jb < jend) {
    load(0, jb);
    while (true) {
      if (jb + 16 < jend) load(1, jb + 16);
      mma(0);
      jb += 16;
      if (jb >= jend) break;
      if (jb + 16 < jend) load(0, jb + 16);
      mma(1);
      jb += 16;
      if (jb >= jend) break;
    }
  }
#pragma unroll
  for (int n = 0; n < NB; ++n)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int bb = b0 + 16 * n + 4 * h + g;
      if (bb < B) {
        f32x4 o = {acc[n][0][g], acc[n][1][g], acc[n][2][g], acc[n][3][g]};
        *reinterpret_cast<f32x4*>(part + ((size_t)blockIdx.y * B + bb) * K + kcol) = o;
      }
    }
}

// dw[j][k] = sum_b dy[b][j] x[b][k]: the batch is the contraction, four rows per MFMA; all loads of up to 32 rows are requested
// before the first MFMA (the loop above waits per four rows)
__global__ __launch_bounds__(256) void linear_bwd_weight_fast_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                     float* __restrict__ dw, int B, int K, int J, int accumulate) {
  const int tid = threadIdx.x, lane = tid & 63, wave = srx_uniform(tid >> 6);
  const int r = lane & 15, h = lane >> 4;
  const int n0 = (blockIdx.x * 4 + wave) * 64;
  if (n0 >= K) return;
  const int j0 = blockIdx.y * 16;
  const int kcol = n0 + 4 * r;
  const __amdgpu_buffer_rsrc_t rx = srx_rsrc(x, (unsigned)((size_t)B * K * 4)), rd = srx_rsrc(dy, (unsigned)((size_t)B * J * 4));
  const unsigned xv = 4u * ((unsigned)h * (unsigned)K + (unsigned)kcol), dv = 4u * ((unsigned)h * (unsigned)J + (unsigned)(j0 + r));
  f32x4 acc[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) acc[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 prev[4];
  if (accumulate) {  // (requested with the operands: the read-modify-write does not wait behind the MFMAs)
#pragma unroll
    for (int g = 0; g < 4; ++g) prev[g] = *reinterpret_cast<const f32x4*>(dw + (size_t)(j0 + 4 * h + g) * K + kcol);
  }
  for (int bb0 = 0; bb0 < B; bb0 += 32) {
    float av[8];
    f32x4 xf[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {  // rows bb0 + 4 i + h (past B: past the tensors, zeros)
      av[i] = srx_bload1(rd, dv, (unsigned)srx_uniform((bb0 + 4 * i) * J * 4));
      xf[i] = srx_bload(rx, xv, (unsigned)srx_uniform((bb0 + 4 * i) * K * 4));
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], xf[i][e], acc[e], 0, 0, 0);
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    f32x4 o = {acc[0][g], acc[1][g], acc[2][g], acc[3][g]};
    if (accumulate) o += prev[g];
    *reinterpret_cast<f32x4*>(dw + (size_t)(j0 + 4 * h + g) * K + kcol) = o;
  }
}

// may the fast forms run?  (32-bit offsets with a row of slack for the out-of-range rows of the last tile)
bool linear_fast_ok(int B, int K, int J, bool need_j16) {
  const size_t lim = 0xfffffff0ull;
  return K % 64 == 0 && (!need_j16 || J % 16 == 0) && ((size_t)J + 16) * K * 4 < lim && ((size_t)B + 64) * K * 4 < lim && ((size_t)B + 64) * J * 4 < lim;
}

int fwd_splits(int K, int J, int B) {
  const int64_t tiles = srx_cdiv(J, 16) * srx_cdiv(B, 64);  // (a launch covers 64 rows)
  int64_t s = srx_cdiv(1024, tiles);
  const int64_t maxs = srx_cdiv(K, 256);
  if (s > maxs) s = maxs;
  if (s < 1) s = 1;
  return (int)s;
}
int bwd_splits(int K, int J, int B) {
  const int64_t blocks = srx_cdiv(K, 256) * srx_cdiv(B, 64);
  int64_t s = srx_cdiv(1024, blocks);
  const int64_t maxs = srx_cdiv(J, 64);
  if (s > maxs) s = maxs;
  if (s < 1) s = 1;
  return (int)s;
}

}  // namespace

extern "C" size_t srx_linear_ws_floats(int B, int K, int J) {
  const size_t f = (size_t)fwd_splits(K, J, B) * B * J;
  const size_t b = (size_t)bwd_splits(K, J, B) * B * K;
  return f > b ? f : b;
}

extern "C" int srx_linear_fwd(const float* x, const float* w, const float* bias, float* y, int B, int K, int J, int act,
                              float slope, float* ws, size_t ws_floats, void* stream) {
  SRX_REQUIRE(x && w && y && ws && B > 0 && K > 0 && J > 0, "linear_fwd: bad argument");
  SRX_REQUIRE(K % 4 == 0, "linear_fwd: in_features must be a multiple of 4");
  const int ns = fwd_splits(K, J, B);
  if ((size_t)ns * B * J > ws_floats) SRX_FAIL(SRX_E_WORKSPACE, "linear_fwd: workspace too small");
  const int kper = (int)srx_roundup(srx_cdiv(K, ns), 256);
  const int nsplit = (int)srx_cdiv(K, kper);
  hipStream_t st = srx_stream(stream);
  for (int b0 = 0; b0 < B; b0 += 64) {  // up to four 16-row blocks per launch share one pass over the weight
    const dim3 grid((unsigned)srx_cdiv(J, 16), nsplit);
    const int nb = (int)srx_cdiv(std::min(B - b0, 64), 16);
    if (linear_fast_ok(B, K, J, false) && nb <= 2) {
      if (nb == 1) hipLaunchKernelGGL(linear_fwd_fast_kernel<1>, grid, dim3(256), 0, st, x, w, ws, B, K, J, b0, kper);
      else hipLaunchKernelGGL(linear_fwd_fast_kernel<2>, grid, dim3(256), 0, st, x, w, ws, B, K, J, b0, kper);
    }
    else if (nb == 1) hipLaunchKernelGGL(linear_fwd_kernel<1>, grid, dim3(256), 0, st, x, w, ws, B, K, J, b0, kper);
    else if (nb == 2) hipLaunchKernelGGL(linear_fwd_kernel<2>, grid, dim3(256), 0, st, x, w, ws, B, K, J, b0, kper);
    else if (nb == 3) hipLaunchKernelGGL(linear_fwd_kernel<3>, grid, dim3(256), 0, st, x, w, ws, B, K, J, b0, kper);
    else hipLaunchKernelGGL(linear_fwd_kernel<4>, grid, dim3(256), 0, st, x, w, ws, B, K, J, b0, kper);
    SRX_CHECK_LAUNCH("linear_fwd_kernel");
  }
  hipLaunchKernelGGL(linear_fwd_final_kernel, dim3((unsigned)srx_cdiv((int64_t)B * J, 256)), dim3(256), 0, st, ws,
                     nsplit, B, J, bias, act, slope, y);
  SRX_CHECK_LAUNCH("linear_fwd_final_kernel");
  return SRX_OK;
}

extern "C" int srx_linear_bwd_data(const float* dy, const float* w, float* dx, int B, int K, int J, float* ws,
                                   size_t ws_floats, void* stream) {
  SRX_REQUIRE(dy && w && dx && ws && B > 0 && K > 0 && J > 0, "linear_bwd_data: bad argument");
  SRX_REQUIRE(K % 4 == 0, "linear_bwd_data: in_features must be a multiple of 4");
  const int ns = bwd_splits(K, J, B);
  if ((size_t)ns * B * K > ws_floats) SRX_FAIL(SRX_E_WORKSPACE, "linear_bwd_data: workspace too small");
  const int jper = (int)srx_roundup(srx_cdiv(J, ns), 16);
  const int nsplit = (int)srx_cdiv(J, jper);
  hipStream_t st = srx_stream(stream);
  for (int b0 = 0; b0 < B; b0 += 64) {  // up to four 16-row blocks per launch share one pass over the weight
    const dim3 grid((unsigned)srx_cdiv(K, 256), nsplit);
    const int nb = (int)srx_cdiv(std::min(B - b0, 64), 16);
    if (linear_fast_ok(B, K, J, true) && nb <= 2) {
      if (nb == 1) hipLaunchKernelGGL(linear_bwd_data_fast_kernel<1>, grid, dim3(256), 0, st, dy, w, ws, B, K, J, b0, jper);
      else hipLaunchKernelGGL(linear_bwd_data_fast_kernel<2>, grid, dim3(256), 0, st, dy, w, ws, B, K, J, b0, jper);
    }
    else if (nb == 1) hipLaunchKernelGGL(linear_bwd_data_kernel<1>, grid, dim3(256), 0, st, dy, w, ws, B, K, J, b0, jper);
    else if (nb == 2) hipLaunchKernelGGL(linear_bwd_data_kernel<2>, grid, dim3(256), 0, st, dy, w, ws, B, K, J, b0, jper);
    else if (nb == 3) hipLaunchKernelGGL(linear_bwd_data_kernel<3>, grid, dim3(256), 0, st, dy, w, ws, B, K, J, b0, jper);
    else hipLaunchKernelGGL(linear_bwd_data_kernel<4>, grid, dim3(256), 0, st, dy, w, ws, B, K, J, b0, jper);
    SRX_CHECK_LAUNCH("linear_bwd_data_kernel");
  }
  const int64_t n = (int64_t)B * K;
  hipLaunchKernelGGL(sum_slabs_kernel, dim3((unsigned)srx_cdiv(n / 4, 256)), dim3(256), 0, st, ws, nsplit, n, dx);
  SRX_CHECK_LAUNCH("sum_slabs_kernel");
  return SRX_OK;
}

extern "C" int srx_linear_bwd_weight(const float* x, const float* dy, float* dw, int accumulate, int B, int K, int J,
                                     void* stream) {
  SRX_REQUIRE(x && dy && dw && B > 0 && K > 0 && J > 0, "linear_bwd_weight: bad argument");
  SRX_REQUIRE(K % 4 == 0, "linear_bwd_weight: in_features must be a multiple of 4");
  if (linear_fast_ok(B, K, J, true))
    hipLaunchKernelGGL(linear_bwd_weight_fast_kernel, dim3((unsigned)srx_cdiv(K, 256), (unsigned)srx_cdiv(J, 16)), dim3(256),
                       0, srx_stream(stream), x, dy, dw, B, K, J, accumulate);
  else
    hipLaunchKernelGGL(linear_bwd_weight_kernel, dim3((unsigned)srx_cdiv(K, 256), (unsigned)srx_cdiv(J, 16)), dim3(256),
                       0, srx_stream(stream), x, dy, dw, B, K, J, accumulate);
  SRX_CHECK_LAUNCH("linear_bwd_weight_kernel");
  return SRX_OK;
}
