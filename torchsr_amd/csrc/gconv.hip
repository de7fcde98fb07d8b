// Convolution as a gather-GEMM on the fp32 matrix cores of gfx950 (MI355X).
//
//   out[opix(m)][n] = epilogue( sum_k A(m,k) * B[n][k] ),   A(m,k) = in[ipix(m)+tap(k)][chan(k)]  (0 outside)
//
// One kernel serves nn.Conv2d forward (any kernel size / stride / padding), the
// data gradient (run per stride-parity class with flipped taps, so stride-2
// layers do no wasted work) and -- through in_shuffle / out_shuffle addressing --
// the PixelShuffle(2) of the SRGAN sub-pixel layer (srgan/residual.py:27-28).
// A second kernel computes the weight gradient dW[n][k] = sum_m dy[m][n] A(m,k).
//
// Data layout: activations NHWC fp32 (channel stride multiple of 4), packed
// weights [N_pad][K_pad] with K = (tap, channel) contiguous, so both MFMA
// operands are "k-contiguous rows" and are staged into XOR-swizzled LDS rows of
// 32 floats read back with conflict-free ds_read_b128.
// MFMA: v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD = 157.3 TFLOP/s chip peak), plus
// v_mfma_f32_16x16x4_f32 for the 16 extra rows of the 144-row tiles and v_mfma_f32_32x32x16_bf16 for
// the bf16-product mode (srx_conv2d_t::precision).
#include "srx_common.h"
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <mutex>
#include <type_traits>
#include <vector>

namespace {

struct GArgs {
  const float* in; const float* w; const float* bias; float* out; float* part;
  int N, Hi, Wi, Ci;
  int Hm, Wm, M, HmWm;
  float inv_HmWm, inv_Wm;
  int in_stride, nth, ntw, dh0, dw0, Ck, K, Kp;
  int Cn, Cs;
  int Ho, Wo, Co;
  int out_stride, oh_off, ow_off;
  int out_shuffle, in_shuffle;
  // up != 0: the conv reads a nearest-neighbour x2 upsampling of `in` that is never materialised (esrgan/generator.py:73,76:
  // F.interpolate(scale_factor=2) feeding a conv).  Hi, Wi are the UPSAMPLED extents (bounds of the taps); the pixel
  // (ih, iw) is fetched from (ih >> 1, iw >> 1) of the stored [N][Hi/2][Wi/2][Ci] tensor.
  int up;
  int act; float slope;
  int linear_out;
  int mtiles;
  // work decomposition: workgroups [0, full_tiles) each own one whole output tile; the remaining
  // (tail) tiles are cut tail_split ways along K so that the last, partially filled "round" of the
  // chip still keeps every CU busy; their raw partial tiles go to `ws` and tail_fixup_kernel
  // finishes them (sum, bias, activation, BatchNorm partials).
  int kchunks, kc_per_split, full_tiles, tail_split;
  float* ws;
  // sizes of the `in` and `w` buffers: both are read through raw buffer descriptors, whose range
  // check returns 0 for the padding taps (no branch, no select, statically countable loads).  The input descriptor is based
  // at the first byte the TILE can touch (round 4: a 64-bit base per workgroup, 32-bit offsets inside it), so a call is
  // not capped at 4 GiB of input; `big`: M >= 2^24, pixel indices are split by exact integer division instead of the float one
  size_t in_bytes;
  unsigned w_bytes;
  int big;
  unsigned in_margin;  // bytes the most negative tap reaches in front of a row's centre pixel
  // epilogue addend, laid out like `out` (null: none): `out` itself when a data gradient is summed into a
  // shared dense-block buffer, or the skip input of a residual block in eval mode (BatchNorm folded into the
  // conv, `x + conv(...)` in one kernel).  Linear outputs only.
  const float* add;
  float oscale;  // applied to act(acc + bias) before the addend (conv5 * 0.2 + x of the dense block); 1 otherwise
  // The addend may have a row stride of its own (add_ld floats; linear outputs only -- otherwise it equals Co), may cover
  // only the leading columns [0, add_hi), and is scaled by ascale: out = act(acc + bias) * oscale + add * ascale.  ESRGAN's
  // trunk keeps every dense block's input in the first 64 of a 192-channel buffer (stride 192) while block gradients
  // are dense 64-channel tensors, and the gradient entering a block is added to the block's input gradient there.
  int add_ld, add_hi;
  float ascale;
  // mask (null: none): the OUTPUT of the activation that produced this conv's input, laid out like `out`.  The data
  // gradient is multiplied by that activation's derivative on its way out, v * (mask[i] > 0 ? 1 : mask_slope), for
  // the output columns [mask_lo, mask_hi) -- the ReLU / LeakyReLU backward of the layer below without a pass of its
  // own.  Applied after the addend: in ESRGAN's dense block the convs' input gradients accumulate in one shared
  // buffer and the conv that completes a 32-channel slice also applies that slice's LeakyReLU mask.
  const float* mask;
  float mask_slope;
  int mask_lo, mask_hi;
  // PR = 2 (bf16 STORAGE, round 5): `in` and `w` hold bf16 values -- described to the kernel as fp32 tensors of half the
  // channel count, so every gather offset below is the fp32 code's -- and `out` / `mask` are bf16 tensors when these are set
  int out16, mask16;
};


constexpr int BK = 32;           // floats per k-chunk (one 128-byte LDS row)
constexpr int INVALID = -20000;  // coordinate that fails every bounds check

// ---------------------------------------------------------------------------
// forward / data-gradient gather-GEMM.  256 threads = 4 waves, wave tile WMxWN.
// ---------------------------------------------------------------------------
// KS > 1: the k-chunks of the tile are dealt round-robin to KS groups of waves (each group owns a
// private pair of LDS staging buffers and covers the whole BM x BN tile); the groups' accumulators are
// folded through LDS at the end.  Used when a launch has fewer tiles than CUs (e.g. the 16x24x24
// residual convs: 144 tiles): it halves the serial chunk chain per wave and puts two waves on a SIMD.
//
// XR = 16: the tile carries 16 more rows (BM + 16 = 144), multiplied on v_mfma_f32_16x16x4_f32 as BN/16
// blocks of 16x16 by the first BN/16 waves.  The SRGAN shapes have M = 2304 * 4^j pixels: with 144-row
// tiles every layer cuts into a power-of-two number of tiles, i.e. whole rounds of the 256 CUs, where
// 128-row tiles leave 12-25 % of a round idle and need the K-split fix-up pass.
//
// PR = 1: bf16 products, fp32 accumulation (srx_conv2d_t::precision).  Global memory stays fp32; a chunk's
// 32 k-values are rounded to bf16 when they are written to LDS (64-byte rows) and multiplied by two
// v_mfma_f32_32x32x16_bf16 per 32x32 block instead of sixteen fp32 MFMAs -- the loop is then bound by the
// L2 -> LDS stream, not by the matrix pipe.
// BIG = 1 (round 4): calls above 2^24 pixels / 4 GiB of input (whole-frame inference).  The input descriptor is based at the
// first byte the TILE can touch -- a 64-bit base per workgroup, 32-bit offsets inside it -- and pixel indices are split by
// exact integer division.  A separate instantiation: with the per-tile base in the common kernel the SRGAN step lost 1 %
// (same-box A/B, 8.87 vs 8.97 ms), so BIG = 0 is the code of round 3 to the letter.
template <int BM, int BN, int WM, int WN, int KS, int XR, int PR, int UP = 0, int BIG = 0>
__device__ __forceinline__ void gconv_body(const GArgs& a, const int bid) {
  static_assert(XR == 0 || (XR == 16 && KS == 1), "extra rows: 16, without the in-workgroup K split");
  static_assert(PR == 0 || XR == 0, "the 16-row extension is fp32 only");
  constexpr int BKL = PR == 1 ? BK / 2 : BK;  // floats per LDS row
  constexpr int NS = PR == 1 ? 2 : 4;         // MFMA sub-steps per chunk
  constexpr int BMT = BM + XR;  // rows of the tile
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int WAVES_N = BN / WN;
  constexpr int WAVES_M = BM / WM;
  constexpr int GT = WAVES_M * WAVES_N * 64;  // threads of one k-group
  constexpr int NT = GT * KS;                 // 256 or 512 threads: the big tiles run 8 waves so that every
                                              // SIMD holds two and one's MFMAs cover the other's loads/barriers
  static_assert(NT == 256 || NT == 512, "4 or 8 waves per workgroup");
  constexpr int RPP = GT / 8;                 // tile rows staged per pass (8 threads x float4 = one 128-byte row)
  constexpr int RA = (BMT + RPP - 1) / RPP, RB = BN / RPP;  // the last A pass may be partly past the tile
  static_assert(RA >= 1 && RB >= 1, "tile too small for the thread count");
  static_assert(XR == 0 || BN / 16 <= GT / 64, "one extra 16x16 block per wave at most");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  // wave-uniform quantities are forced into SGPRs (readfirstlane): loop control then compiles to
  // scalar branches instead of exec-mask juggling around the MFMA blocks
  const int ks = KS == 1 ? 0 : srx_uniform(threadIdx.x / GT);  // k-group of this wave (groups are contiguous)
  const int tid = threadIdx.x - ks * GT, lane = tid & 63, wave = srx_uniform(tid >> 6);
  constexpr int RING = 3;  // LDS chunk buffers per k-group
  float* sA = reinterpret_cast<float*>(smem) + ks * RING * (BMT + BN) * BKL;
  float* sB = sA + RING * BMT * BKL;
  int2* ktab = reinterpret_cast<int2*>(reinterpret_cast<float*>(smem) + KS * RING * (BMT + BN) * BKL);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  int tile = bid, kc_beg = 0, kc_end = a.kchunks;
  bool raw = false;
  float* slab = nullptr;
  if (bid >= a.full_tiles) {
    const int t = bid - a.full_tiles;
    tile = a.full_tiles + t / a.tail_split;
    if (a.tail_split > 1) {
      kc_beg = (t % a.tail_split) * a.kc_per_split;
      kc_end = min(a.kchunks, kc_beg + a.kc_per_split);
      raw = true;
      slab = a.ws + (size_t)t * (BMT * BN);
    }
  }
  tile = srx_uniform(tile); kc_beg = srx_uniform(kc_beg); kc_end = srx_uniform(kc_end);
  const int nt = srx_uniform(tile / a.mtiles), mt = tile - nt * a.mtiles;
  const int m0 = mt * BMT, n0 = nt * BN;
  const int q = tid & 7, r0 = tid >> 3;
  auto split_row = [&](int m, int& n, int& mh, int& mw) {  // m -> (image, row, column) of the M grid
    if constexpr (BIG) {  // exact: the float reciprocal trick holds below 2^24 only
      n = m / a.HmWm;
      const int rem = m - n * a.HmWm;
      mh = rem / a.Wm;
      mw = rem - mh * a.Wm;
    } else {
      int rem;
      srx_divmod(m, a.HmWm, a.inv_HmWm, n, rem);
      srx_divmod(rem, a.Wm, a.inv_Wm, mh, mw);
    }
  };
  // byte offset of row m's centre pixel in the input tensor (monotonic in m)
  auto centre = [&](int n, int ih0, int iw0) -> size_t {
    return 4 * (UP ? (size_t)n * (a.Hi >> 1) * (a.Wi >> 1) * a.Ci
                   : a.in_shuffle ? (((size_t)n * 2 * a.Hi + 2 * ih0) * (2 * a.Wi) + 2 * iw0) * a.Ci
                                  : (((size_t)n * a.Hi + ih0) * a.Wi + iw0) * a.Ci);
  };
  size_t tbase = 0;  // BIG: first byte this tile can touch: its first row's centre less the reach of the most negative tap
  if constexpr (BIG) {
    int n, mh, mw;
    split_row(min(m0, a.M - 1), n, mh, mw);
    const size_t c0 = centre(n, mh * a.in_stride, mw * a.in_stride);
    tbase = c0 > a.in_margin ? c0 - a.in_margin : 0;
    const unsigned tb_in_lo = (unsigned)srx_uniform((int)(unsigned)(tbase & 0xffffffffu));
    const unsigned tb_in_hi = (unsigned)srx_uniform((int)(unsigned)(tbase >> 32));
    tbase = ((size_t)tb_in_hi << 32) | tb_in_lo;
  }
  const size_t in_left = a.in_bytes - tbase;
  const __amdgpu_buffer_rsrc_t rin = BIG ? srx_rsrc(reinterpret_cast<const char*>(a.in) + tbase, in_left > 0xfffffff0ull ? 0xfffffff0u : (unsigned)in_left)
                                         : srx_rsrc(a.in, (unsigned)a.in_bytes);
  const __amdgpu_buffer_rsrc_t rw = srx_rsrc(a.w, a.w_bytes);

  // ---- k table: (dh, dw) and linear input offset for every float4 of K in range
  for (int e = kc_beg * 8 + (int)threadIdx.x; e < kc_end * 8; e += NT) {
    const int k = 4 * e;
    int2 ent;
    if (k < a.K) {
      const int tap = k / a.Ck, c = k - tap * a.Ck;
      const int th = tap / a.ntw, tw = tap - th * a.ntw;
      const int dh = a.dh0 + th, dw = a.dw0 + tw;
      int koff;
      if (UP) {
        koff = c;  // the pixel part of the offset depends on the row's parity: added per load (gload)
      } else if (a.in_shuffle) {
        const int ij = c / a.in_shuffle, cc = c - ij * a.in_shuffle;
        koff = ((2 * dh + (ij >> 1)) * (2 * a.Wi) + 2 * dw + (ij & 1)) * a.Ci + cc;
      } else {
        koff = (dh * a.Wi + dw) * a.Ci + c;
      }
      ent.x = (int)((unsigned)(dh & 0xffff) | ((unsigned)dw << 16));
      ent.y = koff * 4;  // bytes
    } else {
      ent.x = (INVALID & 0xffff);
      ent.y = 0;
    }
    ktab[e - kc_beg * 8] = ent;
  }

  // ---- per-thread rows of the A tile (fixed for the whole k loop)
  int rih[RA], riw[RA];
  unsigned rbase[RA];  // byte offset of the row's centre pixel
#pragma unroll
  for (int p = 0; p < RA; ++p) {
    const int m = m0 + r0 + RPP * p;
    if (m < a.M && r0 + RPP * p < BMT) {
      int n, mh, mw;
      split_row(m, n, mh, mw);
      const int ih0 = mh * a.in_stride, iw0 = mw * a.in_stride;
      rih[p] = ih0;
      riw[p] = iw0;
      if constexpr (BIG) {
        rbase[p] = (unsigned)(centre(n, ih0, iw0) - tbase);  // (relative to the tile's base: a tile spans a few image rows)
      } else {
        rbase[p] = 4u * (unsigned)(UP ? n * (a.Hi >> 1) * (a.Wi >> 1) * a.Ci
                                   : a.in_shuffle ? ((n * 2 * a.Hi + 2 * ih0) * (2 * a.Wi) + 2 * iw0) * a.Ci
                                                  : ((n * a.Hi + ih0) * a.Wi + iw0) * a.Ci);
      }
    } else {
      rih[p] = INVALID; riw[p] = 0; rbase[p] = 0;
    }
  }
  unsigned wvoff[RB];
#pragma unroll
  for (int p = 0; p < RB; ++p) wvoff[p] = 4u * ((unsigned)(n0 + r0 + RPP * p) * (unsigned)a.Kp + q * 4);
  __syncthreads();

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Three register stages: while chunk k is multiplied out of LDS, chunk k+1 is being written to the
  // ring and the loads of chunks k+2 and k+3 are in flight (a load has two full steps to arrive).  With
  // one workgroup per CU a single stage leaves ~0.5 us of L2/HBM latency exposed per chunk.
  f32x4 ra0[RA], rb0[RB], ra1[RA], rb1[RB], ra2[RA], rb2[RB];
  // Every step issues the same RA + RB loads -- past the end of the k range they are pointed out of
  // range and cost nothing -- so that the compiler can count outstanding loads exactly (s_waitcnt
  // vmcnt(N) for the older register stage only) instead of draining both stages at every step.
  auto gload = [&](int kc, f32x4 (&ra)[RA], f32x4 (&rb)[RB]) {
    const bool live = kc < kc_end;  // wave-uniform
    const int2 kt = ktab[(live ? kc - kc_beg : 0) * 8 + q];
    const int dh = (int)(short)(kt.x & 0xffff), dw = kt.x >> 16;
    if (UP) {  // compile time: a run-time test here, even a scalar one, cost the fp32 step 1.2 % (same-box A/B)
#pragma unroll
      for (int p = 0; p < RA; ++p) {
        const int ih = rih[p] + dh, iw = riw[p] + dw;
        const bool ok = live && ((unsigned)ih < (unsigned)a.Hi) && ((unsigned)iw < (unsigned)a.Wi);
        const unsigned pix = 4u * (unsigned)(((ih >> 1) * (a.Wi >> 1) + (iw >> 1)) * a.Ci);  // source pixel of the upsampled one
        ra[p] = srx_bload(rin, ok ? rbase[p] + pix + (unsigned)kt.y : 0xffffffffu, 0);
      }
    } else {
#pragma unroll
      for (int p = 0; p < RA; ++p) {
        const int ih = rih[p] + dh, iw = riw[p] + dw;
        const bool ok = live && ((unsigned)ih < (unsigned)a.Hi) && ((unsigned)iw < (unsigned)a.Wi);
        ra[p] = srx_bload(rin, ok ? rbase[p] + (unsigned)kt.y : 0xffffffffu, 0);  // out of range reads 0
      }
    }
#pragma unroll
    for (int p = 0; p < RB; ++p)
      rb[p] = srx_bload(rw, live ? wvoff[p] : 0xffffffffu, (unsigned)srx_uniform(live ? kc * (BK * 4) : 0));
  };
  // fp32: 16-byte quad q of the row at quad q ^ ((row>>1)&7).  bf16: the quad shrinks to 8 bytes; quads 2g, 2g+1
  // form the 16-byte group g (k = 8g..8g+7, one MFMA operand), stored at group g ^ ((row>>2)&3).
  const int wchunk = PR == 1 ? (((q >> 1) ^ ((r0 >> 2) & 3)) * 4 + (q & 1) * 2) : (q ^ ((r0 >> 1) & 7)) * 4;
  auto put = [&](float* dst, const f32x4& v) {
    if (PR == 1) {
      typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
      const bf16x2 lo = {(__bf16)v[0], (__bf16)v[1]}, hi = {(__bf16)v[2], (__bf16)v[3]};
      *reinterpret_cast<uint2*>(dst) = make_uint2(__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi));
    } else {
      *reinterpret_cast<f32x4*>(dst) = v;
    }
  };
  auto swrite = [&](int buf, const f32x4 (&ra)[RA], const f32x4 (&rb)[RB]) {
    float* dA = sA + buf * BMT * BKL;
    float* dB = sB + buf * BN * BKL;
#pragma unroll
    for (int p = 0; p < RA; ++p)
      if (RPP * (p + 1) <= BMT || r0 + RPP * p < BMT)  // (compile-time true except in a partial last pass)
        put(dA + (r0 + RPP * p) * BKL + wchunk, ra[p]);
#pragma unroll
    for (int p = 0; p < RB; ++p) put(dB + (r0 + RPP * p) * BKL + wchunk, rb[p]);
  };

  const int h = lane >> 5, l31 = lane & 31;
  const int xr = PR == 1 ? (l31 >> 2) & 3 : (l31 >> 1) & 7;
  const int arow = (wm * WM + l31) * BKL, brow = (wn * WN + l31) * BKL;
  // extra 16 rows: wave w < BN/16 owns the 16x16 block of columns 16w..16w+15
  //   v_mfma_f32_16x16x4_f32: A[i = l&15][k = l>>4], B[k = l>>4][j = l&15], D: col = l&15, row = 4(l>>4) + reg
  const bool has_x = XR > 0 && wave < BN / 16;
  const int xi = lane & 15, xg = lane >> 4;
  const int xra = BM + xi, xrb = 16 * wave + xi;  // LDS rows of this lane's A / B fragments
  f32x4 accx = {0.f, 0.f, 0.f, 0.f};
  // ---- the pipelined k loop -----------------------------------------------------------------
  // LDS holds a ring of three chunk buffers.  In step k a wave
  //   1. writes chunk k+1 (in registers since the previous step) into ring slot (k+1) % 3,
  //   2. requests chunk k+3 from global memory into the register stage that just became free (three
  //      stages: a load has two full steps to arrive before it is written to LDS),
  //   3. multiplies chunk k out of slot k % 3, with the step's ONE barrier in the middle.
  // Chunk k+1 is therefore complete in LDS half a step before anyone needs it: the barrier has slack
  // for skew between waves instead of standing between the last MFMA of a chunk and the first LDS
  // read of the next, and that first read (fragment 0 of chunk k+1) is issued under the last MFMAs of
  // chunk k.  Slot (k+1) % 3 was last read in step k-2; the barrier of step k-1 separates the two.
  // (With two slots the write had to wait for the end of the step and every chunk boundary cost the
  // matrix pipe a barrier + a write + a read latency: 28 % idle with one workgroup per CU.)
  f32x4 af[2][TM], bf[2][TN];  // fragment register sets: step s+1 is read while step s is multiplied
  auto frag = [&](int slot, int s, int set) {
    const float* cA = sA + slot * BMT * BKL + arow;
    const float* cB = sB + slot * BN * BKL + brow;
    const int ch = ((2 * s + h) ^ xr) * 4;  // fp32: quad of 4 k;  bf16: group of 8 k (one 32x32x16 operand)
#pragma unroll
    for (int i = 0; i < TM; ++i) af[set][i] = *reinterpret_cast<const f32x4*>(cA + i * 32 * BKL + ch);
#pragma unroll
    for (int j = 0; j < TN; ++j) bf[set][j] = *reinterpret_cast<const f32x4*>(cB + j * 32 * BKL + ch);
  };
  auto mma = [&](int set) {
    __builtin_amdgcn_s_setprio(1);
    if (PR) {  // (PR = 2: a 16-byte quad of the stored tensor IS eight bf16 k-values, one MFMA operand)
      typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[set][i]),
                                                              __builtin_bit_cast(bf16x8, bf[set][j]), acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      return;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][i][e], bf[set][j][e], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
  auto extra = [&](int slot) {  // each b128 holds k = 4Q..4Q+3 of one row; MFMA e of read u contracts k = {4(g + 4u) + e}
    const float* xA = sA + slot * BMT * BKL + xra * BKL;
    const float* xB = sB + slot * BN * BKL + xrb * BKL;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const f32x4 fa = *reinterpret_cast<const f32x4*>(xA + (((xg + 4 * u) ^ ((xra >> 1) & 7)) * 4));
      const f32x4 fb = *reinterpret_cast<const f32x4*>(xB + (((xg + 4 * u) ^ ((xrb >> 1) & 7)) * 4));
#pragma unroll
      for (int e = 0; e < 4; ++e) accx = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[e], fb[e], accx, 0, 0, 0);
    }
  };
  // one step; `slot` = ring slot of chunk kc, fragment 0 of which already sits in register set 0
  auto step = [&](int kc, int slot, f32x4 (&ra_next)[RA], f32x4 (&rb_next)[RB], f32x4 (&ra_free)[RA], f32x4 (&rb_free)[RB]) {
    const int nslot = slot == 2 ? 0 : slot + 1;
    swrite(nslot, ra_next, rb_next);          // chunk kc + KS (zeros past the end of the range: never multiplied)
    gload(kc + 3 * KS, ra_free, rb_free);
    const bool live = kc < kc_end;            // wave-uniform
    if (NS == 2) {  // bf16: two MFMA sub-steps per chunk
      frag(slot, 1, 1);
      if (live) mma(0);
      __syncthreads();
      frag(nslot, 0, 0);
      if (live) mma(1);
      return;
    }
    frag(slot, 1, 1);
    if (live) mma(0);
    frag(slot, 2, 0);
    if (live) mma(1);
    __syncthreads();
    frag(slot, 3, 1);
    if (live) mma(0);
    frag(nslot, 0, 0);
    if (XR > 0 && has_x && live) extra(slot);
    if (live) mma(1);
  };

  // this group's chunks: kc_beg + ks, + KS, ...; every group runs the same number of steps (barriers)
  const int c0 = kc_beg + ks;
  const int nsteps = (kc_end - kc_beg + KS - 1) / KS;
  gload(c0, ra0, rb0);
  gload(c0 + KS, ra1, rb1);
  gload(c0 + 2 * KS, ra2, rb2);
  swrite(0, ra0, rb0);
  __syncthreads();
  frag(0, 0, 0);
  // three steps per trip (ring slots and register stages rotate statically); all three run even past
  // the end of the range: an early exit would make the loop's load count path-dependent and cost the
  // exact vmcnt waits
  for (int j = 0; j < nsteps; j += 3) {
    const int kc = c0 + j * KS;
    step(kc, 0, ra1, rb1, ra0, rb0);
    step(kc + KS, 1, ra2, rb2, ra1, rb1);
    step(kc + 2 * KS, 2, ra0, rb0, ra2, rb2);
  }
  __syncthreads();  // the ring is reused below (fold / statistics)

  if (KS > 1) {  // fold the k-groups' accumulators into group 0 (staging buffers are free now)
    float* fold = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int g = 1; g < KS; ++g) {
      if (ks == g) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int jj = 0; jj < TN; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) fold[((wave * TM * TN + i * TN + jj) * 16 + r) * 64 + lane] = acc[i][jj][r];
      }
      __syncthreads();
      if (ks == 0) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int jj = 0; jj < TN; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jj][r] += fold[((wave * TM * TN + i * TN + jj) * 16 + r) * 64 + lane];
      }
      __syncthreads();
    }
  }
  const bool lead = (ks == 0);

  // ------------------------------------------------------------- epilogue
  // accumulator map (32x32 MFMA): col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  if (raw) {  // tile-local [BM][BN] partial; each half wave writes 128 contiguous bytes
    if (!lead) return;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          slab[row * BN + wn * WN + j * 32 + l31] = acc[i][j][r];
        }
    if (XR > 0 && has_x) {
#pragma unroll
      for (int r = 0; r < 4; ++r) slab[(BM + 4 * xg + r) * BN + 16 * wave + xi] = accx[r];
    }
    return;
  }
  // Straight-line epilogue: the activation is a select on a host-prepared slope (none 1, ReLU 0), rows
  // past M and padding columns are stored through a descriptor at an out-of-range offset (dropped by
  // the hardware) instead of being branched around -- per-element branches used to cost more than
  // the arithmetic.
  auto out_elem = [&](int m) -> size_t {  // element offset of output row m
    if (a.linear_out) return (size_t)m * a.Co;
    int n, mh, mw;
    split_row(m, n, mh, mw);
    const int oh = mh * a.out_stride + a.oh_off, ow = mw * a.out_stride + a.ow_off;
    if constexpr (BIG) return (((size_t)n * a.Ho + oh) * a.Wo + ow) * a.Co;
    else return ((size_t)(n * a.Ho + oh) * a.Wo + ow) * a.Co;
  };
  // descriptor based at the tile's first row: offsets inside a tile are small and never negative
  // (output rows are laid out in increasing m), so 32 bits are enough whatever the tensor size
  const size_t tile_base = out_elem(m0);
  const unsigned tb_lo = (unsigned)srx_uniform((int)(unsigned)(tile_base & 0xffffffffu));
  const unsigned tb_hi = (unsigned)srx_uniform((int)(unsigned)(tile_base >> 32));
  // (PR = 2: bf16 outputs / masks -- offsets below stay 4 x the element index and are halved at the access)
  // (a bf16-storage launch's mask is always bf16; its output type picks one of two copies of the epilogue below)
  const bool out16 = PR == 2 && a.out16;
  constexpr bool mask16 = PR == 2;
  const __amdgpu_buffer_rsrc_t rout = srx_rsrc(reinterpret_cast<char*>(a.out) + ((((size_t)tb_hi << 32) | tb_lo) << (out16 ? 1 : 2)), 0xfffffff0u);
  // (the addend's tile base differs from the output's when it has its own row stride -- linear outputs only; built
  // inside the add-mode paths so that the other epilogues do not carry its scalars)
  auto make_radd = [&]() {
    const size_t add_base = a.linear_out ? (size_t)m0 * a.add_ld : tile_base;
    const unsigned ab_lo = (unsigned)srx_uniform((int)(unsigned)(add_base & 0xffffffffu));
    const unsigned ab_hi = (unsigned)srx_uniform((int)(unsigned)(add_base >> 32));
    return srx_rsrc(a.add + (((size_t)ab_hi << 32) | ab_lo), 0xfffffff0u);
  };
  const __amdgpu_buffer_rsrc_t rmask = srx_rsrc(reinterpret_cast<const char*>(a.mask ? a.mask : a.out) +
                                                    ((((size_t)tb_hi << 32) | tb_lo) << (mask16 ? 1 : 2)), 0xfffffff0u);

  // Row offsets of a pixel-mapped output (strided data gradients, PixelShuffle): ONE thread per tile row works out the row's byte
  // offset (two divisions by a float multiply, the output pixel, ~25 VALU instructions) and parks it in LDS; the epilogue below reads
  // its 16 rows back with four ds_read_b128.  Round 6: every lane used to recompute all 16 of its rows -- 400 VALU instructions per
  // wave and 32-row block, i.e. matrix time (the f32 MFMA runs on the vector ALUs, tools/probe/mfma_valu.hip), 70 % of the launch on
  // the short-K parity classes of the discriminator's first stride-2 data gradient (PMC).  The table sits at the end of the staging
  // ring (free since the barrier behind the k loop; the fold and the statistics use its start).
  unsigned* rowtab = reinterpret_cast<unsigned*>(reinterpret_cast<float*>(smem) + KS * RING * (BMT + BN) * BKL) - ((BMT + 3) & ~3);
  if (!a.linear_out) {
    for (int rr = (int)threadIdx.x; rr < BMT; rr += NT) {
      const int m = m0 + rr;
      rowtab[rr] = m < a.M ? 4u * (unsigned)(out_elem(m) - tile_base) : 4u * (unsigned)(out_elem(m0) - tile_base);
    }
    __syncthreads();
  }
  float bv[TN];
  unsigned ocol[TN];  // byte offset of the column inside its output row
  bool cok[TN];       // this lane stores the column
  bool cmk[TN];       // ... and the column takes the activation mask
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + wn * WN + j * 32 + l31;
    cmk[j] = col >= a.mask_lo && col < a.mask_hi;
    int bidx = col, oc = col;
    if (a.out_shuffle) {
      const int ij = col / a.out_shuffle, cc = col - ij * a.out_shuffle;
      bidx = cc * 4 + ij;
      oc = (((ij >> 1) * a.Wo) + (ij & 1)) * a.Co + cc;  // offset inside the 2x2 output block
    }
    ocol[j] = 4u * (unsigned)oc;
    cok[j] = lead && col < a.Cs;
    bv[j] = (a.bias && col < a.Cn) ? a.bias[bidx] : 0.f;
  }
  float csum[TN], csq[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) { csum[j] = 0.f; csq[j] = 0.f; }

  auto store_tile = [&](auto linear, auto accum, auto o16) {  // one copy per addressing / epilogue mode (bit 0: addend, bit 1: mask), chosen by ONE branch
    constexpr int MODE = decltype(accum)::value;
    constexpr bool OUT16 = decltype(o16)::value;
    __amdgpu_buffer_rsrc_t radd = rout;
    bool cak[TN];  // the column takes the addend
    if constexpr ((MODE & 1) != 0) {
      radd = make_radd();
#pragma unroll
      for (int j = 0; j < TN; ++j) cak[j] = n0 + wn * WN + j * 32 + l31 < a.add_hi;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      // Addend / mask values of the whole 32-row block are requested BEFORE the first store: vmcnt retires in issue
      // order, so a load issued behind a store cannot be consumed until that store has landed, and an epilogue that
      // alternates load -> store per element pays a full memory round trip for each of its 16 x TN elements
      // (+4.5 us on a 15 us dense-block data gradient).
      unsigned offs[16][TN];
      float av[MODE & 1 ? 16 : 1][TN], mv[MODE & 2 ? 16 : 1][TN];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const bool mok = m < a.M;
        const unsigned rowoff = decltype(linear)::value ? 4u * (unsigned)((m - m0) * a.Co) : rowtab[m - m0];
        const unsigned arow = decltype(linear)::value ? 4u * (unsigned)((m - m0) * a.add_ld) : rowoff;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const unsigned off = (mok && cok[j]) ? rowoff + ocol[j] : 0xffffffffu;
          offs[r][j] = off;
          // (a column past add_hi reads nothing -- an out-of-range offset returns 0)
          if (MODE & 1)
            av[r][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                radd, (int)((mok && cok[j] && cak[j]) ? arow + ocol[j] : 0xffffffffu), 0, 0));
          // (a column outside the mask range reads nothing: its offset is pointed out of range)
          if (MODE & 2) {
            const unsigned moff = cmk[j] ? off : 0xffffffffu;
            if (mask16)
              mv[r][j] = __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rmask, (int)(moff == 0xffffffffu ? moff : moff >> 1), 0, 0) << 16);
            else
              mv[r][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rmask, (int)moff, 0, 0));
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const bool mok = m < a.M;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          float v = acc[i][j][r] + bv[j];
          const float vs = mok ? v : 0.f;
          csum[j] += vs;
          csq[j] += vs * vs;
          v = v > 0.f ? v : v * a.slope;
          if (MODE & 1) v = v * a.oscale + av[r][j] * a.ascale;
          if (MODE & 2) v = (cmk[j] && !(mv[r][j] > 0.f)) ? v * a.mask_slope : v;
          if (OUT16)
            __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (__bf16)v), rout,
                                                  (int)(offs[r][j] == 0xffffffffu ? 0xffffffffu : offs[r][j] >> 1), 0, 0);
          else
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rout, offs[r][j], 0, 0);
        }
      }
    }
  };
  using M0 = std::integral_constant<int, 0>; using M1 = std::integral_constant<int, 1>;
  using M2 = std::integral_constant<int, 2>; using M3 = std::integral_constant<int, 3>;
  const int emode = (a.add ? 1 : 0) | (a.mask ? 2 : 0);
  if constexpr (PR == 2) {  // bf16 storage: linear outputs, no addend (the host code asks for nothing else)
    if (out16) {
      if (emode & 2) store_tile(std::true_type{}, M2{}, std::true_type{});
      else store_tile(std::true_type{}, M0{}, std::true_type{});
    } else {
      if (emode & 2) store_tile(std::true_type{}, M2{}, std::false_type{});
      else store_tile(std::true_type{}, M0{}, std::false_type{});
    }
  } else if (a.linear_out) {
    if (emode == 0) store_tile(std::true_type{}, M0{}, std::false_type{});
    else if (emode == 1) store_tile(std::true_type{}, M1{}, std::false_type{});
    else if (emode == 2) store_tile(std::true_type{}, M2{}, std::false_type{});
    else store_tile(std::true_type{}, M3{}, std::false_type{});
  } else {
    if (emode == 0) store_tile(std::false_type{}, M0{}, std::false_type{});
    else if (emode == 1) store_tile(std::false_type{}, M1{}, std::false_type{});
    else if (emode == 2) store_tile(std::false_type{}, M2{}, std::false_type{});
    else store_tile(std::false_type{}, M3{}, std::false_type{});
  }

  float xs1 = 0.f, xs2 = 0.f;
  if (XR > 0 && has_x) {  // the extra 16x16 block: rows BM + 4(l>>4) + reg, column 16 wave + (l&15)
    const int col = n0 + 16 * wave + xi;
    int bidx = col, oc = col;
    if (a.out_shuffle) {
      const int ij = col / a.out_shuffle, cc = col - ij * a.out_shuffle;
      bidx = cc * 4 + ij;
      oc = (((ij >> 1) * a.Wo) + (ij & 1)) * a.Co + cc;
    }
    const float xb = (a.bias && col < a.Cn) ? a.bias[bidx] : 0.f;
    const bool xok = col < a.Cs;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + BM + 4 * xg + r;
      const bool mok = m < a.M;
      const unsigned rowoff = a.linear_out ? 4u * (unsigned)((m - m0) * a.Co) : rowtab[m - m0];
      float v = accx[r] + xb;
      const float vs = mok ? v : 0.f;
      xs1 += vs;
      xs2 += vs * vs;
      v = v > 0.f ? v : v * a.slope;
      const unsigned off = (mok && xok) ? rowoff + 4u * (unsigned)oc : 0xffffffffu;
      if (a.add) {
        const __amdgpu_buffer_rsrc_t radd = make_radd();
        const unsigned arow = a.linear_out ? 4u * (unsigned)((m - m0) * a.add_ld) : rowoff;
        const unsigned aoff = (mok && xok && col < a.add_hi) ? arow + 4u * (unsigned)oc : 0xffffffffu;
        v = v * a.oscale + a.ascale * __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(radd, (int)aoff, 0, 0));
      }
      if (a.mask && col >= a.mask_lo && col < a.mask_hi) {
        const float mv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rmask, (int)off, 0, 0));
        v = mv > 0.f ? v : v * a.mask_slope;
      }
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rout, off, 0, 0);
    }
  }

  if (a.part) {  // per-channel sum / sum of squares of this row block (training-mode BatchNorm)
    constexpr int RED_ROWS = WAVES_M + (XR > 0 ? 1 : 0);
    __syncthreads();  // everyone is done with the staging buffers
    float* red = reinterpret_cast<float*>(smem);  // [RED_ROWS][BN][2]
    if (XR > 0 && has_x) {
      xs1 += __shfl_xor(xs1, 16, 64); xs1 += __shfl_xor(xs1, 32, 64);
      xs2 += __shfl_xor(xs2, 16, 64); xs2 += __shfl_xor(xs2, 32, 64);
      if (xg == 0) {
        red[(WAVES_M * BN + 16 * wave + xi) * 2 + 0] = xs1;
        red[(WAVES_M * BN + 16 * wave + xi) * 2 + 1] = xs2;
      }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float s = csum[j] + __shfl_xor(csum[j], 32, 64);
      float s2 = csq[j] + __shfl_xor(csq[j], 32, 64);
      if (lead && h == 0) {
        const int c = wn * WN + j * 32 + l31;
        red[(wm * BN + c) * 2 + 0] = s;
        red[(wm * BN + c) * 2 + 1] = s2;
      }
    }
    __syncthreads();
    if (lead && tid < BN) {
      float s = 0.f, s2 = 0.f;
#pragma unroll
      for (int w = 0; w < RED_ROWS; ++w) { s += red[(w * BN + tid) * 2]; s2 += red[(w * BN + tid) * 2 + 1]; }
      const int col = n0 + tid;
      if (col < a.Cn) {
        a.part[((size_t)mt * a.Cn + col) * 2 + 0] = s;
        a.part[((size_t)mt * a.Cn + col) * 2 + 1] = s2;
      }
    }
  }
}

template <int BM, int BN, int WM, int WN, int KS, int XR, int PR, int BIG = 0>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64 * KS) void gconv_kernel(const GArgs a) {
  // the fused-upsample gather is a second copy of the body, entered by ONE scalar branch: the common loop is untouched
  if (a.up) gconv_body<BM, BN, WM, WN, KS, XR, PR, 1, BIG>(a, blockIdx.x);
  else gconv_body<BM, BN, WM, WN, KS, XR, PR, 0, BIG>(a, blockIdx.x);
}

// several independent gather-GEMMs in one launch: the stride-parity classes of a strided data
// gradient (four small problems that would each under-fill the chip and pay a kernel boundary)
struct GMulti {
  int n;
  int first[5];  // first[i] = first workgroup of problem i; first[n] = grid size
  GArgs g[4];
};
// KS = 2 (round 6, the 64 x 64 tile only): two wave groups share a tile's k-chunks (gconv_body) -- the deep, small-M classes of
// the discriminators' last strided layers (576..4608 rows, K up to 2048 per class) are bound by the serial chunk chain of their
// heaviest class's workgroups, not by the chip: 64 x 64 tiles give four times the workgroups of 128 x 64 and the in-workgroup
// split halves each chain.
template <int BM, int BN, int WM, int WN, int XR, int PR = 0, int KS = 1>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64 * KS) void gconv_multi_kernel(const GMulti m) {
  int ci = 0;
  while (ci + 1 < m.n && (int)blockIdx.x >= m.first[ci + 1]) ++ci;
  const GArgs a = m.g[ci];
  gconv_body<BM, BN, WM, WN, KS, XR, PR>(a, blockIdx.x - m.first[ci]);
}

// ---------------------------------------------------------------------------
// Strided data gradient, all stride-parity classes of a tile in ONE workgroup and ONE pipelined K loop (round 6).
//
// gconv_multi_kernel runs the classes as separate workgroups: for the 3x3 / stride-2 layers of the discriminators that is four
// gather-GEMMs with 1, 2, 2 and 4 taps -- K loops of 2..8 chunks at 64 channels, each with its own k table, row set-up, pipeline
// fill and epilogue (34-44 % MFMA-busy under the counters).  The classes of a layer whose extents are multiples of the stride
// share EVERYTHING but their taps, their weight matrix and a constant output offset: the same M grid (one row per gradient
// pixel neighbourhood), hence the same A rows, bounds and row offsets.  Here a workgroup owns one (row tile, column tile) and
// walks the classes back to back: one chunk sequence -- each class's chunks padded to whole trips of the three-slot ring, the
// padding chunks loading nothing and multiplying nothing --, one table entry per chunk (tap coordinates, gather offset, the
// chunk's place in its class's weight matrix: wave-uniform, so they live in SGPRs and the per-thread k table of gconv_body
// disappears), the loads of class c+1 already in flight while class c's accumulators are stored.  A chunk lies inside one tap
// (channels per tap a multiple of 32: checked on the host; other layers keep gconv_multi_kernel).
// ---------------------------------------------------------------------------
struct S2Class {
  int nth, ntw, dh0, dw0;  // taps of the class and its most negative tap
  int chunks, pstart;      // k-chunks (32 floats) of the class; its first chunk in the padded sequence
  int oh_off, ow_off;      // output parity
  unsigned wbase, kpb;     // byte offset of the class's weight matrix in the packed buffer; bytes per row of it (Kp * 4)
};
struct GFused {
  GArgs g;                 // what the classes share (w / w_bytes: the WHOLE packed data-gradient buffer)
  int nclass, cpt, ptotal; // classes; chunks per tap (Ck / 32); chunks of the padded sequence
  S2Class c[4];
};

// (the 64-column tiles: at most 128 registers, so that two 8-wave workgroups share a CU -- one's epilogues and dead steps
// under the other's MFMAs)
template <int BM, int BN, int WM, int WN, int XR, int PR>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64, (BM / WM) * (BN / WN) == 8 && BN == 64 ? 4 : 1) void gconv_s2f_kernel(const GFused f) {
  static_assert(PR == 0 || (PR == 1 && XR == 0), "fp32, or bf16 products without the 16-row extension");
  const GArgs& a = f.g;
  constexpr int BKL = PR == 1 ? BK / 2 : BK;
  constexpr int NS = PR == 1 ? 2 : 4;
  constexpr int BMT = BM + XR;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int WAVES_N = BN / WN, WAVES_M = BM / WM;
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr int RPP = NT / 8;
  constexpr int RA = (BMT + RPP - 1) / RPP, RB = BN / RPP;
  static_assert(RA >= 1 && RB >= 1, "tile too small for the thread count");
  static_assert(XR == 0 || BN / 16 <= NT / 64, "one extra 16x16 block per wave at most");
  constexpr int RING = 3;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = srx_uniform(tid >> 6);
  float* sA = reinterpret_cast<float*>(smem);
  float* sB = sA + RING * BMT * BKL;
  int4* ctab = reinterpret_cast<int4*>(sA + RING * (BMT + BN) * BKL);        // [ptotal + 3]
  unsigned* rowtab = reinterpret_cast<unsigned*>(ctab + (f.ptotal + 3));      // [BMT]
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int tile = srx_uniform((int)blockIdx.x);
  const int nt = srx_uniform(tile / a.mtiles), mt = tile - nt * a.mtiles;
  const int m0 = mt * BMT, n0 = nt * BN;
  const int q = tid & 7, r0 = tid >> 3;
  auto split_row = [&](int m, int& n, int& mh, int& mw) {
    int rem;
    srx_divmod(m, a.HmWm, a.inv_HmWm, n, rem);
    srx_divmod(rem, a.Wm, a.inv_Wm, mh, mw);
  };
  const __amdgpu_buffer_rsrc_t rin = srx_rsrc(a.in, (unsigned)a.in_bytes);
  const __amdgpu_buffer_rsrc_t rw = srx_rsrc(a.w, a.w_bytes);

  // ---- chunk table: one entry per chunk of the padded sequence (+ 3 dead ones the pipeline runs ahead into)
  for (int e = tid; e < f.ptotal + 3; e += NT) {
    int4 ent = {(int)(INVALID & 0xffff), 0, 0, 0};  // dead: taps out of every image, weight row pitch 0
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (c < f.nclass) {
        const S2Class& k = f.c[c];
        const int j = e - k.pstart;
        if (j >= 0 && j < k.chunks) {
          const int tap = j / f.cpt, cin = j - tap * f.cpt;
          const int th = tap / k.ntw, tw = tap - th * k.ntw;
          const int dh = k.dh0 + th, dw = k.dw0 + tw;
          ent.x = (int)((unsigned)(dh & 0xffff) | ((unsigned)dw << 16));
          ent.y = ((dh * a.Wi + dw) * a.Ci + cin * BK) * 4;
          ent.z = (int)(k.wbase + (unsigned)j * (BK * 4));
          ent.w = (int)k.kpb;
        }
      }
    }
    ctab[e] = ent;
  }
  // ---- per-thread rows of the A tile: the same for every class
  int rih[RA], riw[RA];
  unsigned rbase[RA];
#pragma unroll
  for (int p = 0; p < RA; ++p) {
    const int m = m0 + r0 + RPP * p;
    if (m < a.M && r0 + RPP * p < BMT) {
      int n, mh, mw;
      split_row(m, n, mh, mw);
      rih[p] = mh; riw[p] = mw;
      rbase[p] = 4u * (unsigned)(((n * a.Hi + mh) * a.Wi + mw) * a.Ci) + 16u * (unsigned)q;
    } else {
      rih[p] = INVALID; riw[p] = 0; rbase[p] = 0;
    }
  }
  unsigned wrow[RB];
#pragma unroll
  for (int p = 0; p < RB; ++p) wrow[p] = (unsigned)(n0 + r0 + RPP * p);
  // ---- byte offsets of the tile's output rows for parity (0, 0), relative to the tile's first row (a class adds a constant)
  auto out_elem = [&](int m) -> size_t {
    int n, mh, mw;
    split_row(m, n, mh, mw);
    return ((size_t)(n * a.Ho + mh * a.out_stride) * a.Wo + mw * a.out_stride) * a.Co;
  };
  const size_t tile_base = out_elem(m0);
  for (int rr = tid; rr < BMT; rr += NT) {
    const int m = m0 + rr;
    rowtab[rr] = m < a.M ? 4u * (unsigned)(out_elem(m) - tile_base) : 0u;
  }
  __syncthreads();

  f32x16 acc[TM][TN];
  f32x4 accx = {0.f, 0.f, 0.f, 0.f};
  f32x4 ra0[RA], rb0[RB], ra1[RA], rb1[RB], ra2[RA], rb2[RB];
  auto gload = [&](int kc, f32x4 (&ra)[RA], f32x4 (&rb)[RB]) {
    const int4 ct = ctab[kc];  // (one address for the whole wave: a broadcast read)
    const int cx = srx_uniform(ct.x), cy = srx_uniform(ct.y), cz = srx_uniform(ct.z), cw = srx_uniform(ct.w);
    const int dh = (int)(short)(cx & 0xffff), dw = cx >> 16;
#pragma unroll
    for (int p = 0; p < RA; ++p) {
      const int ih = rih[p] + dh, iw = riw[p] + dw;
      const bool ok = ((unsigned)ih < (unsigned)a.Hi) && ((unsigned)iw < (unsigned)a.Wi);
      ra[p] = srx_bload(rin, ok ? rbase[p] + (unsigned)cy : 0xffffffffu, 0);  // out of range (padding taps, dead chunks) reads 0
    }
#pragma unroll
    for (int p = 0; p < RB; ++p)
      rb[p] = srx_bload(rw, cw ? wrow[p] * (unsigned)cw + 16u * (unsigned)q : 0xffffffffu, (unsigned)cz);
  };
  const int wchunk = PR == 1 ? (((q >> 1) ^ ((r0 >> 2) & 3)) * 4 + (q & 1) * 2) : (q ^ ((r0 >> 1) & 7)) * 4;
  auto put = [&](float* dst, const f32x4& v) {
    if (PR == 1) {
      typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
      const bf16x2 lo = {(__bf16)v[0], (__bf16)v[1]}, hi = {(__bf16)v[2], (__bf16)v[3]};
      *reinterpret_cast<uint2*>(dst) = make_uint2(__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi));
    } else {
      *reinterpret_cast<f32x4*>(dst) = v;
    }
  };
  auto swrite = [&](int buf, const f32x4 (&ra)[RA], const f32x4 (&rb)[RB]) {
    float* dA = sA + buf * BMT * BKL;
    float* dB = sB + buf * BN * BKL;
#pragma unroll
    for (int p = 0; p < RA; ++p)
      if (RPP * (p + 1) <= BMT || r0 + RPP * p < BMT) put(dA + (r0 + RPP * p) * BKL + wchunk, ra[p]);
#pragma unroll
    for (int p = 0; p < RB; ++p) put(dB + (r0 + RPP * p) * BKL + wchunk, rb[p]);
  };
  const int h = lane >> 5, l31 = lane & 31;
  const int xr = PR == 1 ? (l31 >> 2) & 3 : (l31 >> 1) & 7;
  const int arow = (wm * WM + l31) * BKL, brow = (wn * WN + l31) * BKL;
  const bool has_x = XR > 0 && wave < BN / 16;
  const int xi = lane & 15, xg = lane >> 4;
  const int xra = BM + xi, xrb = 16 * wave + xi;
  f32x4 af[2][TM], bf[2][TN];
  auto frag = [&](int slot, int s, int set) {
    const float* cA = sA + slot * BMT * BKL + arow;
    const float* cB = sB + slot * BN * BKL + brow;
    const int ch = ((2 * s + h) ^ xr) * 4;
#pragma unroll
    for (int i = 0; i < TM; ++i) af[set][i] = *reinterpret_cast<const f32x4*>(cA + i * 32 * BKL + ch);
#pragma unroll
    for (int j = 0; j < TN; ++j) bf[set][j] = *reinterpret_cast<const f32x4*>(cB + j * 32 * BKL + ch);
  };
  auto mma = [&](int set) {
    __builtin_amdgcn_s_setprio(1);
    if (PR) {
      typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[set][i]),
                                                              __builtin_bit_cast(bf16x8, bf[set][j]), acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      return;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][i][e], bf[set][j][e], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
  auto extra = [&](int slot) {
    const float* xA = sA + slot * BMT * BKL + xra * BKL;
    const float* xB = sB + slot * BN * BKL + xrb * BKL;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const f32x4 fa = *reinterpret_cast<const f32x4*>(xA + (((xg + 4 * u) ^ ((xra >> 1) & 7)) * 4));
      const f32x4 fb = *reinterpret_cast<const f32x4*>(xB + (((xg + 4 * u) ^ ((xrb >> 1) & 7)) * 4));
#pragma unroll
      for (int e = 0; e < 4; ++e) accx = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[e], fb[e], accx, 0, 0, 0);
    }
  };
  // one step of gconv_body's pipeline (see there); `live`: the chunk is a real one of the class
  auto step = [&](int kc, int slot, f32x4 (&ra_next)[RA], f32x4 (&rb_next)[RB], f32x4 (&ra_free)[RA], f32x4 (&rb_free)[RB]) {
    const int nslot = slot == 2 ? 0 : slot + 1;
    swrite(nslot, ra_next, rb_next);
    gload(kc + 3, ra_free, rb_free);
    const bool live = srx_uniform(ctab[kc].w) != 0;
    if (NS == 2) {
      frag(slot, 1, 1);
      if (live) mma(0);
      __syncthreads();
      frag(nslot, 0, 0);
      if (live) mma(1);
      return;
    }
    frag(slot, 1, 1);
    if (live) mma(0);
    frag(slot, 2, 0);
    if (live) mma(1);
    __syncthreads();
    frag(slot, 3, 1);
    if (live) mma(0);
    frag(nslot, 0, 0);
    if (XR > 0 && has_x && live) extra(slot);
    if (live) mma(1);
  };

  // ---- epilogue constants (class-independent)
  const unsigned tb_lo = (unsigned)srx_uniform((int)(unsigned)(tile_base & 0xffffffffu));
  const unsigned tb_hi = (unsigned)srx_uniform((int)(unsigned)(tile_base >> 32));
  unsigned ocol[TN];
  bool cok[TN], cmk[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + wn * WN + j * 32 + l31;
    cmk[j] = col >= a.mask_lo && col < a.mask_hi;
    ocol[j] = 4u * (unsigned)col;
    cok[j] = col < a.Cs;
  }
  auto store_class = [&](auto masked, const __amdgpu_buffer_rsrc_t rout, const __amdgpu_buffer_rsrc_t rmask) {
    constexpr bool MASK = decltype(masked)::value;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int hr = 0; hr < 16; hr += 8) {  // eight rows at a time: the offsets and mask values of a half block are what the
        unsigned offs[8][TN];               // epilogue holds on top of the three load stages in flight (two workgroups per CU)
        float mv[MASK ? 8 : 1][TN];
#pragma unroll
        for (int r8 = 0; r8 < 8; ++r8) {  // (every mask value of the half block is requested before its first store: see gconv_body)
          const int r = hr + r8;
          const int rl = wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          const bool mok = m0 + rl < a.M;
          const unsigned rowoff = rowtab[rl];
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const unsigned off = (mok && cok[j]) ? rowoff + ocol[j] : 0xffffffffu;
            offs[r8][j] = off;
            if (MASK) mv[r8][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rmask, (int)(cmk[j] ? off : 0xffffffffu), 0, 0));
          }
        }
#pragma unroll
        for (int r8 = 0; r8 < 8; ++r8)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            float v = acc[i][j][hr + r8];
            if (MASK) v = (cmk[j] && !(mv[r8][j] > 0.f)) ? v * a.mask_slope : v;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rout, offs[r8][j], 0, 0);
          }
      }
    if (XR > 0 && has_x) {  // the extra 16x16 block: rows BM + 4(l>>4) + reg, column 16 wave + (l&15)
      const int col = n0 + 16 * wave + xi;
      const bool xok = col < a.Cs;
      const bool xmk = MASK && col >= a.mask_lo && col < a.mask_hi;
      unsigned xoff[4];
      float xm[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rl = BM + 4 * xg + r;
        xoff[r] = (m0 + rl < a.M && xok) ? rowtab[rl] + 4u * (unsigned)col : 0xffffffffu;
        xm[r] = MASK ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rmask, (int)(xmk ? xoff[r] : 0xffffffffu), 0, 0)) : 1.f;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = accx[r];
        if (MASK) v = (xmk && !(xm[r] > 0.f)) ? v * a.mask_slope : v;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rout, xoff[r], 0, 0);
      }
    }
  };

  // ---- the pipelined loop over the padded chunk sequence, class by class
  gload(0, ra0, rb0);
  gload(1, ra1, rb1);
  gload(2, ra2, rb2);
  swrite(0, ra0, rb0);
  __syncthreads();
  frag(0, 0, 0);
  int kc = 0;
  for (int c = 0; c < f.nclass; ++c) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    accx = f32x4{0.f, 0.f, 0.f, 0.f};
    const int kend = srx_uniform(c + 1 < f.nclass ? f.c[c + 1].pstart : f.ptotal);
    for (; kc < kend; kc += 3) {  // whole trips of the ring (a class's chunks are padded to a multiple of three)
      step(kc, 0, ra1, rb1, ra0, rb0);
      step(kc + 1, 1, ra2, rb2, ra1, rb1);
      step(kc + 2, 2, ra0, rb0, ra2, rb2);
    }
    // this class's outputs: parity (oh_off, ow_off) of the tile's 2x2 (stride x stride) output blocks
    const S2Class& k = f.c[c];
    const size_t delta = ((size_t)k.oh_off * a.Wo + k.ow_off) * a.Co;
    const size_t cbase = ((((size_t)tb_hi << 32) | tb_lo) + delta) << 2;
    const unsigned cb_lo = (unsigned)srx_uniform((int)(unsigned)(cbase & 0xffffffffu));
    const unsigned cb_hi = (unsigned)srx_uniform((int)(unsigned)(cbase >> 32));
    const size_t cb = ((size_t)cb_hi << 32) | cb_lo;
    const __amdgpu_buffer_rsrc_t rout = srx_rsrc(reinterpret_cast<char*>(a.out) + cb, 0xfffffff0u);
    const __amdgpu_buffer_rsrc_t rmask = srx_rsrc(reinterpret_cast<const char*>(a.mask ? a.mask : a.out) + cb, 0xfffffff0u);
    if (a.mask) store_class(std::true_type{}, rout, rmask);
    else store_class(std::false_type{}, rout, rmask);
  }
}

// finishes the K-split tail tiles: out = act(sum_z partial[z] + bias) and, when asked, the tile's
// per-channel (sum, sum of squares) row of the BatchNorm partial table.  One workgroup per
// (tile, 16-column group): every column's statistics stay inside one workgroup, and a tile with
// many K splits is still drained by BN/16 workgroups rather than one.
template <int BM, int BN>
__global__ __launch_bounds__(256) void tail_fixup_kernel(const GArgs a) {
  constexpr int CG = BN / 16;
  __shared__ f32x4 red[2][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = tid & 3, rl = tid >> 2;  // 4 float4 quads x 64 row lanes
  const int t = blockIdx.x / CG, cg = blockIdx.x % CG;
  const int tile = a.full_tiles + t;
  const int mt = tile % a.mtiles, nt = tile / a.mtiles;
  const int m0 = mt * BM, n0 = nt * BN;
  const float* slab = a.ws + (size_t)t * a.tail_split * (BM * BN);
  const int lc = cg * 16 + q * 4;  // column inside the tile
  const int col = n0 + lc;
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  if (a.bias) {
#pragma unroll
    for (int e = 0; e < 4; ++e) bv[e] = (col + e < a.Cn) ? a.bias[col + e] : 0.f;
  }
  f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
  // All loads first, without branches around them (rows past the tile or the image read a valid row and are dropped
  // afterwards): the K-split partials four at a time, the addend and the mask -- the launch is a few load round trips long,
  // and a load behind a data-dependent branch or a runtime-count loop waits for the one before it (8 -> ~5 us per launch).
  // The partials are added in the order z = 0, 1, 2, ... as before.
  constexpr int NR = (BM + 63) / 64;
  f32x4 v[NR], av[NR], mv[NR];
  bool ok[NR];
  const bool has_add = a.add != nullptr && col < a.Cs && col < a.add_hi;
  const bool has_mask = a.mask != nullptr && col < a.Cs && col >= a.mask_lo && col < a.mask_hi;
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    const int r = rl + 64 * j;
    ok[j] = r < BM && m0 + r < a.M;
    const int mc = min(m0 + (r < BM ? r : rl), a.M - 1);
    av[j] = has_add ? *reinterpret_cast<const f32x4*>(a.add + (size_t)mc * a.add_ld + col) : (f32x4){0.f, 0.f, 0.f, 0.f};
    if (has_mask && a.mask16) {  // (bf16 mask tensor: four values in 8 bytes)
      const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(a.mask) + (size_t)mc * a.Co + col);
      mv[j] = (f32x4){__builtin_bit_cast(float, u.x << 16), __builtin_bit_cast(float, u.x & 0xffff0000u),
                      __builtin_bit_cast(float, u.y << 16), __builtin_bit_cast(float, u.y & 0xffff0000u)};
    } else {
      mv[j] = has_mask ? *reinterpret_cast<const f32x4*>(a.mask + (size_t)mc * a.Co + col) : (f32x4){1.f, 1.f, 1.f, 1.f};
    }
    v[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  for (int z0 = 0; z0 < a.tail_split; z0 += 4) {
    f32x4 t[NR][4];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      const int rc = rl + 64 * j < BM ? rl + 64 * j : rl;
#pragma unroll
      for (int u = 0; u < 4; ++u)
        t[j][u] = *reinterpret_cast<const f32x4*>(slab + (size_t)min(z0 + u, a.tail_split - 1) * (BM * BN) + rc * BN + lc);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (z0 + u < a.tail_split) {
#pragma unroll
        for (int j = 0; j < NR; ++j) v[j] = (z0 + u == 0) ? t[j][u] : v[j] + t[j][u];
      }
  }
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    if (!ok[j]) continue;
    const int m = m0 + rl + 64 * j;
    f32x4 w = v[j] + bv;
    s1 += w;
    s2 += w * w;
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = w[e] > 0.f ? w[e] : w[e] * a.slope;
    if (col < a.Cs) {  // Cs is a multiple of 4
      if (a.add) {  // (add_hi is a multiple of 4 or covers every column)
        w = w * a.oscale;
        if (col < a.add_hi) w += av[j] * a.ascale;
      }
      if (has_mask) {  // (mask ranges are whole quads: channel counts are multiples of 4)
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = mv[j][e] > 0.f ? w[e] : w[e] * a.mask_slope;
      }
      if (a.out16) {
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        const bf16x2 lo = {(__bf16)w[0], (__bf16)w[1]}, hi = {(__bf16)w[2], (__bf16)w[3]};
        *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(a.out) + (size_t)m * a.Co + col) =
            make_uint2(__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi));
      } else {
        *reinterpret_cast<f32x4*>(a.out + (size_t)m * a.Co + col) = w;
      }
    }
  }
  if (a.part) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int o = 4; o < 64; o <<= 1) { s1[e] += __shfl_xor(s1[e], o, 64); s2[e] += __shfl_xor(s2[e], o, 64); }
    if (lane < 4) { red[0][wave * 4 + lane] = s1; red[1][wave * 4 + lane] = s2; }
    __syncthreads();
    if (tid < 4) {
      const f32x4 t1 = red[0][tid] + red[0][4 + tid] + red[0][8 + tid] + red[0][12 + tid];
      const f32x4 t2 = red[1][tid] + red[1][4 + tid] + red[1][8 + tid] + red[1][12 + tid];
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (col + e < a.Cn) {
          a.part[((size_t)mt * a.Cn + col + e) * 2 + 0] = t1[e];
          a.part[((size_t)mt * a.Cn + col + e) * 2 + 1] = t2[e];
        }
    }
  }
}

// ---------------------------------------------------------------------------
// weight gradient: slab[z][n][k] = sum_{m in split z} dy[m][n] * A(m,k)
// workgroup tile 64(n) x 64(k), 4 waves of 32x32, m consumed 32 rows at a time.
// ---------------------------------------------------------------------------
struct WArgs {
  const float* in; const float* dy; float* slab;
  int N, Hi, Wi, Ci, Hm, Wm, M, HmWm;
  float inv_HmWm, inv_Wm;
  int in_stride, nth, ntw, dh0, dw0, Ck, K, Kw;
  int Cd, Cdv, dy_shuffle, Cnw;
  int rows_per_split, ktiles;
  unsigned in_bytes, dy_bytes;  // raw-buffer ranges (see GArgs)
  // A thread's rows advance by 32 per chunk; its pixel coordinates and both element offsets follow
  // incrementally from these host-computed steps (no division in the loop):
  //   s_c = 32 % Wm, s_rm = (32 / Wm) % Hm; dX0/dD0 plain step, dX1/dD1 extra on a column wrap
  //   (mw -= Wm, mh += 1), dX2 extra on a row wrap (mh -= Hm, next image)
  int s_c, s_rm, dX0, dX1, dX2, dD0, dD1;
  // bias gradient db[n] = sum_m dy[m][n]: the k-tile-0 workgroups already stage every dy row of their
  // row split, so they add the column sums up on the way and write one [Cnw] row per split here
  // (null: not wanted); wgrad_reduce_kernel sums the rows.  Replaces two column-sum launches per layer.
  float* bslab;
  int nsplit, nprob;
};

// Several weight-gradient problems of ONE geometry in one launch (blockIdx.y = problem): the 33 residual convs
// of the SRGAN generator are 0.68 GFLOP each -- alone, such a problem is all pipeline fill and slab reduction
// (19 + 5 us for 4.3 us of matrix work) -- and nothing downstream waits for a weight gradient before the
// optimiser, so the host collects them during the backward pass and issues them together: long loops, every CU
// holding several workgroups, one reduction.  Problems may also be SEGMENTS of one gradient (the discriminator's
// real and fake passes): `per_out` consecutive problems are summed into one output.
constexpr int WG_MAXP = 72;
struct WMulti {
  WArgs a;
  const float* x[WG_MAXP];
  const float* dy[WG_MAXP];
};
struct WReduce {
  float* dw[WG_MAXP];
  float* db[WG_MAXP];  // entries may be null
  // multiplies the output's weight and bias gradient: the layer's dy tensor stands for scale * dy (a dense block's conv5
  // sees the block's output gradient times scale_ratio, which is then never written out)
  float scale[WG_MAXP];
  // Paired problems (rows_lo > 0): the tile rows [0, rows_lo) are the gradient of one conv (dw, db; cin_lo input channels)
  // and the rows above that of a second conv (dw_hi, db_hi; Cin input channels) that reads the SAME input buffer and
  // whose output gradient is the adjacent channel slice -- two convs of a dense block (esrgan/residual.py:81-85), which
  // alone are 32 columns wide and would each leave half of every 64-column tile multiplying padding.
  float* dw_hi[WG_MAXP];
  float* db_hi[WG_MAXP];
  int rows_lo, cin_lo;
};

// PR = 1: bf16 products (the autocast mode).  The contraction runs over pixels, so an MFMA operand is eight
// CONSECUTIVE ROWS of one column.  A thread therefore loads two adjacent rows (2 r0, 2 r0 + 1), rounds them
// to bf16 and stores them interleaved -- one 32-bit word per (row pair, column) -- so that a lane collects
// its eight rows as four words; two v_mfma_f32_32x32x16_bf16 per chunk replace sixteen fp32 MFMAs.
template <int PR>
__global__ __launch_bounds__(256) void wgrad_kernel(const WMulti mp) {
  const WArgs& a = mp.a;
  // Work item = (tile, problem, row split), tile fastest.  The (up to 9 x Cout/64) tiles of one problem's row split read the
  // same dy and x rows, so they should share an L2: workgroups are dealt round-robin over the 8 XCDs, and this remap gives
  // every XCD a contiguous range of work items (MI355X_MICROARCH.md, XCD placement; a speed matter only -- with the plain
  // order the 33-problem group read 1.23 GB through the fabric per launch, 8x its operands).
  int wi;
  {
    const int W = (int)gridDim.x, b = (int)blockIdx.x, xcd = b & 7, slot = b >> 3, q = W >> 3, r = W & 7;
    wi = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int tiles_ = a.ktiles * (a.Cnw / 64);
  const int tile_id = wi % tiles_, rest_ = wi / tiles_;
  const int prob = srx_uniform(rest_ % a.nprob), zsplit = srx_uniform(rest_ / a.nprob);
  // LDS image of a chunk (32 pixels x 64 columns).  fp32: [pixel][64] floats.  bf16: [pixel][64] bf16 in rows of PSTR = 192
  // bytes (the 64-byte pad puts the four rows of a transposing read on different banks): operands are stored as they
  // arrive -- one 8-byte store per loaded quad -- and an MFMA operand (eight consecutive PIXELS of one column) is two
  // ds_read_b64_tr_b16, the hardware's transposing read, instead of four scalar reads of hand-interleaved row pairs
  constexpr int PSTR = 192;
  constexpr int BUF_FLOATS = PR ? 32 * PSTR / 4 : 32 * 64;
  __shared__ __attribute__((aligned(16))) float sD[2][BUF_FLOATS];
  __shared__ __attribute__((aligned(16))) float sX[2][BUF_FLOATS];
  const int tid = threadIdx.x, lane = tid & 63, wave = srx_uniform(tid >> 6);
  const int ntile = srx_uniform(tile_id / a.ktiles), kt = srx_uniform(tile_id - ntile * a.ktiles);
  const int k0 = kt * 64, n0 = ntile * 64;
  const int q = tid & 15, r0 = tid >> 4;
  const __amdgpu_buffer_rsrc_t rx_ = srx_rsrc(mp.x[prob], a.in_bytes), rd_ = srx_rsrc(mp.dy[prob], a.dy_bytes);

  // this thread's fixed k (A gather) and fixed dy column
  const int k = k0 + 4 * q;
  const bool kvalid = k < a.K;
  int dh = 0, dw = 0, kc = 0;
  if (kvalid) {
    const int tap = k / a.Ck;
    kc = k - tap * a.Ck;
    const int th = tap / a.ntw, tw = tap - th * a.ntw;
    dh = a.dh0 + th;
    dw = a.dw0 + tw;
  }
  const int col = n0 + 4 * q;
  const bool cvalid = col < a.Cdv;
  int sh_i = 0, sh_j = 0, sh_c = col;
  if (a.dy_shuffle) {
    const int ij = col / a.dy_shuffle;
    sh_c = col - ij * a.dy_shuffle;
    sh_i = ij >> 1;
    sh_j = ij & 1;
  }

  const int mbeg = zsplit * a.rows_per_split;
  const int mend = min(a.M, mbeg + a.rows_per_split);

  // Four register stages: a workgroup that is alone on its CU (small layers: one row split per CU)
  // multiplies a chunk in ~0.45 us but waits ~2 us for a load, so chunk c+4 is requested while c runs.
  f32x4 rd0[2], rx0[2], rd1[2], rx1[2], rd2[2], rx2[2], rd3[2], rx3[2];
  // row state of this thread's two rows (r0 + 16p of the current chunk; 2 r0 + p for bf16); chunks are
  // requested strictly in order, so every gload advances the state by one chunk
  int rm[2], rmh[2], rmw[2];
  unsigned rox[2], rod[2];  // element offsets: x at (n, mh*stride + dh, mw*stride + dw, kc); dy at the thread's column
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int m = PR ? mbeg + 2 * r0 + p : mbeg + r0 + 16 * p;
    int n, rem, mh, mw;
    srx_divmod(m, a.HmWm, a.inv_HmWm, n, rem);
    srx_divmod(rem, a.Wm, a.inv_Wm, mh, mw);
    rm[p] = m; rmh[p] = mh; rmw[p] = mw;
    rox[p] = (unsigned)(((n * a.Hi + mh * a.in_stride + dh) * a.Wi + mw * a.in_stride + dw) * a.Ci + kc);
    rod[p] = (unsigned)(a.dy_shuffle ? ((n * 2 * a.Hm + 2 * mh + sh_i) * (2 * a.Wm) + 2 * mw + sh_j) * a.Cd + sh_c
                                     : m * a.Cd + col);
  }
  auto gload = [&](f32x4 (&rd)[2], f32x4 (&rx)[2]) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const bool valid = rm[p] < mend;
      const int ih = rmh[p] * a.in_stride + dh, iw = rmw[p] * a.in_stride + dw;
      const bool okx = valid && kvalid && ((unsigned)ih < (unsigned)a.Hi) && ((unsigned)iw < (unsigned)a.Wi);
      rx[p] = srx_bload(rx_, okx ? 4u * rox[p] : 0xffffffffu, 0);  // out of range reads 0
      rd[p] = srx_bload(rd_, (valid && cvalid) ? 4u * rod[p] : 0xffffffffu, 0);
      // advance 32 rows
      rm[p] += 32; rmw[p] += a.s_c; rmh[p] += a.s_rm; rox[p] += (unsigned)a.dX0; rod[p] += (unsigned)a.dD0;
      const bool wc = rmw[p] >= a.Wm;
      rmw[p] -= wc ? a.Wm : 0; rmh[p] += wc ? 1 : 0;
      rox[p] += wc ? (unsigned)a.dX1 : 0u; rod[p] += wc ? (unsigned)a.dD1 : 0u;
      const bool wr = rmh[p] >= a.Hm;
      rmh[p] -= wr ? a.Hm : 0;
      rox[p] += wr ? (unsigned)a.dX2 : 0u;
    }
  };
  const bool want_bias = a.bslab != nullptr && kt == 0;  // workgroup-uniform
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
  auto swrite = [&](int buf, const f32x4 (&rd)[2], const f32x4 (&rx)[2]) {
    if (want_bias) bsum += rd[0] + rd[1];  // (fp32 values, whatever the product precision)
    if (PR) {  // row 2 r0 + p, columns 4q .. 4q + 3, rounded to bf16
      typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const bf16x4 pd = {(__bf16)rd[p][0], (__bf16)rd[p][1], (__bf16)rd[p][2], (__bf16)rd[p][3]};
        const bf16x4 px = {(__bf16)rx[p][0], (__bf16)rx[p][1], (__bf16)rx[p][2], (__bf16)rx[p][3]};
        unsigned char* bD = reinterpret_cast<unsigned char*>(&sD[buf][0]) + (2 * r0 + p) * PSTR + 8 * q;
        unsigned char* bX = reinterpret_cast<unsigned char*>(&sX[buf][0]) + (2 * r0 + p) * PSTR + 8 * q;
        *reinterpret_cast<bf16x4*>(bD) = pd;
        *reinterpret_cast<bf16x4*>(bX) = px;
      }
      return;
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      *reinterpret_cast<f32x4*>(&sD[buf][(r0 + 16 * p) * 64 + q * 4]) = rd[p];
      *reinterpret_cast<f32x4*>(&sX[buf][(r0 + 16 * p) * 64 + q * 4]) = rx[p];
    }
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int h = lane >> 5, l31 = lane & 31;
  const int wn = wave >> 1, wk = wave & 1;

  auto compute = [&](int buf) {
    if (PR) {  // MFMA s contracts pixels 16 s + 8 h .. + 7: two transposing reads of 4 pixels x 16 columns per 16-lane group
      typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
      typedef short s16x4 __attribute__((ext_vector_type(4)));
      typedef short s16x8 __attribute__((ext_vector_type(8)));
      typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
      const int li = lane & 15, colh = 16 * ((lane >> 4) & 1) + 4 * (li & 3), rq = li >> 2;
      const unsigned char* bD = reinterpret_cast<const unsigned char*>(&sD[buf][0]) + (8 * h + rq) * PSTR + 2 * (wn * 32 + colh);
      const unsigned char* bX = reinterpret_cast<const unsigned char*>(&sX[buf][0]) + (8 * h + rq) * PSTR + 2 * (wk * 32 + colh);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const s16x4 d0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(bD + (16 * s) * PSTR));
        const s16x4 d1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(bD + (16 * s + 4) * PSTR));
        const s16x4 x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(bX + (16 * s) * PSTR));
        const s16x4 x1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(bX + (16 * s + 4) * PSTR));
        const s16x8 fd = {d0[0], d0[1], d0[2], d0[3], d1[0], d1[1], d1[2], d1[3]};
        const s16x8 fx = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fd), __builtin_bit_cast(bf16x8, fx), acc, 0, 0, 0);
      }
      return;
    }
    const float* cD = &sD[buf][h * 64 + wn * 32 + l31];
    const float* cX = &sX[buf][h * 64 + wk * 32 + l31];
    // Operands run two MFMA pairs ahead of the multiplies (one ds_read2st64_b32 fetches a pair's d or x): left to itself
    // the compiler read each pair right before its MFMAs and waited for it -- an LDS round trip per 128 MFMA cycles, which
    // three waves per SIMD did not hide (PMC, round 3: MFMA pipe 54 % busy, 61 % of wave time in s_waitcnt)
    float dv[16], xv[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) { dv[s] = cD[s * 128]; xv[s] = cX[s * 128]; }
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(dv[s], xv[s], acc, 0, 0, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      if (i < 6) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    }
  };
  // every step issues the same four loads (rows past `mend` are pointed out of range and read 0), so
  // the prefetch waits are exact vmcnt counts -- see gconv_body
  gload(rd0, rx0);
  gload(rd1, rx1);
  gload(rd2, rx2);
  gload(rd3, rx3);
  swrite(0, rd0, rx0);
  __syncthreads();
  for (int mb = mbeg; mb < mend; mb += 128) {
    gload(rd0, rx0);  // chunk at mb + 128
    compute(0);
    swrite(1, rd1, rx1);
    __syncthreads();
    gload(rd1, rx1);
    if (mb + 32 < mend) compute(1);
    swrite(0, rd2, rx2);
    __syncthreads();
    gload(rd2, rx2);
    if (mb + 64 < mend) compute(0);
    swrite(1, rd3, rx3);
    __syncthreads();
    gload(rd3, rx3);
    if (mb + 96 < mend) compute(1);
    swrite(0, rd0, rx0);
    __syncthreads();
  }
  const size_t slab_id = (size_t)prob * a.nsplit + zsplit;
  float* slab = a.slab + slab_id * a.Cnw * a.Kw;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = n0 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
    slab[(size_t)row * a.Kw + k0 + wk * 32 + l31] = acc[r];
  }
  if (want_bias) {  // 16 row lanes x 16 column quads -> 64 column sums of this row split
    f32x4* red = reinterpret_cast<f32x4*>(&sD[0][0]);  // (the loop ended on a barrier: the buffers are free)
    red[r0 * 16 + q] = bsum;
    __syncthreads();
    if (tid < 16) {
      f32x4 t = red[tid];
#pragma unroll
      for (int r = 1; r < 16; ++r) t += red[r * 16 + tid];
      *reinterpret_cast<f32x4*>(a.bslab + slab_id * a.Cnw + n0 + 4 * tid) = t;
    }
  }
}

// ---------------------------------------------------------------------------
// fp32 weight gradient with LDS-DMA staging (round 4).  wgrad_kernel<0> moves every chunk global -> registers (four stages
// of 16 VGPRs) -> ds_write_b128 -> LDS and spends 61 % of its wave time in s_waitcnt at 66 % MFMA-busy
// (profiles/r03_pmc_wgrad.txt).  Both operands are "row r0 = tid / 16, quad tid % 16" images of [32 rows][64 floats]: byte
// 16 * tid of the chunk buffer, i.e. lane-linear -- so buffer_load_dwordx4 ... lds can land them in LDS directly (per-lane
// SOURCE address = the gather; rows past the split and padding taps are pointed out of range and arrive as zeros), with no
// staging registers, no LDS stores and a ring of three chunk buffers: chunk c + 2 is requested while chunk c is multiplied,
// and the only wait in the loop is a counted vmcnt that leaves the newest chunk's four requests in flight.
// Same work decomposition, slab layout and bias rows as wgrad_kernel<0>: the reduction kernel is shared.
// ---------------------------------------------------------------------------
// WIDE (round 6): on gfx950 the f32 MFMA runs on the vector ALUs -- every VALU instruction of the gather's bookkeeping is matrix time
// lost (tools/probe/mfma_valu.hip) -- and the row state (pixel coordinates, two element offsets, three wrap tests: ~25 instructions)
// was advanced for TWO rows per thread and chunk.  When a tap's channels come in multiples of 64 (every layer of the SRGAN step that
// runs here) a thread owns ONE row of the chunk and both 32-float halves of it (same tap, same validity: the second request is the
// first + 32 elements), and the chunk image in LDS is [half][32 rows][32 floats] -- still lane-linear for the DMA, and a wave's MFMA
// operand is exactly one half.  Half the bookkeeping per MFMA.
// LIN (round 6, with WIDE): stride 1, output as large as the input, no PixelShuffle on dy, K and the dy columns in whole tiles of 64.
// Then both element offsets are LINEAR in the row m -- x: (m + dh Wi + dw) Ci + kc, dy: m Cd + col -- and all that is left of the row
// state is one bit per row: does tap (dh, dw) of pixel m fall inside the image.  The workgroup's tap is uniform (a k-tile of 64 lies
// inside one tap), so the bits of its row split are built ONCE per workgroup -- one ballot per 64 rows -- into an LDS table of one word
// per chunk, and a request is a bit test, two selects and three adds where it was ~40 instructions of coordinates and wrap tests
// (every one of them matrix time: the f32 MFMA shares the vector ALUs).  Same loads in the same order: bit-identical slabs.
constexpr int WG_MASKW = 768;  // chunks (incl. the two look-ahead requests) a LIN workgroup can index: 3 KB next to the 48 KB ring
template <bool WIDE, bool LIN = false>
__global__ __launch_bounds__(256) void wgrad_dma_kernel(const WMulti mp) {
  static_assert(!LIN || WIDE, "LIN is a form of WIDE");
  const WArgs& a = mp.a;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  int wi;
  {  // XCD-contiguous work order (see wgrad_kernel)
    const int W = (int)gridDim.x, b = (int)blockIdx.x, xcd = b & 7, slot = b >> 3, q = W >> 3, r = W & 7;
    wi = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int tiles_ = a.ktiles * (a.Cnw / 64);
  const int tile_id = wi % tiles_, rest_ = wi / tiles_;
  const int prob = srx_uniform(rest_ % a.nprob), zsplit = srx_uniform(rest_ / a.nprob);
  constexpr int NSLOT = 3, CHUNK = 32 * 64;  // floats per operand and chunk
  __shared__ __attribute__((aligned(16))) float sD[NSLOT][CHUNK];
  __shared__ __attribute__((aligned(16))) float sX[NSLOT][CHUNK];
  const int tid = threadIdx.x, lane = tid & 63, wave = srx_uniform(tid >> 6);
  const int ntile = srx_uniform(tile_id / a.ktiles), kt = srx_uniform(tile_id - ntile * a.ktiles);
  const int k0 = kt * 64, n0 = ntile * 64;
  constexpr int NP = WIDE ? 1 : 2;                  // row states per thread
  const int q = WIDE ? (tid & 7) : (tid & 15), r0 = WIDE ? (tid >> 3) : (tid >> 4);
  auto make_rsrc = [](const void* p, unsigned bytes) {
    const unsigned long long v = (unsigned long long)p;
    u32x4 r;
    r[0] = (unsigned)srx_uniform((int)(unsigned)v);
    r[1] = (unsigned)srx_uniform((int)((unsigned)(v >> 32) & 0xffffu));
    r[2] = bytes;
    r[3] = 0x00020000u;
    return r;
  };
  const u32x4 rx_ = make_rsrc(mp.x[prob], a.in_bytes), rd_ = make_rsrc(mp.dy[prob], a.dy_bytes);

  // this thread's fixed k (A gather) and fixed dy column (WIDE: of the first half; the second half is 32 further, same tap)
  const int k = k0 + 4 * q;
  const bool kvalid = k < a.K, kvalid1 = WIDE && k + 32 < a.K;
  int dh = 0, dw = 0, kc = 0;
  if (kvalid) {
    const int tap = k / a.Ck;
    kc = k - tap * a.Ck;
    const int th = tap / a.ntw, tw = tap - th * a.ntw;
    dh = a.dh0 + th;
    dw = a.dw0 + tw;
  }
  const int col = n0 + 4 * q;
  const bool cvalid = col < a.Cdv, cvalid1 = WIDE && col + 32 < a.Cdv;
  int sh_i = 0, sh_j = 0, sh_c = col;
  if (a.dy_shuffle) {  // (WIDE: a sub-pixel's channels come in multiples of 64 too, checked on the host: both halves in one sub-pixel)
    const int ij = col / a.dy_shuffle;
    sh_c = col - ij * a.dy_shuffle;
    sh_i = ij >> 1;
    sh_j = ij & 1;
  }
  const int mbeg = zsplit * a.rows_per_split;
  const int mend = min(a.M, mbeg + a.rows_per_split);

  __shared__ unsigned okmask[LIN ? WG_MASKW : 1];
  if constexpr (LIN) {  // bit r of word c: tap (dh, dw) of row mbeg + 32 c + r is inside the image (and the row inside the split)
    const int nrows = (((mend - mbeg + 31) / 32 + 2) * 32 + 63) & ~63;
    for (int base = wave * 64; base < nrows; base += 256) {
      const int m = mbeg + base + lane;
      int n, rem, mh, mw;
      srx_divmod(m, a.HmWm, a.inv_HmWm, n, rem);
      srx_divmod(rem, a.Wm, a.inv_Wm, mh, mw);
      const bool ok = m < mend && ((unsigned)(mh + dh) < (unsigned)a.Hi) && ((unsigned)(mw + dw) < (unsigned)a.Wi);
      const unsigned long long bits = __ballot(ok);
      if (lane == 0) { okmask[base >> 5] = (unsigned)bits; okmask[(base >> 5) + 1] = (unsigned)(bits >> 32); }
    }
    __syncthreads();
  }
  // row state of this thread's rows (r0 [+ 16 p] of the current chunk), advanced by one chunk per request
  int rm[NP], rmh[NP], rmw[NP];
  unsigned rox[NP], rod[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int m = mbeg + r0 + 16 * p;
    int n, rem, mh, mw;
    srx_divmod(m, a.HmWm, a.inv_HmWm, n, rem);
    srx_divmod(rem, a.Wm, a.inv_Wm, mh, mw);
    rm[p] = m; rmh[p] = mh; rmw[p] = mw;
    rox[p] = (unsigned)(((n * a.Hi + mh * a.in_stride + dh) * a.Wi + mw * a.in_stride + dw) * a.Ci + kc);
    rod[p] = (unsigned)(a.dy_shuffle ? ((n * 2 * a.Hm + 2 * mh + sh_i) * (2 * a.Wm) + 2 * mw + sh_j) * a.Cd + sh_c
                                     : m * a.Cd + col);
  }
  const unsigned ldsX = (unsigned)(size_t)&sX[0][0], ldsD = (unsigned)(size_t)&sD[0][0];
  auto dma = [&](const u32x4& rs, unsigned voff, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(dst), "s"(rs) : "memory");
  };
  // four requests per wave and chunk, always (so that the waits can be counted).  Two rows per thread: the thread's 16 bytes of row
  // r0 + 16 p land at byte 16 tid + 4096 p of the slot ([32 rows][64 floats]).  WIDE: the thread's 16 bytes of half hf of row r0 land
  // at byte 16 tid + 4096 hf ([half][32 rows][32 floats])
  constexpr unsigned OORL = 0xfffff000u;  // (an out-of-range offset that stays out of range with the second half's 128 bytes added)
  unsigned lin_x = 0, lin_d = 0, lin_w = 0;
  int lin_c = 0;
  if constexpr (LIN) {
    lin_x = 4u * (unsigned)((mbeg + r0 + dh * a.Wi + dw) * a.Ci + kc);  // (wraps for rows whose tap lies in front of the tensor: masked)
    lin_d = 4u * (unsigned)((mbeg + r0) * a.Cd + col);
    lin_w = okmask[0];
  }
  const unsigned lin_sx = 128u * (unsigned)a.Ci, lin_sd = 128u * (unsigned)a.Cd;  // 32 rows further, in bytes
  auto request = [&](int slot) {
    if constexpr (LIN) {
      const bool okx = (lin_w >> r0) & 1u;
      const bool okd = rm[0] < mend;
      const unsigned vx = okx ? lin_x : OORL, vd = okd ? lin_d : OORL;
      const unsigned dst = (unsigned)srx_uniform((int)((unsigned)(slot * CHUNK * 4) + (unsigned)(wave * 1024)));
      dma(rx_, vx, ldsX + dst);
      dma(rd_, vd, ldsD + dst);
      dma(rx_, vx + 128u, ldsX + dst + 4096u);
      dma(rd_, vd + 128u, ldsD + dst + 4096u);
      rm[0] += 32; lin_x += lin_sx; lin_d += lin_sd;
      lin_c += 1;
      lin_w = okmask[lin_c];  // (the next request's word: back long before it is tested)
      return;
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const bool valid = rm[p] < mend;
      const int ih = rmh[p] * a.in_stride + dh, iw = rmw[p] * a.in_stride + dw;
      const bool okx = valid && kvalid && ((unsigned)ih < (unsigned)a.Hi) && ((unsigned)iw < (unsigned)a.Wi);
      const unsigned dst = (unsigned)srx_uniform((int)((unsigned)(slot * CHUNK * 4) + (unsigned)(p * 4096 + wave * 1024)));
      dma(rx_, okx ? 4u * rox[p] : 0xffffffffu, ldsX + dst);
      dma(rd_, (valid && cvalid) ? 4u * rod[p] : 0xffffffffu, ldsD + dst);
      if constexpr (WIDE) {
        dma(rx_, (okx && kvalid1) ? 4u * rox[p] + 128u : 0xffffffffu, ldsX + dst + 4096u);
        dma(rd_, (valid && cvalid1) ? 4u * rod[p] + 128u : 0xffffffffu, ldsD + dst + 4096u);
      }
      rm[p] += 32; rmw[p] += a.s_c; rmh[p] += a.s_rm; rox[p] += (unsigned)a.dX0; rod[p] += (unsigned)a.dD0;
      const bool wc = rmw[p] >= a.Wm;
      rmw[p] -= wc ? a.Wm : 0; rmh[p] += wc ? 1 : 0;
      rox[p] += wc ? (unsigned)a.dX1 : 0u; rod[p] += wc ? (unsigned)a.dD1 : 0u;
      const bool wr = rmh[p] >= a.Hm;
      rmh[p] -= wr ? a.Hm : 0;
      rox[p] += wr ? (unsigned)a.dX2 : 0u;
    }
  };
  const bool want_bias = a.bslab != nullptr && kt == 0;  // workgroup-uniform
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f}, bsum1 = {0.f, 0.f, 0.f, 0.f};
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int h = lane >> 5, l31 = lane & 31;
  const int wn = wave >> 1, wk = wave & 1;
  // MFMA step s multiplies rows 2 s + h of the chunk: floats between two steps / between the two lane halves / to this wave's columns
  constexpr int SSTEP = WIDE ? 64 : 128, HSTEP = WIDE ? 32 : 64, WSTEP = WIDE ? 1024 : 32;
  auto compute = [&](int slot) {
    const float* cD = &sD[0][0] + slot * CHUNK + h * HSTEP + wn * WSTEP + l31;
    const float* cX = &sX[0][0] + slot * CHUNK + h * HSTEP + wk * WSTEP + l31;
    if (want_bias) {  // (fp32 values of this thread's two dy quads, as wgrad_kernel<0> adds them)
      if constexpr (WIDE) {
        const float* bd = &sD[0][0] + slot * CHUNK + r0 * 32 + q * 4;
        bsum += *reinterpret_cast<const f32x4*>(bd);
        bsum1 += *reinterpret_cast<const f32x4*>(bd + 1024);
      } else {
        const float* bd = &sD[0][0] + slot * CHUNK + r0 * 64 + q * 4;
        bsum += *reinterpret_cast<const f32x4*>(bd) + *reinterpret_cast<const f32x4*>(bd + 16 * 64);
      }
    }
    float dv[16], xv[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) { dv[s] = cD[s * SSTEP]; xv[s] = cX[s * SSTEP]; }
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(dv[s], xv[s], acc, 0, 0, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      if (i < 6) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    }
  };
  const int nchunks = (mend - mbeg + 31) / 32;
  request(0);
  request(1);  // (past the split's end: every lane out of range, zeros land -- never multiplied)
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // chunk 0 has landed, chunk 1 flies on
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  int slot = 0;
  for (int c = 0; c < nchunks; ++c) {
    int s2 = slot + 2; s2 = s2 >= NSLOT ? s2 - NSLOT : s2;
    request(s2);    // chunk c + 2 into the slot chunk c - 1 was read from (free since the barrier)
    compute(slot);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // chunk c + 1 has landed; the four requests of chunk c + 2 fly on
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    slot = slot + 1 == NSLOT ? 0 : slot + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the trailing requests write LDS: land before the slots are reused below)
  const size_t slab_id = (size_t)prob * a.nsplit + zsplit;
  float* slab = a.slab + slab_id * a.Cnw * a.Kw;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = n0 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
    slab[(size_t)row * a.Kw + k0 + wk * 32 + l31] = acc[r];
  }
  if (want_bias) {  // row lanes x 16 column quads -> 64 column sums of this row split
    __syncthreads();
    f32x4* red = reinterpret_cast<f32x4*>(&sD[0][0]);
    constexpr int NR = WIDE ? 32 : 16;  // row lanes
    if constexpr (WIDE) { red[r0 * 16 + q] = bsum; red[r0 * 16 + 8 + q] = bsum1; }
    else red[r0 * 16 + q] = bsum;
    __syncthreads();
    if (tid < 16) {
      f32x4 t = red[tid];
#pragma unroll
      for (int r = 1; r < NR; ++r) t += red[r * 16 + tid];
      *reinterpret_cast<f32x4*>(a.bslab + slab_id * a.Cnw + n0 + 4 * tid) = t;
    }
  }
}

// ---------------------------------------------------------------------------
// bf16 weight gradient of 3x3 / stride 1 / pad 1 convs with 64 output columns, on WHOLE IMAGE ROWS (round 3).
// wgrad_kernel above gathers one (tap, channel) k-tile per workgroup: every tap re-reads the same x pixels and every k-tile
// re-reads the dy tile -- 16 FLOP per byte pulled through the L2s, and with bf16 MFMAs (16x the fp32 rate) ESRGAN's 207
// dense-block problems per step (32 GB of reads) ran at the speed of that traffic: 205 TFLOP/s, unmoved by a 4x longer
// chunk per barrier or by half the LDS instructions (tools/experiments/README.md).  Here a workgroup owns 32 input channels
// of one problem and ALL NINE taps: per image row it loads one new x row (the window of three rows rolls through four LDS
// slots, zero columns left and right) and one dy row, rounds them to bf16 as they arrive, and multiplies
// dy[row]^T (64 columns) with the nine shifted views of the window -- 95 FLOP per byte.  Operands are pixel-major in LDS
// and an MFMA operand (eight consecutive PIXELS of one column) is two ds_read_b64_tr_b16; a tap is an address offset.
// Wave (wn, th): output-column half wn, taps 0..4 (th = 0) or 5..8 (th = 1): 5 / 4 accumulators of 32 columns x 32 channels.
// Rows outside the image are skipped tap-wise (the slots hold the neighbouring image's rows).  The slab layout is
// wgrad_kernel's ([n][tap * Ck + channel]), so wgrad_reduce_kernel, the pairs and the scales work unchanged.
// ---------------------------------------------------------------------------
template <int WPX>
__global__ __launch_bounds__(256) void wgrad_rows_bf16_kernel(const WMulti mp) {
  const WArgs& a = mp.a;
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
  constexpr int XPS = 64, XROW = (WPX + 2) * XPS;  // window: bytes per pixel (32 channels), per row (one zero pixel each side)
  constexpr int DPS = 192, DROW = WPX * DPS;       // dy: bytes per pixel (64 columns + pad: see wgrad_kernel), per row
  constexpr int XR = (WPX * 8 + 255) / 256, DR = (WPX * 16 + 255) / 256, KS = WPX / 16;
  __shared__ __attribute__((aligned(16))) unsigned char sX[5 * XROW];  // four window slots + a row of zeros (slot 4)
  __shared__ __attribute__((aligned(16))) unsigned char sD[2 * DROW > 4096 ? 2 * DROW : 4096];
  const int groups = a.Ck >> 5;
  int wi;
  {  // XCD-contiguous work order (see wgrad_kernel): the channel groups of one (problem, row split) share their dy rows
    const int G = (int)gridDim.x, b = (int)blockIdx.x, xcd = b & 7, slot = b >> 3, q = G >> 3, r = G & 7;
    wi = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int cg = srx_uniform(wi % groups), rest = wi / groups;
  const int prob = srx_uniform(rest % a.nprob), zsplit = srx_uniform(rest / a.nprob);
  const int tid = threadIdx.x, lane = tid & 63, wave = srx_uniform(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31, wn = wave & 1, th = wave >> 1;
  const int rbeg = zsplit * a.rows_per_split / WPX;
  const int rend = min(a.M, (zsplit + 1) * a.rows_per_split) / WPX;  // global image rows [rbeg, rend)
  const int totrows = a.N * a.Hi;
  const __amdgpu_buffer_rsrc_t rx_ = srx_rsrc(mp.x[prob], a.in_bytes), rd_ = srx_rsrc(mp.dy[prob], a.dy_bytes);
  const bool want_bias = a.bslab != nullptr && cg == 0;  // workgroup-uniform
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};

  auto xload = [&](int gr, f32x4 (&v)[XR]) {
#pragma unroll
    for (int u = 0; u < XR; ++u) {
      const int idx = u * 256 + tid, px = idx >> 3, quad = idx & 7;
      const bool ok = idx < WPX * 8 && (unsigned)gr < (unsigned)totrows;
      v[u] = srx_bload(rx_, ok ? 4u * (unsigned)((gr * WPX + px) * a.Ci + 32 * cg + 4 * quad) : 0xffffffffu, 0);
    }
  };
  auto xstore = [&](int gr, const f32x4 (&v)[XR]) {
#pragma unroll
    for (int u = 0; u < XR; ++u) {
      const int idx = u * 256 + tid, px = idx >> 3, quad = idx & 7;
      if (idx >= WPX * 8) continue;
      const bf16x4 pk = {(__bf16)v[u][0], (__bf16)v[u][1], (__bf16)v[u][2], (__bf16)v[u][3]};
      *reinterpret_cast<bf16x4*>(sX + (gr & 3) * XROW + (px + 1) * XPS + 8 * quad) = pk;
    }
  };
  auto dload = [&](int gr, f32x4 (&v)[DR]) {
#pragma unroll
    for (int u = 0; u < DR; ++u) {
      const int idx = u * 256 + tid, px = idx >> 4, quad = idx & 15;
      const bool ok = idx < WPX * 16 && (unsigned)gr < (unsigned)totrows;
      v[u] = srx_bload(rd_, ok ? 4u * (unsigned)((gr * WPX + px) * a.Cd + 4 * quad) : 0xffffffffu, 0);
    }
  };
  auto dstore = [&](int gr, const f32x4 (&v)[DR]) {
#pragma unroll
    for (int u = 0; u < DR; ++u) {
      const int idx = u * 256 + tid, px = idx >> 4, quad = idx & 15;
      if (idx >= WPX * 16) continue;
      if (want_bias && gr < rend) bsum += v[u];  // (fp32 values, whatever the product precision)
      const bf16x4 pk = {(__bf16)v[u][0], (__bf16)v[u][1], (__bf16)v[u][2], (__bf16)v[u][3]};
      *reinterpret_cast<bf16x4*>(sD + (gr & 1) * DROW + px * DPS + 8 * quad) = pk;
    }
  };

  f32x16 acc[5];
#pragma unroll
  for (int t = 0; t < 5; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const int li = lane & 15, colh = 16 * ((lane >> 4) & 1) + 4 * (li & 3), rq = li >> 2;

  // taps T0 .. T0 + NT - 1 of output row gr
  auto compute = [&](int gr, auto t0_c, auto nt_c) {
    constexpr int T0 = decltype(t0_c)::value, NT = decltype(nt_c)::value;
    const int ih = gr % a.Hi;
    const unsigned char* dbase = sD + (gr & 1) * DROW + (8 * h + rq) * DPS + 2 * (32 * wn + colh);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const s16x4 d0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(dbase + (16 * s) * DPS));
      const s16x4 d1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(dbase + (16 * s + 4) * DPS));
      const s16x8 fd = {d0[0], d0[1], d0[2], d0[3], d1[0], d1[1], d1[2], d1[3]};
#pragma unroll
      for (int ti = 0; ti < NT; ++ti) {
        constexpr int dummy = 0; (void)dummy;
        const int t = T0 + ti, ty = t / 3 - 1, tx = t % 3 - 1;
        // (wave-uniform) the row above / below lies outside the image: the tap multiplies the row of zeros.  Branch-free on
        // purpose -- with a `continue` here hipcc kept the five accumulators in different AGPRs on the two paths and moved
        // them at every join (16 v_accvgpr_mov per accumulator and row: 20 VALU instructions per MFMA, profiles/r04_pmc_esrgan.txt)
        const int slot = (unsigned)(ih + ty) < (unsigned)a.Hi ? ((gr + ty) & 3) : 4;
        const unsigned char* xb = sX + slot * XROW + (16 * s + 8 * h + rq + tx + 1) * XPS + 2 * colh;
        const s16x4 x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(xb));
        const s16x4 x1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(xb + 4 * XPS));
        const s16x8 fx = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
        acc[ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fd), __builtin_bit_cast(bf16x8, fx), acc[ti], 0, 0, 0);
      }
    }
  };

  // the row of zeros; zero columns left and right of every window slot
  if (tid < XROW / 16) *reinterpret_cast<f32x4*>(sX + 4 * XROW + 16 * tid) = f32x4{0.f, 0.f, 0.f, 0.f};
  if (tid < 32) {
    const int slot = tid >> 3, side = (tid >> 2) & 1, part = tid & 3;
    *reinterpret_cast<f32x4*>(sX + slot * XROW + (side ? (WPX + 1) * XPS : 0) + 16 * part) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  f32x4 xa[XR], xb2[XR], xc[XR], da[DR];
  xload(rbeg - 1, xa);
  xload(rbeg, xb2);
  xload(rbeg + 1, xc);
  dload(rbeg, da);
  xstore(rbeg - 1, xa);
  xstore(rbeg, xb2);
  xstore(rbeg + 1, xc);
  dstore(rbeg, da);
  xload(rbeg + 2, xa);
  dload(rbeg + 1, da);
  // one copy of the row loop per tap range (th is wave-uniform): with the choice INSIDE the loop the two paths kept the
  // accumulators in different AGPRs and every row paid 80 v_accvgpr_mov to bring them back together
  auto rows = [&](auto t0_c, auto nt_c) {
    for (int gr = rbeg; gr < rend; ++gr) {
      __syncthreads();  // rows gr - 1 .. gr + 1 and dy row gr are in LDS; the slots of x row gr - 2 and dy row gr - 1 are free
      xstore(gr + 2, xa);
      dstore(gr + 1, da);
      xload(gr + 3, xa);
      dload(gr + 2, da);
      compute(gr, t0_c, nt_c);
    }
  };
  if (th == 0) rows(std::integral_constant<int, 0>{}, std::integral_constant<int, 5>{});
  else rows(std::integral_constant<int, 5>{}, std::integral_constant<int, 4>{});
  const size_t slab_id = (size_t)prob * a.nsplit + zsplit;
  float* slab = a.slab + slab_id * a.Cnw * a.Kw;
#pragma unroll
  for (int ti = 0; ti < 5; ++ti) {
    const int t = 5 * th + ti;
    if (t >= 9) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      slab[(size_t)row * a.Kw + t * a.Ck + 32 * cg + l31] = acc[ti][r];
    }
  }
  if (want_bias) {  // 16 pixel lanes x 16 column quads -> 64 column sums of this row split
    __syncthreads();  // (every wave is done with the dy slots)
    f32x4* red = reinterpret_cast<f32x4*>(sD);
    red[tid] = bsum;
    __syncthreads();
    if (tid < 16) {
      f32x4 t = red[tid];
#pragma unroll
      for (int r = 1; r < 16; ++r) t += red[r * 16 + tid];
      *reinterpret_cast<f32x4*>(a.bslab + slab_id * a.Cnw + 4 * tid) = t;
    }
  }
}

// slab sums -> OIHW gradient.  blockIdx.y = output; its `nslab` slabs (row splits x segments) are consecutive.
__global__ void wgrad_reduce_kernel(const float* __restrict__ slab_all, int nslab, int Cnw, int Kw, int K, int Ck,
                                    int Cout, int Cin, int KH, int KW, int shuffle_cps, const WReduce outs,
                                    int accumulate, const float* __restrict__ bslab_all) {
  const int o = blockIdx.y;
  float* __restrict__ dw = outs.dw[o];
  float* __restrict__ db = outs.db[o];
  const float* __restrict__ slab = slab_all + (size_t)o * nslab * Cnw * Kw;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int rows_lo = outs.rows_lo;  // 0: one conv per output
  if (idx < Cout) {  // bias gradient: the slabs' column sums (the grid has >= Cout threads)
    float* __restrict__ dbp = (rows_lo && idx >= rows_lo) ? outs.db_hi[o] : db;
    if (dbp) {
      const float* __restrict__ bslab = bslab_all + (size_t)o * nslab * Cnw;
      float s = 0.f;
      for (int z = 0; z < nslab; ++z) s += bslab[(size_t)z * Cnw + idx];
      s *= outs.scale[o];
      int bi = (rows_lo && idx >= rows_lo) ? (int)idx - rows_lo : (int)idx;
      // PixelShuffle layers: the slab's columns are in packed (sub-pixel, channel) order, the bias in the conv's own
      if (shuffle_cps) { const int ij = (int)idx / shuffle_cps, cc = (int)idx - ij * shuffle_cps; bi = cc * 4 + ij; }
      dbp[bi] = accumulate ? dbp[bi] + s : s;
    }
  }
  if (idx >= (int64_t)Cout * K) return;
  const int srow = (int)(idx / K);  // row of the slab
  const int k = (int)(idx - (int64_t)srow * K);
  const int tap = k / Ck, ci = k - tap * Ck;
  int np = srow;                    // output channel of the conv the row belongs to
  if (rows_lo) {  // (no PixelShuffle on paired problems)
    if (np < rows_lo) { Cin = outs.cin_lo; } else { np -= rows_lo; dw = outs.dw_hi[o]; }
  }
  if (ci >= Cin) return;
  const float* sp = slab + (size_t)srow * Kw + k;
  const size_t zs = (size_t)Cnw * Kw;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int z = 0;
  for (; z + 3 < nslab; z += 4) {
    s0 += sp[(size_t)z * zs];
    s1 += sp[(size_t)(z + 1) * zs];
    s2 += sp[(size_t)(z + 2) * zs];
    s3 += sp[(size_t)(z + 3) * zs];
  }
  for (; z < nslab; ++z) s0 += sp[(size_t)z * zs];
  const float s = ((s0 + s1) + (s2 + s3)) * outs.scale[o];
  int co = np;
  if (shuffle_cps) { const int ij = np / shuffle_cps, cc = np - ij * shuffle_cps; co = cc * 4 + ij; }
  const int kh = tap / KW, kw = tap - kh * KW;
  float* op = dw + (((size_t)co * Cin + ci) * KH + kh) * KW + kw;
  *op = accumulate ? *op + s : s;
}

// The same reduction, one workgroup per slab ROW (round 5): the row's K sums are formed with coalesced reads ([k] contiguous in every
// slab), parked in LDS, and written out in the gradient's own order -- OIHW, (ci, kh, kw) contiguous per output channel --
// so the stores are coalesced too.  The kernel above writes 4-byte elements 36 bytes apart (k = (tap, ci) -> address
// (ci * 9 + tap)): 28 us per launch on ESRGAN's grouped gradients for 50 MB of traffic (1.8 TB/s).  Same sums in the same
// order: bit-identical results.  blockIdx.x = slab row, blockIdx.y = output; dynamic LDS: K floats.
__global__ __launch_bounds__(256) void wgrad_reduce_rows_kernel(const float* __restrict__ slab_all, int nslab, int Cnw, int Kw, int K, int Ck,
                                                                int Cout, int Cin, int KH, int KW, int shuffle_cps, const WReduce outs,
                                                                int accumulate, const float* __restrict__ bslab_all) {
  extern __shared__ __attribute__((aligned(16))) float rsum[];
  const int o = blockIdx.y, srow = blockIdx.x, tid = threadIdx.x;
  const float* __restrict__ slab = slab_all + (size_t)o * nslab * Cnw * Kw + (size_t)srow * Kw;
  const size_t zs = (size_t)Cnw * Kw;
  const float scale = outs.scale[o];
  const int rows_lo = outs.rows_lo;  // 0: one conv per output
  float* __restrict__ dw = outs.dw[o];
  float* __restrict__ dbp = outs.db[o];
  int np = srow;
  if (rows_lo) {  // (no PixelShuffle on paired problems)
    if (np < rows_lo) { Cin = outs.cin_lo; } else { np -= rows_lo; dw = outs.dw_hi[o]; dbp = outs.db_hi[o]; }
  }
  int co = np;
  if (shuffle_cps) { const int ij = np / shuffle_cps, cc = np - ij * shuffle_cps; co = cc * 4 + ij; }
  const int T = KH * KW, nE = Cin * T;
  float* __restrict__ orow = dw + (size_t)co * Cin * T;
  // round 6: a workgroup lives for three dependent round trips (slab rows, then -- accumulating -- the gradient's old values, then
  // the bias slabs one after the other in thread 0) and moves ~17 KB; the old values and the bias column are requested FIRST, next
  // to the slab rows.  Same sums in the same order.
  constexpr int PRE = 8;  // old values held in registers (Cin * T <= 2048: every layer of the two models)
  float oldv[PRE];
#pragma unroll
  for (int i = 0; i < PRE; ++i) {
    const int e = tid + 256 * i;
    oldv[i] = (accumulate && e < nE) ? orow[e] : 0.f;
  }
  // bias gradient of this row: the slabs' column sums, one slab per lane of the last wave (summed in slab order below)
  float bpart = 0.f;
  const bool bias_wave = dbp != nullptr && tid >= 192;
  if (bias_wave && nslab <= 64 && tid - 192 < nslab) bpart = bslab_all[((size_t)o * nslab + (tid - 192)) * Cnw + srow];
  // (four consecutive k per thread, 16-byte loads: K and the slab pitch Kw are multiples of 4; with one float per thread the pass was
  // bound by the few bytes it kept in flight, not by its stores: 27 us per launch either way)
  for (int k = 4 * tid; k < K; k += 1024) {
    const float* sp = slab + k;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    int z = 0;
    for (; z + 3 < nslab; z += 4) {
      s0 += *reinterpret_cast<const f32x4*>(sp + (size_t)z * zs);
      s1 += *reinterpret_cast<const f32x4*>(sp + (size_t)(z + 1) * zs);
      s2 += *reinterpret_cast<const f32x4*>(sp + (size_t)(z + 2) * zs);
      s3 += *reinterpret_cast<const f32x4*>(sp + (size_t)(z + 3) * zs);
    }
    for (; z < nslab; ++z) s0 += *reinterpret_cast<const f32x4*>(sp + (size_t)z * zs);
    *reinterpret_cast<f32x4*>(rsum + k) = ((s0 + s1) + (s2 + s3)) * scale;
  }
  if (bias_wave) {
    float s = 0.f;
    if (nslab <= 64) {
      for (int z = 0; z < nslab; ++z) s += __shfl(bpart, z, 64);
    } else {
      const float* __restrict__ bslab = bslab_all + (size_t)o * nslab * Cnw + srow;
      for (int z = 0; z < nslab; ++z) s += bslab[(size_t)z * Cnw];
    }
    if (tid == 192) {
      s *= scale;
      int bi = np;
      if (shuffle_cps) { const int ij = srow / shuffle_cps, cc = srow - ij * shuffle_cps; bi = cc * 4 + ij; }
      dbp[bi] = accumulate ? dbp[bi] + s : s;
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < PRE; ++i) {
    const int e = tid + 256 * i;
    if (e < nE) {
      const int ci = e / T, tap = e - ci * T;
      const float v = rsum[tap * Ck + ci];
      orow[e] = accumulate ? oldv[i] + v : v;
    }
  }
  for (int e = tid + 256 * PRE; e < nE; e += 256) {
    const int ci = e / T, tap = e - ci * T;
    const float v = rsum[tap * Ck + ci];
    orow[e] = accumulate ? orow[e] + v : v;
  }
}

// ---------------------------------------------------------------------------
// weight packing
// ---------------------------------------------------------------------------
__global__ void pack_fwd_kernel(const float* __restrict__ w, float* __restrict__ p, int Cout, int Cin, int KH, int KW,
                                int Ck, int K, int Kp, int Cnp, int shuffle_cps) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)Cnp * Kp) return;
  const int np = (int)(idx / Kp), k = (int)(idx - (int64_t)np * Kp);
  float v = 0.f;
  if (np < Cout && k < K) {
    const int tap = k / Ck, ci = k - tap * Ck;
    if (ci < Cin) {
      int co = np;
      if (shuffle_cps) { const int ij = np / shuffle_cps, cc = np - ij * shuffle_cps; co = cc * 4 + ij; }
      const int kh = tap / KW, kw = tap - kh * KW;
      v = w[(((size_t)co * Cin + ci) * KH + kh) * KW + kw];
    }
  }
  p[idx] = v;
}

// one stride-parity class of the data gradient: B[ci][(th,tw,c)] = W[co(c)][ci][kh(th)][kw(tw)]
__global__ void pack_bwd_kernel(const float* __restrict__ w, float* __restrict__ p, int Cout, int Cin, int KH, int KW,
                                int stride, int pad, int ph, int pw, int dminh, int dminw, int ntw, int Ck, int K,
                                int Kp, int Cnp, int shuffle_cps) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)Cnp * Kp) return;
  const int ci = (int)(idx / Kp), k = (int)(idx - (int64_t)ci * Kp);
  float v = 0.f;
  if (ci < Cin && k < K) {
    const int tap = k / Ck, c = k - tap * Ck;
    if (c < Cout) {
      const int th = tap / ntw, tw = tap - th * ntw;
      const int kh = ph + pad - stride * (dminh + th), kw = pw + pad - stride * (dminw + tw);
      int co = c;
      if (shuffle_cps) { const int ij = c / shuffle_cps, cc = c - ij * shuffle_cps; co = cc * 4 + ij; }
      v = w[(((size_t)co * Cin + ci) * KH + kh) * KW + kw];
    }
  }
  p[idx] = v;
}

// forward pack and every data-gradient class in ONE launch (blockIdx.y = segment): a repack after each
// optimiser step used to be 1 + stride^2 tiny launches per layer, ~130 per train step
struct PackSeg { float* dst; int rows, K, Kp, Ck, bwd, ph, pw, dminh, dminw, ntw; };
struct PackArgs {
  const float* w;
  int Cout, Cin, KH, KW, stride, pad, cps, nseg;
  PackSeg seg[17];
};
__global__ void pack_all_kernel(const PackArgs a) {
  const PackSeg sg = a.seg[blockIdx.y];
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)sg.rows * sg.Kp) return;
  const int row = (int)(idx / sg.Kp), k = (int)(idx - (int64_t)row * sg.Kp);
  float v = 0.f;
  if (k < sg.K) {
    const int tap = k / sg.Ck, c = k - tap * sg.Ck;
    if (!sg.bwd) {  // row = packed output channel n', c = input channel
      if (row < a.Cout && c < a.Cin) {
        int co = row;
        if (a.cps) { const int ij = row / a.cps, cc = row - ij * a.cps; co = cc * 4 + ij; }
        const int kh = tap / a.KW, kw = tap - kh * a.KW;
        v = a.w[(((size_t)co * a.Cin + c) * a.KH + kh) * a.KW + kw];
      }
    } else {        // row = input channel ci, c = (packed) output channel
      if (row < a.Cin && c < a.Cout) {
        const int th = tap / sg.ntw, tw = tap - th * sg.ntw;
        const int kh = sg.ph + a.pad - a.stride * (sg.dminh + th), kw = sg.pw + a.pad - a.stride * (sg.dminw + tw);
        int co = c;
        if (a.cps) { const int ij = c / a.cps, cc = c - ij * a.cps; co = cc * 4 + ij; }
        v = a.w[(((size_t)co * a.Cin + row) * a.KH + kh) * a.KW + kw];
      }
    }
  }
  sg.dst[idx] = v;
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
struct Geo {  // derived sizes of one conv
  int Ho, Wo, Ck, K, Kp, Cnp, cps;
};

int check_desc(const srx_conv2d_t* d) {
  SRX_REQUIRE(d, "conv2d: null descriptor");
  SRX_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0, "conv2d: non-positive size");
  SRX_REQUIRE(d->Cin_s >= d->Cin && d->Cin_s % 4 == 0, "conv2d: Cin_s must be a multiple of 4 and >= Cin");
  SRX_REQUIRE(d->KH > 0 && d->KW > 0 && d->stride > 0 && d->pad >= 0, "conv2d: bad kernel geometry");
  SRX_REQUIRE(d->shuffle == 0 || d->shuffle == 2, "conv2d: shuffle must be 0 or 2");
  SRX_REQUIRE(d->up == 0 || d->up == 1 || d->up == 2, "conv2d: up must be 0 / 1 (none) or 2 (nearest x2 in the gather)");
  if (d->up == 2) SRX_REQUIRE(d->stride == 1 && !d->shuffle, "conv2d: up = 2 needs stride 1 and no PixelShuffle");
  if (d->shuffle) {
    SRX_REQUIRE(d->Cout % 16 == 0, "conv2d: shuffle needs Cout multiple of 16");
    SRX_REQUIRE(d->Cout_s >= d->Cout / 4 && d->Cout_s % 4 == 0, "conv2d: bad Cout_s for shuffle");
    SRX_REQUIRE(d->stride == 1, "conv2d: shuffle needs stride 1");
  } else {
    SRX_REQUIRE(d->Cout_s >= d->Cout && d->Cout_s % 4 == 0, "conv2d: Cout_s must be a multiple of 4 and >= Cout");
  }
  SRX_REQUIRE(d->act == SRX_ACT_NONE || d->act == SRX_ACT_RELU || d->act == SRX_ACT_LRELU, "conv2d: bad act");
  SRX_REQUIRE(d->precision == 0 || d->precision == 1 || (d->precision == 2 && srx_thin_fwd_applicable(d)),
              "conv2d: precision must be 0 (fp32), 1 (bf16 products) or, for the forward of a 64 -> <= 4 channel layer, 2 (bf16 products there too)");
  const int uf = d->up == 2 ? 2 : 1;
  const int Ho = (uf * d->H + 2 * d->pad - d->KH) / d->stride + 1, Wo = (uf * d->W + 2 * d->pad - d->KW) / d->stride + 1;
  SRX_REQUIRE(Ho > 0 && Wo > 0, "conv2d: empty output");
  // (the generic forward / data-gradient kernel addresses by 64-bit tile bases and splits pixel indices exactly above 2^24;
  // the kernels that do not -- weight gradients, the row tile, the 3-channel input layers -- check `small_enough` themselves)
  SRX_REQUIRE((int64_t)d->N * uf * d->H * uf * d->W < (1LL << 31) - 1024 && (int64_t)d->N * Ho * Wo < (1LL << 31) - 1024,
              "conv2d: more than 2^31 pixels per call; tile the image");
  SRX_REQUIRE((int64_t)uf * d->W * d->Cin_s * 4 * (d->KH + 160) < (1LL << 32), "conv2d: image rows too long for 32-bit offsets inside a tile");
  SRX_REQUIRE(d->pad < 16000 && d->KH < 16000, "conv2d: kernel too large");
  return SRX_OK;
}

// below 2^24 pixels and 4 GiB of input: what the kernels with 32-bit tensor offsets / float index division can take
bool small_enough(const srx_conv2d_t* d) {
  const int uf = d->up == 2 ? 2 : 1;
  const int Ho = (uf * d->H + 2 * d->pad - d->KH) / d->stride + 1, Wo = (uf * d->W + 2 * d->pad - d->KW) / d->stride + 1;
  return (int64_t)d->N * uf * d->H * uf * d->W < (1 << 24) && (int64_t)d->N * Ho * Wo < (1 << 24) &&
         (int64_t)d->N * d->H * d->W * d->Cin_s < (1LL << 30) - 4;
}

int pad_rows(int c) { return c <= 32 ? 32 : (int)srx_roundup(c, 64); }  // rows of a packed operand: whole 64-column tiles (32 for the narrow tile)

Geo fwd_geo(const srx_conv2d_t* d) {
  Geo g;
  const int uf = d->up == 2 ? 2 : 1;  // extents the conv sees
  g.Ho = (uf * d->H + 2 * d->pad - d->KH) / d->stride + 1;
  g.Wo = (uf * d->W + 2 * d->pad - d->KW) / d->stride + 1;
  g.Ck = (int)srx_roundup(d->Cin, 4);  // k-space channels; the tensor's channel stride Cin_s may be larger
  g.K = d->KH * d->KW * g.Ck;
  g.Kp = (int)srx_roundup(g.K, BK);
  g.Cnp = pad_rows(d->Cout);
  g.cps = d->shuffle ? d->Cout / 4 : 0;
  return g;
}

struct BwdClass {  // one stride-parity class of the data gradient
  int ph, pw, nth, ntw, dminh, dminw, Hm, Wm, K, Kp;
  size_t woff;  // offset (floats) into the packed bwd buffer
};

// taps of class `par` along one axis: kernel size Kd; returns count, sets dmin
int class_taps(int par, int pad, int stride, int Kd, int& dmin) {
  const int kfirst = (par + pad) % stride;
  if (kfirst >= Kd) { dmin = 0; return 0; }
  const int cnt = (Kd - 1 - kfirst) / stride + 1;
  const int dmax = (par + pad - kfirst) / stride;
  dmin = dmax - (cnt - 1);
  return cnt;
}

// channels per tap of the data gradient's K axis: the layer's output channels, NOT the channel stride of
// the gradient tensor (a conv may write a 32-channel slice of a 192-channel dense-block buffer)
int bwd_ck(const srx_conv2d_t* d) { return d->shuffle ? d->Cout : (int)srx_roundup(d->Cout, 4); }

int bwd_classes(const srx_conv2d_t* d, BwdClass* cls, size_t& total_floats) {
  const int Ck = bwd_ck(d);
  const int Cnp = pad_rows(d->Cin);
  int n = 0;
  total_floats = 0;
  for (int ph = 0; ph < d->stride; ++ph)
    for (int pw = 0; pw < d->stride; ++pw) {
      BwdClass c;
      c.ph = ph; c.pw = pw;
      c.nth = class_taps(ph, d->pad, d->stride, d->KH, c.dminh);
      c.ntw = class_taps(pw, d->pad, d->stride, d->KW, c.dminw);
      c.Hm = (d->H - ph + d->stride - 1) / d->stride;
      c.Wm = (d->W - pw + d->stride - 1) / d->stride;
      c.K = c.nth * c.ntw * Ck;
      c.Kp = (int)srx_roundup(c.K > 0 ? c.K : 1, BK);
      c.woff = total_floats;
      total_floats += (size_t)Cnp * c.Kp;
      cls[n++] = c;
    }
  return n;
}

// Launch plan.  The chip finishes when its busiest CU does, and a CU's matrix pipes are shared by
// whatever workgroups sit on it, so time ~ (tiles on the busiest CU) x (tile work).  With T equal
// tiles on P CUs that is ceil(T/P): 288 tiles cost as much as 512.  The plan therefore runs the
// first floor(T/P)*P tiles whole and cuts the remaining r = T mod P tiles `split` ways along K so
// that the last round is r*split <= ~P small workgroups (a data-parallel + split-K-tail hybrid);
// only those r tiles take the partial-sum round trip through HBM.  Larger tiles are preferred
// (fewer L2->LDS bytes per MFMA: the 64x64 tile moves 16 B/clk/workgroup and stalls on L1/L2)
// unless they leave the chip under-filled.
struct Plan { int BM, BN, mtiles, ntiles, tiles, full, tail, split, kc_per_split, ks; float cost; };

int device_cus() { return srx_plan_cus(); }  // the device's CUs less the reserved ones (api.cpp)

Plan make_plan(int M, int Cnp, int kchunks, bool can_split, bool bf16 = false, bool big = false) {
  const int P = device_cus();
  constexpr int NC = 8;
  // (64, 32): narrow layers (the 32-channel growth convs of ESRGAN's dense blocks) on twice as many tiles, the
  // k-chunks of each tile split over 2 (fp32) / 4 (bf16) wave groups inside the workgroup -- no fix-up pass
  // (256, 128), bf16 only (round 4): 64 x 64 per wave, one LDS fragment read per MFMA instead of 1.5 -- the bf16 loop is bound
  // by LDS traffic: 256 -> 256 at 32 x 32 x 32 pixels 560 -> 650 TFLOP/s, 128 -> 256 at 64^2 484 -> 546; with fp32 products the
  // same tile LOSES a third (72 vs 110 TFLOP/s: 212 registers, and nothing there was LDS-bound) and is never planned
  const int cand[NC][2] = {{144, 128}, {144, 64}, {128, 128}, {128, 64}, {64, 64}, {128, 32}, {64, 32}, {256, 128}};
  const float eff[NC] = {0.91f, 0.82f, 0.95f, 0.85f, 0.60f, 0.50f, 0.50f, 1.08f};  // measured MFMA efficiency in steady state
  Plan best{};
  best.cost = 1e30f;
  for (int i = 0; i < NC; ++i) {
    const int bm = cand[i][0], bn = cand[i][1];
    if (bf16 && bm == 144) continue;
    if (!bf16 && bm == 256) continue;
    if (big && bm != 128) continue;  // whole-frame calls (BIG instantiations): 128 x {128, 64, 32}
    if (Cnp == 32) { if (bn != 32) continue; }
    else if (bn == 32 || Cnp % bn != 0) continue;
    Plan p{};
    p.BM = bm; p.BN = bn;
    p.mtiles = (int)srx_cdiv(M, bm);
    p.ntiles = Cnp / bn;
    p.tiles = p.mtiles * p.ntiles;
    // microseconds for one whole tile on an otherwise idle CU (4 SIMDs x 64 FLOP/clk at ~2.1 GHz).  bf16 products:
    // the loop is bound by the fp32 operand stream, not by the matrix pipe, and runs ~3x the fp32 rate (measured
    // 240-360 vs ~105 TFLOP/s on the VGG layers) -- the fixed cost of a K-split fix-up pass weighs 3x more
    const float t_tile = 2.0f * bm * bn * (float)kchunks * BK / (4 * 64 * 2.1e3f) / eff[i] / (bf16 ? 3.0f : 1.0f);
    const int rounds = p.tiles / P, r = p.tiles % P;
    p.full = rounds * P; p.tail = r; p.split = 1; p.kc_per_split = kchunks;
    float tail_cost = r ? t_tile : 0.f;
    if (r && can_split && !big) {
      const int smax = kchunks / 4 < 16 ? kchunks / 4 : 16;
      for (int s = 2; s <= smax; ++s) {
        const int kcs = (int)srx_cdiv(kchunks, s);
        const int s_eff = (int)srx_cdiv(kchunks, kcs);
        const float busiest = (float)srx_cdiv((int64_t)r * s_eff, P) * kcs / (float)kchunks;
        // fix-up: (s+1) passes over the tail tiles at ~3 TB/s plus one more kernel boundary
        const float fix = 4.0f + (float)r * (s_eff + 1) * bm * bn * 4.0f / 3.0e6f + (float)s_eff * bm * 64.0f / 40.0e3f;
        const float c = busiest * t_tile + fix;
        if (c < tail_cost) { tail_cost = c; p.split = s_eff; p.kc_per_split = kcs; }
      }
    }
    p.cost = rounds * t_tile + tail_cost;
    if (p.cost < best.cost) best = p;
  }
  // developer override for tile experiments: SRX_FORCE_PLAN="BM,BN,split,ks" (split: K-split of ALL tiles)
  if (srx_dev().force_plan && !big) {
    const int bm = srx_dev().plan[0], bn = srx_dev().plan[1], sp = srx_dev().plan[2], ks = srx_dev().plan[3];
    if (bn > 0 && Cnp % bn == 0) {
      Plan p{};
      p.BM = bm; p.BN = bn; p.mtiles = (int)srx_cdiv(M, bm); p.ntiles = Cnp / bn; p.tiles = p.mtiles * p.ntiles;
      if (sp > 1 && can_split) { p.full = 0; p.tail = p.tiles; p.kc_per_split = (int)srx_cdiv(kchunks, sp); p.split = (int)srx_cdiv(kchunks, p.kc_per_split); }
      else { p.full = p.tiles; p.tail = 0; p.split = 1; p.kc_per_split = kchunks; }
      p.ks = (bm == 64 && bn == 64 && ks == 2) ? 2 : 1;
      if (bm == 64 && bn == 32) p.ks = bf16 ? 4 : 2;
      return p;
    }
  }
  // fewer workgroups than CUs and a 4-wave tile: split its k-chunks over two wave groups (KS = 2)
  best.ks = (best.BM == 64 && best.BN == 64 && best.full + best.tail * best.split <= P && best.kc_per_split >= 4) ? 2 : 1;
  if (best.BM == 64 && best.BN == 32) best.ks = bf16 ? 4 : 2;
  if (big) best.ks = 1;
  return best;
}

size_t plan_ws_floats(const Plan& p) { return p.split > 1 ? (size_t)p.tail * p.split * p.BM * p.BN : 0; }

template <int BM, int BN, int WM, int WN, int KS, int XR, int PR = 0, int BIG = 0>
int launch_gconv(const GArgs& a, const Plan& p, hipStream_t st) {
  const int ktab_chunks = p.full > 0 || p.split == 1 ? a.kchunks : p.kc_per_split;
  const size_t lds = (size_t)(KS * 3 * (BM + XR + BN) * BK) * (PR == 1 ? 2 : 4) + (size_t)ktab_chunks * 8 * sizeof(int2);
  if (lds > 160 * 1024) SRX_FAIL(SRX_E_UNSUPPORTED, "conv2d: K range needs %zu bytes of LDS", lds);
  static std::once_flag once;
  std::call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gconv_kernel<BM, BN, WM, WN, KS, XR, PR, BIG>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  dim3 grid(p.full + p.tail * p.split);
  char nm[112];  // kernel + GEMM shape: the roofline leg of bench.py groups launches by both
  if (srx_prof_on())
    snprintf(nm, sizeof(nm), "gconv_kernel<%d, %d, %d, %d, %d, %d, %d%s> MxNxK=%dx%dx%d", BM, BN, WM, WN, KS, XR, PR, BIG ? ", big" : "",
             a.M, a.Cn, PR == 2 ? 2 * a.K : a.K);  // (PR = 2: K counts floats = pairs of bf16 values)
  SRX_LAUNCH_PROF(nm, 2.0 * a.M * a.Cn * (PR == 2 ? 2 * a.K : a.K), (gconv_kernel<BM, BN, WM, WN, KS, XR, PR, BIG>), grid,
                  dim3((BM / WM) * (BN / WN) * 64 * KS), lds, st, a);
  SRX_CHECK_LAUNCH("gconv_kernel");
  if (p.split > 1) {
    hipLaunchKernelGGL((tail_fixup_kernel<BM + XR, BN>), dim3(p.tail * (BN / 16)), dim3(256), 0, st, a);
    SRX_CHECK_LAUNCH("tail_fixup_kernel");
  }
  return SRX_OK;
}

template <int BM, int BN, int WM, int WN, int XR, int PR = 0, int KS = 1>
int launch_gconv_multi(const GMulti& m, size_t lds, hipStream_t st) {
  static std::once_flag once;
  std::call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gconv_multi_kernel<BM, BN, WM, WN, XR, PR, KS>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  char nm[64];
  double fl = 0.0;
  if (srx_prof_on()) {
    snprintf(nm, sizeof(nm), "gconv_multi_kernel<%d, %d, %d, %d, %d, %d, %d>", BM, BN, WM, WN, XR, PR, KS);
    for (int i = 0; i < m.n; ++i) fl += 2.0 * m.g[i].M * m.g[i].Cn * m.g[i].K;
  }
  SRX_LAUNCH_PROF(nm, fl, (gconv_multi_kernel<BM, BN, WM, WN, XR, PR, KS>), dim3(m.first[m.n]), dim3((BM / WM) * (BN / WN) * 64 * KS),
                  lds, st, m);
  SRX_CHECK_LAUNCH("gconv_multi_kernel");
  return SRX_OK;
}

// Tile for a multi-problem launch.  The stride-parity classes of a data gradient have the same M but
// 1x..4x different K, and their strided outputs cannot take the split-K fix-up path, so balance comes
// from the tile size alone: every candidate's workgroups are list-scheduled heavy-first (the order
// run_gconv_multi launches them in) over the CUs and the shortest makespan wins.
float multi_tile(const GMulti& m, int Cnp, int& BM, int& BN, int& KS, bool bf16 = false) {
  const int P = device_cus();
  constexpr int NC = 7;
  // (the last candidate: 64 x 64 with the k-chunks of a tile dealt to two wave groups -- half the chain, one workgroup per CU)
  const int cand[NC][2] = {{144, 128}, {144, 64}, {128, 128}, {128, 64}, {64, 64}, {128, 32}, {64, 64}};
  const float eff[NC] = {0.91f, 0.82f, 0.95f, 0.85f, 0.60f, 0.50f, 0.80f};
  float best = 1e30f;
  BM = 128; BN = Cnp == 32 ? 32 : 64; KS = 1;
  if (srx_dev().force_plan && srx_dev().plan[1] > 0 && Cnp % srx_dev().plan[1] == 0) {  // developer override (SRX_FORCE_PLAN)
    BM = srx_dev().plan[0]; BN = srx_dev().plan[1]; KS = (BM == 64 && BN == 64 && srx_dev().plan[3] == 2) ? 2 : 1;
    return 0.f;
  }
  std::vector<float> heap;
  for (int i = 0; i < NC; ++i) {
    const int bm = cand[i][0], bn = cand[i][1];
    if (bf16 && bm == 144) continue;  // (the 16-row extension is fp32 only)
    if (i == 6 && (srx_dev().s2_mode & 2)) continue;
    if (Cnp == 32) { if (bn != 32) continue; }
    else if (bn == 32 || Cnp % bn != 0) continue;
    heap.assign(P, 0.f);  // min-heap of CU finish times
    auto later = [](float x, float y) { return x > y; };
    float makespan = 0.f;
    for (int pass = 4; pass >= 1; --pass)  // heavy-first: K of a class is 1..4 tap groups; order by chunk count
      for (int c = 0; c < m.n; ++c) {
        const int kch = m.g[c].Kp / BK;
        int rank = 1;
        for (int o = 0; o < m.n; ++o) rank += (m.g[o].Kp < m.g[c].Kp);
        if (rank != pass) continue;
        // (see make_plan; the two-group tile: half the chain at 0.8, and a fixed cost nothing overlaps -- set-up and fold of a
        // 512-thread workgroup that has the CU to itself: measured ~6.5 us on the discriminator's layers)
        const float t = 2.0f * bm * bn * (float)kch * BK / (4 * 64 * 2.1e3f) / (bf16 ? 3.0f : 1.0f);
        const float cost = i == 6 ? 6.5f + 0.5f * t / 0.8f : 1.0f + t / eff[i];
        const int tiles = (int)srx_cdiv(m.g[c].M, bm) * (int)srx_cdiv(m.g[c].Cn, bn);
        for (int t = 0; t < tiles; ++t) {
          std::pop_heap(heap.begin(), heap.end(), later);
          heap.back() += cost;
          if (heap.back() > makespan) makespan = heap.back();
          std::push_heap(heap.begin(), heap.end(), later);
        }
      }
    if (makespan < best) { best = makespan; BM = bm; BN = bn; KS = i == 6 ? 2 : 1; }
  }
  return best;
}

// all problems use tile (BM, BN); no K split.  Problems are launched heaviest (largest K) first.
int run_gconv_multi(GMulti& m, int BM, int BN, int KS, hipStream_t st, int precision = 0) {
  std::stable_sort(m.g, m.g + m.n, [](const GArgs& x, const GArgs& y) { return x.Kp > y.Kp; });
  int maxk = 0;
  m.first[0] = 0;
  for (int i = 0; i < m.n; ++i) {
    GArgs& a = m.g[i];
    a.kchunks = a.Kp / BK;
    a.mtiles = (int)srx_cdiv(a.M, BM);
    a.kc_per_split = a.kchunks;
    a.tail_split = 1;
    a.ws = nullptr;
    const int tiles = a.mtiles * (int)srx_cdiv(a.Cn, BN);
    a.full_tiles = tiles;
    m.first[i + 1] = m.first[i] + tiles;
    if (a.kchunks > maxk) maxk = a.kchunks;
  }
  const size_t lds = (size_t)(KS * 3 * (BM + BN) * BK) * (precision ? 2 : 4) + (size_t)maxk * 8 * sizeof(int2);
  if (lds > 160 * 1024) SRX_FAIL(SRX_E_UNSUPPORTED, "conv2d: K range needs %zu bytes of LDS", lds);
  if (KS == 2) {
    if (BM != 64 || BN != 64) SRX_FAIL(SRX_E_UNSUPPORTED, "conv2d: internal: two wave groups on a tile other than 64 x 64");
    if (precision) return launch_gconv_multi<64, 64, 32, 32, 0, 1, 2>(m, lds, st);
    return launch_gconv_multi<64, 64, 32, 32, 0, 0, 2>(m, lds, st);
  }
  if (precision) {  // bf16 products (round 4: the stride-parity classes of a strided data gradient too)
    if (BM == 128 && BN == 128) return launch_gconv_multi<128, 128, 64, 32, 0, 1>(m, lds, st);
    if (BM == 128 && BN == 64) return launch_gconv_multi<128, 64, 32, 32, 0, 1>(m, lds, st);
    if (BM == 64 && BN == 64) return launch_gconv_multi<64, 64, 32, 32, 0, 1>(m, lds, st);
    return launch_gconv_multi<128, 32, 32, 32, 0, 1>(m, lds, st);
  }
  if (BM == 144 && BN == 128) return launch_gconv_multi<128, 128, 64, 32, 16>(m, lds, st);
  if (BM == 144 && BN == 64) return launch_gconv_multi<128, 64, 32, 32, 16>(m, lds, st);
  if (BM == 128 && BN == 128) return launch_gconv_multi<128, 128, 64, 32, 0>(m, lds, st);
  if (BM == 128 && BN == 64) return launch_gconv_multi<128, 64, 32, 32, 0>(m, lds, st);
  if (BM == 64 && BN == 64) return launch_gconv_multi<64, 64, 32, 32, 0>(m, lds, st);
  return launch_gconv_multi<128, 32, 32, 32, 0>(m, lds, st);
}

// ---- the fused-class strided data gradient (gconv_s2f_kernel) ----
// Eligible: 2..4 non-empty classes on ONE M grid (extents multiples of the stride), whole chunks per tap, plain gather.
bool s2f_eligible(const srx_conv2d_t* d, const BwdClass* cls, int nc) {
  if ((srx_dev().s2_mode & 1) || nc < 2 || nc > 4 || d->shuffle || d->up || d->precision > 1) return false;
  if (bwd_ck(d) % BK != 0 || pad_rows(d->Cin) % 64 != 0) return false;
  for (int i = 0; i < nc; ++i)
    if (cls[i].K == 0 || cls[i].Hm != cls[0].Hm || cls[i].Wm != cls[0].Wm || cls[i].Hm <= 0 || cls[i].Wm <= 0) return false;
  return true;
}

// tile of the fused launch and its estimated time (the model of make_plan / multi_tile: one workgroup slot per CU); a
// workgroup pays one set-up, one pipeline fill and an epilogue per class
float s2f_tile(const srx_conv2d_t* d, const BwdClass* cls, int nc, int& BM, int& BN) {
  const int P = device_cus();
  const bool bf16 = d->precision != 0;
  const int Cnp = pad_rows(d->Cin);
  const int M = d->N * cls[0].Hm * cls[0].Wm;
  int chunks = 0;
  for (int i = 0; i < nc; ++i) chunks += cls[i].K / BK;
  constexpr int NC = 5;
  const int cand[NC][2] = {{144, 128}, {144, 64}, {128, 128}, {128, 64}, {64, 64}};
  const float eff[NC] = {0.91f, 0.82f, 0.95f, 0.85f, 0.60f};
  float best = 1e30f;
  BM = 0; BN = 0;
  for (int i = 0; i < NC; ++i) {
    const int bm = cand[i][0], bn = cand[i][1];
    if ((bf16 && bm == 144) || Cnp % bn != 0) continue;
    const float t = 2.0f * bm * bn * (float)chunks * BK / (4 * 64 * 2.1e3f) / eff[i] / (bf16 ? 3.0f : 1.0f);
    const int64_t tiles = srx_cdiv(M, bm) * (Cnp / bn);
    const float cost = (float)srx_cdiv(tiles, P) * (1.5f + 0.7f * nc * (bm * bn / 8192.0f) + t);
    if (cost < best) { best = cost; BM = bm; BN = bn; }
  }
  return best;
}

template <int BM, int BN, int WM, int WN, int XR, int PR>
int launch_gconv_s2f(const GFused& f, dim3 grid, size_t lds, double fl, hipStream_t st) {
  static std::once_flag once;
  std::call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gconv_s2f_kernel<BM, BN, WM, WN, XR, PR>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  char nm[64];
  if (srx_prof_on()) snprintf(nm, sizeof(nm), "gconv_s2f_kernel<%d, %d, %d, %d, %d, %d>", BM, BN, WM, WN, XR, PR);
  SRX_LAUNCH_PROF(nm, fl, (gconv_s2f_kernel<BM, BN, WM, WN, XR, PR>), grid, dim3((BM / WM) * (BN / WN) * 64), lds, st, f);
  SRX_CHECK_LAUNCH("gconv_s2f_kernel");
  return SRX_OK;
}

// `shared`: the launch arguments of any one class (gconv_multi's), `wpk` / `wfloats`: the whole packed data-gradient buffer
int run_gconv_s2f(const srx_conv2d_t* d, const GArgs& shared, const BwdClass* cls, int nc, const float* wpk, size_t wfloats,
                  int BM, int BN, hipStream_t st) {
  GFused f{};
  f.g = shared;
  f.g.w = wpk;
  if (wfloats * sizeof(float) >= 0xfffffff0ull) SRX_FAIL(SRX_E_UNSUPPORTED, "conv2d_bwd_data: packed weights above 4 GiB");
  f.g.w_bytes = (unsigned)(wfloats * sizeof(float));
  f.g.mtiles = (int)srx_cdiv(shared.M, BM);
  f.nclass = nc;
  f.cpt = bwd_ck(d) / BK;
  int pos = 0;
  double fl = 0.0;
  for (int i = 0; i < nc; ++i) {
    S2Class& k = f.c[i];
    k.nth = cls[i].nth; k.ntw = cls[i].ntw; k.dh0 = cls[i].dminh; k.dw0 = cls[i].dminw;
    k.chunks = cls[i].K / BK; k.pstart = pos;
    k.oh_off = cls[i].ph; k.ow_off = cls[i].pw;
    k.wbase = (unsigned)(cls[i].woff * sizeof(float)); k.kpb = (unsigned)(cls[i].Kp * sizeof(float));
    pos += (int)srx_roundup(k.chunks, 3);
    fl += 2.0 * shared.M * shared.Cn * cls[i].K;
  }
  f.ptotal = pos;
  const int Cnp = pad_rows(d->Cin);
  const bool bf16 = d->precision != 0;
  const size_t lds = (size_t)(3 * (BM + BN) * BK) * (bf16 ? 2 : 4) + (size_t)(pos + 3) * sizeof(int4) + (size_t)srx_roundup(BM, 4) * 4;
  if (lds > 160 * 1024) SRX_FAIL(SRX_E_UNSUPPORTED, "conv2d_bwd_data: K range needs %zu bytes of LDS", lds);
  const dim3 grid((unsigned)(f.g.mtiles * (Cnp / BN)));
  if (bf16) {
    if (BM == 128 && BN == 128) return launch_gconv_s2f<128, 128, 64, 32, 0, 1>(f, grid, lds, fl, st);
    if (BM == 128 && BN == 64) return launch_gconv_s2f<128, 64, 32, 32, 0, 1>(f, grid, lds, fl, st);
    if (BM == 64 && BN == 64) return launch_gconv_s2f<64, 64, 32, 32, 0, 1>(f, grid, lds, fl, st);
  } else {
    if (BM == 144 && BN == 128) return launch_gconv_s2f<128, 128, 64, 32, 16, 0>(f, grid, lds, fl, st);
    if (BM == 144 && BN == 64) return launch_gconv_s2f<128, 64, 32, 32, 16, 0>(f, grid, lds, fl, st);
    if (BM == 128 && BN == 128) return launch_gconv_s2f<128, 128, 64, 32, 0, 0>(f, grid, lds, fl, st);
    if (BM == 128 && BN == 64) return launch_gconv_s2f<128, 64, 32, 32, 0, 0>(f, grid, lds, fl, st);
    if (BM == 64 && BN == 64) return launch_gconv_s2f<64, 64, 32, 32, 0, 0>(f, grid, lds, fl, st);
  }
  SRX_FAIL(SRX_E_UNSUPPORTED, "conv2d_bwd_data: internal: no fused-class instantiation for tile %d x %d", BM, BN);
}

int run_gconv(GArgs& a, const Plan& p, float* ws, size_t ws_floats, hipStream_t st, int precision = 0) {
  a.kchunks = a.Kp / BK;
  const size_t need = plan_ws_floats(p);
  if (need && (!ws || need > ws_floats))
    SRX_FAIL(SRX_E_WORKSPACE, "conv2d: split-K workspace %zu < %zu floats", ws_floats, need);
  if (p.split > 1 && !a.linear_out) SRX_FAIL(SRX_E_UNSUPPORTED, "conv2d: internal: split plan on a strided output");
  a.mtiles = p.mtiles;
  a.kc_per_split = p.kc_per_split;
  a.full_tiles = p.full;
  a.tail_split = p.split;
  a.ws = ws;
  if (a.big) {  // whole-frame calls: three tiles, no K split (plans made with `big` ask for nothing else)
    if (p.split > 1) SRX_FAIL(SRX_E_UNSUPPORTED, "conv2d: internal: split plan on a call above 2^24 pixels");
    if (precision) {
      if (p.BM == 128 && p.BN == 128) return launch_gconv<128, 128, 64, 32, 1, 0, 1, 1>(a, p, st);
      if (p.BM == 128 && p.BN == 64) return launch_gconv<128, 64, 32, 32, 1, 0, 1, 1>(a, p, st);
      if (p.BM == 128 && p.BN == 32) return launch_gconv<128, 32, 32, 32, 1, 0, 1, 1>(a, p, st);
    } else {
      if (p.BM == 128 && p.BN == 128) return launch_gconv<128, 128, 64, 32, 1, 0, 0, 1>(a, p, st);
      if (p.BM == 128 && p.BN == 64) return launch_gconv<128, 64, 32, 32, 1, 0, 0, 1>(a, p, st);
      if (p.BM == 128 && p.BN == 32) return launch_gconv<128, 32, 32, 32, 1, 0, 0, 1>(a, p, st);
    }
    SRX_FAIL(SRX_E_UNSUPPORTED, "conv2d: no whole-frame kernel for tile %dx%d", p.BM, p.BN);
  }
  if (precision == 2) {  // bf16 storage (round 5): the tensors hold bf16, a chunk is 64 k-values
    if (p.BM == 256 && p.BN == 128) return launch_gconv<256, 128, 64, 64, 1, 0, 2>(a, p, st);
    if (p.BM == 128 && p.BN == 128) return launch_gconv<128, 128, 64, 32, 1, 0, 2>(a, p, st);
    if (p.BM == 128 && p.BN == 64) return launch_gconv<128, 64, 32, 32, 1, 0, 2>(a, p, st);
    if (p.BM == 64 && p.BN == 64)
      return p.ks == 2 ? launch_gconv<64, 64, 32, 32, 2, 0, 2>(a, p, st) : launch_gconv<64, 64, 32, 32, 1, 0, 2>(a, p, st);
    SRX_FAIL(SRX_E_UNSUPPORTED, "conv2d: no bf16-storage kernel for tile %dx%d", p.BM, p.BN);
  }
  if (precision) {  // bf16 products (plans made with `bf16 = true` never ask for the 144-row tiles)
    if (p.BM == 256 && p.BN == 128) return launch_gconv<256, 128, 64, 64, 1, 0, 1>(a, p, st);
    if (p.BM == 128 && p.BN == 128) return launch_gconv<128, 128, 64, 32, 1, 0, 1>(a, p, st);
    if (p.BM == 128 && p.BN == 64) return launch_gconv<128, 64, 32, 32, 1, 0, 1>(a, p, st);
    if (p.BM == 64 && p.BN == 64)
      return p.ks == 2 ? launch_gconv<64, 64, 32, 32, 2, 0, 1>(a, p, st) : launch_gconv<64, 64, 32, 32, 1, 0, 1>(a, p, st);
    if (p.BM == 128 && p.BN == 32) return launch_gconv<128, 32, 32, 32, 1, 0, 1>(a, p, st);
    if (p.BM == 64 && p.BN == 32) return launch_gconv<64, 32, 32, 32, 4, 0, 1>(a, p, st);
    SRX_FAIL(SRX_E_UNSUPPORTED, "conv2d: no bf16 kernel for tile %dx%d", p.BM, p.BN);
  }
  if (p.BM == 144 && p.BN == 128) return launch_gconv<128, 128, 64, 32, 1, 16>(a, p, st);
  if (p.BM == 144 && p.BN == 64) return launch_gconv<128, 64, 32, 32, 1, 16>(a, p, st);
  if (p.BM == 128 && p.BN == 128) return launch_gconv<128, 128, 64, 32, 1, 0>(a, p, st);
  if (p.BM == 128 && p.BN == 64) return launch_gconv<128, 64, 32, 32, 1, 0>(a, p, st);
  if (p.BM == 64 && p.BN == 64)
    return p.ks == 2 ? launch_gconv<64, 64, 32, 32, 2, 0>(a, p, st) : launch_gconv<64, 64, 32, 32, 1, 0>(a, p, st);
  if (p.BM == 64 && p.BN == 32) return launch_gconv<64, 32, 32, 32, 2, 0>(a, p, st);
  return launch_gconv<128, 32, 32, 32, 1, 0>(a, p, st);
}

void set_mgrid(GArgs& a, int N, int Hm, int Wm) {
  a.N = N; a.Hm = Hm; a.Wm = Wm; a.HmWm = Hm * Wm; a.M = N * Hm * Wm;
  a.inv_HmWm = 1.0f / (float)a.HmWm;
  a.inv_Wm = 1.0f / (float)Wm;
}

// above 2^24 pixels or 4 GiB of input: the BIG instantiations (64-bit tile bases, exact index division)
bool is_big(const srx_conv2d_t* d) { return !small_enough(d); }

Plan fwd_plan(const srx_conv2d_t* d, const Geo& g) {
  return make_plan(d->N * g.Ho * g.Wo, g.Cnp, g.Kp / BK, !d->shuffle, d->precision != 0, is_big(d));
}

Plan bwd_plan(const srx_conv2d_t* d, const BwdClass& c) {
  return make_plan(d->N * c.Hm * c.Wm, pad_rows(d->Cin), c.Kp / BK, d->stride == 1, d->precision != 0);
}

int stat_rows_for(const srx_conv2d_t* d) {
  if (srx_rt36_applicable(d)) return srx_rt36_rows(d);
  const Geo g = fwd_geo(d);
  return fwd_plan(d, g).mtiles;
}

}  // namespace

extern "C" size_t srx_conv2d_packed_fwd_floats(const srx_conv2d_t* d) {
  if (check_desc(d)) return 0;
  const Geo g = fwd_geo(d);
  return (size_t)g.Cnp * g.Kp;
}

extern "C" size_t srx_conv2d_packed_bwd_floats(const srx_conv2d_t* d) {
  if (check_desc(d)) return 0;
  if (d->stride > 4) return 0;
  BwdClass cls[16];
  size_t total;
  bwd_classes(d, cls, total);
  return total;
}

extern "C" size_t srx_conv2d_fwd_ws_floats(const srx_conv2d_t* d) {
  if (check_desc(d)) return 0;
  if (srx_rt36_applicable(d)) return 0;
  const Geo g = fwd_geo(d);
  return plan_ws_floats(fwd_plan(d, g));
}

// up = 2 outside the forward pass: the data gradient is taken at the upsampled size into scratch and summed over each
// 2x2 block (the adjoint of nearest upsampling), the weight gradient reads a scratch copy of the upsampled input.
static srx_conv2d_t upsampled_desc(const srx_conv2d_t* d) {
  srx_conv2d_t h = *d;
  h.up = 0; h.H = 2 * d->H; h.W = 2 * d->W;
  return h;
}
static size_t upsampled_floats(const srx_conv2d_t* d) { return (size_t)d->N * 2 * d->H * 2 * d->W * d->Cin_s; }
extern "C" int srx_upsample_nearest2x_fwd(const float* x, float* y, int N, int H, int W, int C, void* stream);
extern "C" int srx_upsample_nearest2x_bwd(const float* dy, float* dx, int N, int H, int W, int C, void* stream);

extern "C" size_t srx_conv2d_bwd_data_ws_floats(const srx_conv2d_t* d) {
  if (check_desc(d)) return 0;
  if (d->up == 2) { const srx_conv2d_t h = upsampled_desc(d); return upsampled_floats(d) + srx_conv2d_bwd_data_ws_floats(&h); }
  if (d->stride != 1) return 0;
  BwdClass cls[16];
  size_t total;
  bwd_classes(d, cls, total);
  return plan_ws_floats(bwd_plan(d, cls[0]));  // (also covers srx_conv2d_bwd_data_act, which never takes the row-tile path)
}

// Row splits of a (group of) weight-gradient problem(s): the slabs cost a write + a read each, and workgroup counts just
// above a whole round of resident workgroups leave a nearly empty last round.  Calibrated on the SRGAN layer shapes
// (bf16: tools/bench_kernels.py --graph, round 1; fp32: tools/wgrad_sweep.sh, round 3); SRX_WGRAD_NSPLIT overrides for experiments.
static int wgrad_nsplit(int M, int64_t tiles, int nprob, int Cnw, int Kw, int precision) {
  const int cus = srx_plan_cus();
  const int max_by_rows = (int)srx_cdiv(M, 128);  // at least 128 rows per split
  int nsplit = 1;
  float best_cost = 1e30f;
  for (int ns = 1; ns <= 64 && ns <= max_by_rows; ++ns) {
    const int rps = (int)srx_roundup(srx_cdiv(M, ns), 32);
    if ((int)srx_cdiv(M, rps) != ns) continue;  // not reachable after rounding to whole chunks
    float cost;
    if (precision == 0) {
      // fp32 (round 3, least-squares fit of 57 in-graph timings of the SRGAN layer shapes, `tools/wgrad_sweep.sh`, rms 4.7 us):
      // three workgroups are resident per CU (VGPRs); a round of three takes 1.70 us per 32-row chunk, a last round of two
      // 1.18 us, of one 0.63 us -- so a workgroup count just ABOVE a multiple of three per CU pays a whole extra round
      // (73728 x 128 x 576: 111 us with 42 splits = 2.95 per CU, 146 us with 43) -- plus 3.5 chunks of fill per round.
      const int w = (int)srx_cdiv(tiles * nprob * ns, cus), f = w / 3, r = w - 3 * f;
      cost = (rps / 32 + 3.5f) * (f * 1.696f + (r == 1 ? 0.633f : r == 2 ? 1.177f : 0.f)) + (float)nprob * ns * Cnw * Kw * 8.0f / 50.0e6f;
    } else {
      const int L = (int)srx_cdiv(tiles * nprob * ns, cus);
      const float hide = L >= 4 ? 0.65f : (L == 3 ? 0.7f : (L == 2 ? 0.8f : 1.0f));
      cost = 1.07f * L * (rps / 32 + 6) * hide + (float)nprob * ns * Cnw * Kw * 8.0f / 3.0e6f;
    }
    if (cost < best_cost) { best_cost = cost; nsplit = ns; }
  }
  if (const int v = srx_dev().wgrad_nsplit; v > 0 && v <= 64 && v <= max_by_rows) nsplit = v;
  const int rps = (int)srx_roundup(srx_cdiv(M, nsplit), 32);
  return (int)srx_cdiv(M, rps);
}

// bf16 products, 3x3 / stride 1 / pad 1, 64 output columns, whole 32-channel groups, image rows of 16 or 32 pixels (ESRGAN's
// dense blocks at the training crop size): the image-row kernel (wgrad_rows_bf16_kernel)
static bool wgrad_rows_ok(const srx_conv2d_t* d) {
  if (srx_dev().no_wgrad_rows || !d->precision || d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != 1 || d->shuffle || d->up == 2) return false;
  return d->Cout == 64 && srx_roundup(d->Cin, 4) % 32 == 0 && (d->W == 32 || d->W == 16);
}
// row splits of the image-row kernel: workgroups = channel groups x problems x splits, four resident per CU
static int wgrad_rows_nsplit(const srx_conv2d_t* d, int nprob) {
  const int cus = srx_plan_cus();
  const int groups = (int)srx_roundup(d->Cin, 4) / 32, rows = d->N * d->H;
  int best = 1;
  float best_cost = 1e30f;
  for (int ns = 1; ns <= 32 && ns <= rows; ++ns) {
    const int rps = (int)srx_cdiv(rows, ns);
    if ((int)srx_cdiv(rows, rps) != ns) continue;
    const int rounds = (int)srx_cdiv((int64_t)groups * nprob * ns, 4 * cus);
    const float cost = rounds * (rps + 8.0f) + 0.25f * ns;  // (+ the slab reduction grows with the splits)
    if (cost < best_cost) { best_cost = cost; best = ns; }
  }
  if (const int v = srx_dev().wgrad_rows_nsplit; v > 0 && v <= 64 && v <= rows) best = v;
  const int rps = (int)srx_cdiv(rows, best);
  return (int)srx_cdiv(rows, rps);
}

extern "C" size_t srx_conv2d_bwd_weight_multi_ws_floats(const srx_conv2d_t* d, int nprob) {
  if (check_desc(d) || nprob < 1 || nprob > WG_MAXP) return 0;
  if (d->up == 2) { const srx_conv2d_t h = upsampled_desc(d); return nprob * upsampled_floats(d) + srx_conv2d_bwd_weight_multi_ws_floats(&h, nprob); }
  if (srx_thin_wgrad_applicable(d))  // (one call per problem) + the column-sum scratch of an optional bias gradient
    return srx_thin_wgrad_ws_floats(d) + srx_colsum_ws_floats((int64_t)d->N * d->H * d->W, d->Cout);
  const Geo g = fwd_geo(d);
  const size_t Cnw = (size_t)srx_roundup(d->Cout, 64), Kw = (size_t)srx_roundup(g.K, 64);
  const int ns = wgrad_rows_ok(d) ? wgrad_rows_nsplit(d, nprob)
                                  : wgrad_nsplit(d->N * g.Ho * g.Wo, (int64_t)(Kw / 64) * (Cnw / 64), nprob, (int)Cnw, (int)Kw, d->precision);
  return Cnw * (Kw + 1) * (size_t)ns * nprob;  // one slab (+ one bias row) per problem and row split
}

extern "C" size_t srx_conv2d_bwd_weight_ws_floats(const srx_conv2d_t* d) { return srx_conv2d_bwd_weight_multi_ws_floats(d, 1); }

extern "C" int srx_conv2d_stat_rows(const srx_conv2d_t* d) {
  if (check_desc(d)) return 0;
  return stat_rows_for(d);
}

// which: 0 = forward, 1 = data gradient.  out[6] = {BM, BN, tail split, workgroups, KS, multi}
// (multi = 1: the stride-parity classes run as one gconv_multi_kernel launch; 2: as one gconv_s2f_kernel launch)
extern "C" int srx_conv2d_plan(const srx_conv2d_t* d, int which, int* out) {
  if (int rc = check_desc(d)) return rc;
  SRX_REQUIRE(out, "conv2d_plan: null pointer");
  Plan p;
  int multi = 0;
  if (srx_rt36_applicable(d)) {  // 36-pixel row tiles (rowtile.hip), forward and data gradient alike
    const int rows = srx_rt36_rows(d);
    out[0] = (int)((int64_t)d->N * d->H * d->W / rows); out[1] = 64; out[2] = 1; out[3] = rows; out[4] = 1; out[5] = 0;
    return SRX_OK;
  }
  if (which == 0) {
    p = fwd_plan(d, fwd_geo(d));
  } else {
    SRX_REQUIRE(d->stride <= 4, "conv2d_plan: stride > 4 unsupported");
    BwdClass cls[16];
    size_t total;
    const int nc = bwd_classes(d, cls, total);
    if (d->stride > 1 && nc <= 4) {
      GMulti m{};
      for (int i = 0; i < nc; ++i) {
        if (cls[i].K == 0 || cls[i].Hm <= 0 || cls[i].Wm <= 0) continue;
        m.g[m.n].M = d->N * cls[i].Hm * cls[i].Wm;
        m.g[m.n].Cn = d->Cin;
        m.g[m.n++].Kp = cls[i].Kp;
      }
      p = Plan{};
      const float t_multi = multi_tile(m, pad_rows(d->Cin), p.BM, p.BN, p.ks, d->precision != 0);
      p.split = 1; p.tail = 0;
      int fbm = 0, fbn = 0;
      const bool fused = m.n == nc && s2f_eligible(d, cls, nc) &&
                         (0.97f * s2f_tile(d, cls, nc, fbm, fbn) < t_multi || (srx_dev().s2_mode & 4)) && fbm > 0;
      if (fused) {  // (multi = 2: all classes of a tile in one workgroup, gconv_s2f_kernel)
        p.BM = fbm; p.BN = fbn; p.ks = 1;
        p.full = (int)srx_cdiv(m.g[0].M, fbm) * (pad_rows(d->Cin) / fbn);
        multi = 2;
      } else {
        for (int i = 0; i < m.n; ++i) p.full += (int)srx_cdiv(m.g[i].M, p.BM) * (int)srx_cdiv(d->Cin, p.BN);
        multi = 1;
      }
    } else {
      p = bwd_plan(d, cls[0]);
    }
  }
  out[0] = p.BM; out[1] = p.BN; out[2] = p.split; out[3] = p.full + p.tail * p.split; out[4] = p.ks; out[5] = multi;
  return SRX_OK;
}

extern "C" int srx_conv2d_pack(const srx_conv2d_t* d, const float* w, float* wpk_fwd, float* wpk_bwd, void* stream) {
  if (int rc = check_desc(d)) return rc;
  SRX_REQUIRE(w && wpk_fwd, "conv2d_pack: null pointer");
  hipStream_t st = srx_stream(stream);
  const Geo g = fwd_geo(d);
  PackArgs pa{};
  pa.w = w; pa.Cout = d->Cout; pa.Cin = d->Cin; pa.KH = d->KH; pa.KW = d->KW; pa.stride = d->stride; pa.pad = d->pad;
  pa.cps = g.cps;
  int64_t maxn = 0;
  if (srx_thin_fwd_applicable(d)) {
    if (int rc = srx_thin_pack(d, w, wpk_fwd, 0, st)) return rc;
  } else {
    PackSeg& sg = pa.seg[pa.nseg++];
    sg = PackSeg{wpk_fwd, g.Cnp, g.K, g.Kp, g.Ck, 0, 0, 0, 0, 0, 1};
    maxn = (int64_t)g.Cnp * g.Kp;
  }
  if (wpk_bwd && srx_thin_dgrad_applicable(d)) {
    if (int rc = srx_thin_pack(d, w, wpk_bwd, 1, st)) return rc;
  } else if (wpk_bwd) {
    SRX_REQUIRE(d->stride <= 4, "conv2d_pack: stride > 4 unsupported for the data gradient");
    BwdClass cls[16];
    size_t total;
    const int nc = bwd_classes(d, cls, total);
    const int Ck = bwd_ck(d);
    const int Cnp = pad_rows(d->Cin);
    for (int i = 0; i < nc; ++i) {
      const BwdClass& c = cls[i];
      PackSeg& sg = pa.seg[pa.nseg++];
      sg = PackSeg{wpk_bwd + c.woff, Cnp, c.K, c.Kp, Ck, 1, c.ph, c.pw, c.dminh, c.dminw, c.ntw > 0 ? c.ntw : 1};
      if ((int64_t)Cnp * c.Kp > maxn) maxn = (int64_t)Cnp * c.Kp;
    }
  }
  if (pa.nseg > 0) {
    hipLaunchKernelGGL(pack_all_kernel, dim3((unsigned)srx_cdiv(maxn, 256), pa.nseg), dim3(256), 0, st, pa);
    SRX_CHECK_LAUNCH("pack_all_kernel");
  }
  return SRX_OK;
}

// ---------------------------------------------------------------------------
// All layers of a model repacked by ONE launch.  The record table is built once on the host (every
// pointer and size in it is fixed for the life of the model), kept in device memory by the caller and
// replayed after each optimiser step: 37 + 8 pack launches per SRGAN step, ~370 per ESRGAN step, become 2.
// ---------------------------------------------------------------------------
struct PackRec {
  float* dst; const float* w;
  long long n_elems;
  int kind;  // 0 forward, 1 data-gradient class, 2 thin forward, 3 thin data gradient
  int rows, K, Kp, Ck, ph, pw, dminh, dminw, ntw;
  int Cout, Cin, KH, KW, stride, pad, cps, pad_;
};

__global__ void pack_table_kernel(const PackRec* __restrict__ table) {
  const PackRec r = table[blockIdx.y];
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= r.n_elems) return;
  float v = 0.f;
  if (r.kind >= 4) {  // Winograd-domain weights (wino.hip): 4 the layer, 5 its data gradient, 6 the layer with a PixelShuffle store
    srx_wino_pack_one(r.w, r.dst, r.Cout, r.Cin, r.kind == 5 ? 1 : (r.kind == 6 ? 2 : 0), idx);
    return;
  }
  if (r.kind >= 2) {  // thin.hip layout: p[c][tap][ch]
    const int taps = r.KH * r.KW;
    const int ch = (int)(idx & 63), tap = (int)((idx >> 6) % taps), c = (int)(idx / (64 * taps));
    const int Cthin = r.kind == 2 ? r.Cout : r.Cin;
    if (c < Cthin) {
      const int kh = tap / r.KW, kw = tap - kh * r.KW;
      v = r.kind == 2 ? r.w[(((size_t)c * 64 + ch) * r.KH + kh) * r.KW + kw]
                      : r.w[(((size_t)ch * Cthin + c) * r.KH + (r.KH - 1 - kh)) * r.KW + (r.KW - 1 - kw)];
    }
    r.dst[idx] = v;
    return;
  }
  // One thread per (packed row, channel): it reads the channel's taps -- adjacent floats of the OIHW weight, so a wave reads one
  // contiguous stretch once -- and writes each to its k = tap * Ck + channel (a wave: 256 contiguous bytes per tap).  One thread
  // per DESTINATION element made every tap's wave pull the same lines through the L2 again (9x the weight bytes for 3x3).
  // The threads behind the (row, channel) range zero the K .. Kp padding.
  const int taps = r.K / r.Ck;  // K = taps * Ck
  const int64_t nmain = (int64_t)r.rows * r.Ck;
  if (idx >= nmain) {
    const int padk = r.Kp - r.K;
    const int64_t j = idx - nmain;
    if (padk > 0 && j < (int64_t)r.rows * padk) {
      const int row = (int)(j / padk);
      r.dst[(size_t)row * r.Kp + r.K + (int)(j - (int64_t)row * padk)] = 0.f;
    }
    return;
  }
  const int row = (int)(idx / r.Ck), c = (int)(idx - (int64_t)row * r.Ck);
  float* d = r.dst + (size_t)row * r.Kp + c;
  const bool live = r.kind == 0 ? (row < r.Cout && c < r.Cin) : (row < r.Cin && c < r.Cout);
  int co = r.kind == 0 ? row : c;  // the conv's output channel this element belongs to
  if (r.cps) { const int ij = co / r.cps, cc = co - ij * r.cps; co = cc * 4 + ij; }
  const float* src = r.w + ((size_t)co * r.Cin + (r.kind == 0 ? c : row)) * (r.KH * r.KW);
  for (int t0 = 0; t0 < taps; t0 += 9) {
    float v[9];
#pragma unroll
    for (int u = 0; u < 9; ++u) {
      const int tap = min(t0 + u, taps - 1);
      int off = tap;  // forward: k runs over (kh, kw) in the weight's own order
      if (r.kind == 1) {
        const int th = tap / r.ntw, tw = tap - th * r.ntw;
        off = (r.ph + r.pad - r.stride * (r.dminh + th)) * r.KW + (r.pw + r.pad - r.stride * (r.dminw + tw));
      }
      v[u] = live ? src[off] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 9; ++u)
      if (t0 + u < taps) d[(size_t)(t0 + u) * r.Ck] = v[u];
  }
}

extern "C" size_t srx_pack_table_bytes(int n_layers) { return (size_t)n_layers * 17 * sizeof(PackRec); }

extern "C" int srx_pack_table_build(const srx_conv2d_t* descs, int n, const float* const* w, float* const* wpk_fwd,
                                    float* const* wpk_bwd, void* host_table, int* nrec_out, long long* max_elems_out) {
  SRX_REQUIRE(descs && w && wpk_fwd && wpk_bwd && host_table && nrec_out && max_elems_out && n > 0,
              "pack_table_build: bad argument");
  PackRec* out = static_cast<PackRec*>(host_table);
  int nrec = 0;
  long long maxn = 0;
  for (int i = 0; i < n; ++i) {
    const srx_conv2d_t* d = descs + i;
    if (int rc = check_desc(d)) return rc;
    SRX_REQUIRE(w[i] && wpk_fwd[i], "pack_table_build: null pointer in layer %d", i);
    const Geo g = fwd_geo(d);
    PackRec base{};
    base.w = w[i];
    base.Cout = d->Cout; base.Cin = d->Cin; base.KH = d->KH; base.KW = d->KW; base.stride = d->stride; base.pad = d->pad;
    base.cps = g.cps;
    auto emit = [&](PackRec r) {
      if (r.n_elems > maxn) maxn = r.n_elems;
      out[nrec++] = r;
    };
    if (srx_thin_fwd_applicable(d)) {
      PackRec r = base; r.kind = 2; r.dst = wpk_fwd[i]; r.n_elems = 4LL * d->KH * d->KW * 64; emit(r);
    } else {
      PackRec r = base; r.kind = 0; r.dst = wpk_fwd[i]; r.rows = g.Cnp; r.K = g.K; r.Kp = g.Kp; r.Ck = g.Ck; r.ntw = 1;
      r.n_elems = (long long)g.Cnp * (g.Ck + g.Kp - g.K); emit(r);  // work items: (row, channel) pairs + the K..Kp padding
    }
    if (!wpk_bwd[i]) continue;
    if (srx_thin_dgrad_applicable(d)) {
      PackRec r = base; r.kind = 3; r.dst = wpk_bwd[i]; r.n_elems = 4LL * d->KH * d->KW * 64; emit(r);
      continue;
    }
    SRX_REQUIRE(d->stride <= 4, "pack_table_build: stride > 4 unsupported for the data gradient");
    BwdClass cls[16];
    size_t total;
    const int nc = bwd_classes(d, cls, total);
    const int Cnp = pad_rows(d->Cin);
    for (int c = 0; c < nc; ++c) {
      PackRec r = base; r.kind = 1; r.dst = wpk_bwd[i] + cls[c].woff;
      r.rows = Cnp; r.K = cls[c].K; r.Kp = cls[c].Kp; r.Ck = bwd_ck(d);
      r.ph = cls[c].ph; r.pw = cls[c].pw; r.dminh = cls[c].dminh; r.dminw = cls[c].dminw;
      r.ntw = cls[c].ntw > 0 ? cls[c].ntw : 1;
      r.n_elems = (long long)Cnp * (r.Ck + cls[c].Kp - cls[c].K);
      emit(r);
    }
  }
  *nrec_out = nrec;
  *max_elems_out = maxn;
  return SRX_OK;
}

// appends the record that refreshes a layer's Winograd-domain weights (srx_wino_pack) to a host table under construction
extern "C" int srx_pack_table_add_wino(void* host_table, int* nrec, long long* max_elems, const srx_conv2d_t* d, const float* w,
                                       float* upk, int transpose) {
  SRX_REQUIRE(host_table && nrec && max_elems && d && w && upk && *nrec >= 0, "pack_table_add_wino: bad argument");
  SRX_REQUIRE(srx_wino_applicable(d) || srx_wino_packed_floats(d) > 0, "pack_table_add_wino: not a Winograd layer");
  PackRec r{};
  SRX_REQUIRE(!(transpose && d->shuffle), "pack_table_add_wino: a PixelShuffle layer has no Winograd data gradient");
  r.dst = upk; r.w = w; r.kind = transpose ? 5 : (d->shuffle ? 6 : 4);
  r.Cout = d->Cout; r.Cin = d->Cin; r.KH = 3; r.KW = 3; r.stride = 1; r.pad = 1;
  r.n_elems = (long long)d->Cout * d->Cin;
  static_cast<PackRec*>(host_table)[(*nrec)++] = r;
  if (r.n_elems > *max_elems) *max_elems = r.n_elems;
  return SRX_OK;
}

extern "C" int srx_pack_table_run(const void* dev_table, int nrec, long long max_elems, void* stream) {
  SRX_REQUIRE(dev_table && nrec > 0 && nrec <= 65535 && max_elems > 0, "pack_table_run: bad argument");
  hipLaunchKernelGGL(pack_table_kernel, dim3((unsigned)srx_cdiv(max_elems, 256), (unsigned)nrec), dim3(256), 0,
                     srx_stream(stream), static_cast<const PackRec*>(dev_table));
  SRX_CHECK_LAUNCH("pack_table_kernel");
  return SRX_OK;
}

static int conv_fwd_impl(const srx_conv2d_t* d, const float* x, const float* wpk, const float* bias, const float* residual,
                         float out_scale, float* y, float* bn_partials, float* ws, size_t ws_floats, void* stream) {
  if (int rc = check_desc(d)) return rc;
  SRX_REQUIRE(x && wpk && y, "conv2d_fwd: null pointer");
  if (residual && d->shuffle) SRX_FAIL(SRX_E_UNSUPPORTED, "conv2d_fwd: residual with PixelShuffle is not implemented");
  hipStream_t st = srx_stream(stream);
  if (srx_thin_fwd_applicable(d) && !bn_partials && !residual && d->up != 2) return srx_thin_fwd(d, x, wpk, bias, y, d->Cout, st);
  if (srx_first3_fwd_applicable(d) && !bn_partials && !residual && d->act != SRX_ACT_PRELU)
    return srx_first3_fwd(d, x, wpk, fwd_geo(d).Kp, bias, y, st);
  if (srx_rt36_applicable(d) && out_scale == 1.f)
    return srx_rt36_run(d, x, wpk, bias, residual, y, bn_partials, d->act, d->slope, st);
  if (srx_rt36_applicable(d)) SRX_FAIL(SRX_E_UNSUPPORTED, "conv2d_fwd_residual: out_scale != 1 on the 36-pixel row tile");
  const Geo g = fwd_geo(d);
  GArgs a{};
  a.in = x; a.w = wpk; a.bias = bias; a.part = nullptr;
  set_mgrid(a, d->N, g.Ho, g.Wo);
  a.up = d->up == 2;
  a.Hi = (a.up ? 2 : 1) * d->H; a.Wi = (a.up ? 2 : 1) * d->W; a.Ci = d->Cin_s;
  a.in_stride = d->stride; a.nth = d->KH; a.ntw = d->KW; a.dh0 = -d->pad; a.dw0 = -d->pad;
  a.Ck = g.Ck; a.K = g.K; a.Kp = g.Kp;
  a.Cn = d->Cout;
  a.act = d->act;
  a.slope = d->act == SRX_ACT_RELU ? 0.f : (d->act == SRX_ACT_LRELU ? d->slope : 1.f);  // v > 0 ? v : v * slope
  a.in_shuffle = 0;
  if (d->shuffle) {
    a.out_shuffle = g.cps; a.Cs = d->Cout; a.Ho = 2 * g.Ho; a.Wo = 2 * g.Wo; a.Co = d->Cout_s;
    a.out_stride = 2; a.oh_off = 0; a.ow_off = 0; a.linear_out = 0;
  } else {
    a.out_shuffle = 0; a.Cs = (int)srx_roundup(d->Cout, 4); a.Ho = g.Ho; a.Wo = g.Wo; a.Co = d->Cout_s;
    a.out_stride = 1; a.oh_off = 0; a.ow_off = 0; a.linear_out = 1;
  }
  a.part = bn_partials;
  a.out = y;
  a.add = residual;
  a.oscale = out_scale;
  a.add_ld = a.Co; a.add_hi = 0x7fffffff; a.ascale = 1.f;
  a.in_bytes = (size_t)d->N * d->H * d->W * d->Cin_s * sizeof(float);
  a.w_bytes = (unsigned)((size_t)g.Cnp * g.Kp * sizeof(float));
  a.big = is_big(d);
  a.in_margin = a.up ? 0u : 4u * (unsigned)((d->pad * a.Wi + d->pad) * a.Ci);  // taps start `pad` rows and columns before the centre
  return run_gconv(a, fwd_plan(d, g), ws, ws_floats, st, d->precision);
}

extern "C" int srx_conv2d_fwd(const srx_conv2d_t* d, const float* x, const float* wpk, const float* bias, float* y,
                              float* bn_partials, float* ws, size_t ws_floats, void* stream) {
  return conv_fwd_impl(d, x, wpk, bias, nullptr, 1.f, y, bn_partials, ws, ws_floats, stream);
}

// The forward of a 64 -> <= 4 channel layer (the generator's output conv, srgan/generator.py:58) whose INPUT is a bf16 NHWC
// tensor -- the end of the bf16-native inference chain (c64.hip); precision must be 2 (bf16 products), y is fp32.
extern "C" int srx_conv2d_fwd_bf16in(const srx_conv2d_t* d, const void* x_bf16, const float* wpk, const float* bias, float* y,
                                     void* stream) {
  if (int rc = check_desc(d)) return rc;
  SRX_REQUIRE(x_bf16 && wpk && y, "conv2d_fwd_bf16in: null pointer");
  if (!srx_thin_fwd_applicable(d) || d->precision != 2 || d->up == 2)
    SRX_FAIL(SRX_E_UNSUPPORTED, "conv2d_fwd_bf16in: 64 -> <= 4 channel layers with precision = 2 only");
  SRX_REQUIRE((int64_t)d->N * d->H * d->W * 128 < (1LL << 32), "conv2d_fwd_bf16in: input above 4 GiB; tile the image");
  return srx_thin_fwd(d, static_cast<const float*>(x_bf16), wpk, bias, y, d->Cout, srx_stream(stream), 1);
}

extern "C" int srx_conv2d_fwd_residual(const srx_conv2d_t* d, const float* x, const float* wpk, const float* bias,
                                       const float* residual, float out_scale, float* y, float* ws, size_t ws_floats,
                                       void* stream) {
  SRX_REQUIRE(residual, "conv2d_fwd_residual: null residual");
  return conv_fwd_impl(d, x, wpk, bias, residual, out_scale, y, nullptr, ws, ws_floats, stream);
}

static int conv_bwd_data_impl(const srx_conv2d_t* d, const float* dy, const float* wpk_bwd, float* dx, int accumulate,
                              const float* act_out, float act_slope, int c_lo, int c_hi, float* ws, size_t ws_floats,
                              void* stream, const float* addend = nullptr, int addend_ld = 0, int addend_channels = 0,
                              float addend_scale = 1.f, float out_scale = 1.f) {
  if (int rc = check_desc(d)) return rc;
  SRX_REQUIRE(dy && wpk_bwd && dx, "conv2d_bwd_data: null pointer");
  SRX_REQUIRE(d->stride <= 4, "conv2d_bwd_data: stride > 4 unsupported");
  if (addend_ld == 0) addend_ld = d->Cin_s;
  if (addend_channels == 0) addend_channels = 0x7fffffff;
  const bool plain_addend = addend_ld == d->Cin_s && addend_channels >= d->Cin && addend_scale == 1.f && out_scale == 1.f;
  if (!plain_addend) {
    SRX_REQUIRE(addend && !accumulate, "conv2d_bwd_data_ex: addend stride / channels / scales without an addend");
    SRX_REQUIRE(addend_ld >= 4 && addend_ld % 4 == 0 && (addend_channels % 4 == 0 || addend_channels >= d->Cin),
                "conv2d_bwd_data_ex: the addend's stride and channel count must be made of whole quads");
    // the epilogue reads min(addend_channels, Cin rounded up to a quad) floats of every addend row: more than the
    // row stride would walk into the next pixel and, on the last rows, past the end of the addend tensor
    SRX_REQUIRE(addend_channels > 0 && (addend_channels < d->Cin ? addend_channels : (int)srx_roundup(d->Cin, 4)) <= addend_ld,
                "conv2d_bwd_data_ex: the addend has %d channels per row but a row stride of %d floats",
                addend_channels < d->Cin ? addend_channels : (int)srx_roundup(d->Cin, 4), addend_ld);
  }
  if (d->up == 2) {
    if (accumulate || act_out || addend)
      SRX_FAIL(SRX_E_UNSUPPORTED, "conv2d_bwd_data: up = 2 with accumulate / an addend / a folded activation");
    SRX_REQUIRE(small_enough(d), "conv2d_bwd_data: up = 2 above 2^24 pixels (the 2x2 block sum keeps 32-bit offsets)");
    const srx_conv2d_t h = upsampled_desc(d);
    const size_t tmp = upsampled_floats(d);
    SRX_REQUIRE(ws && ws_floats >= tmp + srx_conv2d_bwd_data_ws_floats(&h), "conv2d_bwd_data: workspace too small for up = 2");
    if (int rc = conv_bwd_data_impl(&h, dy, wpk_bwd, ws, 0, nullptr, 1.f, 0, 0, ws + tmp, ws_floats - tmp, stream)) return rc;
    return srx_upsample_nearest2x_bwd(ws, dx, d->N, d->H, d->W, d->Cin_s, stream);
  }
  // (the row-tile kernel has neither a masked epilogue nor a strided / scaled addend)
  const bool rt36 = srx_rt36_applicable(d) && !act_out && plain_addend;
  if (addend && (accumulate || d->stride != 1 || srx_thin_dgrad_applicable(d) || d->up == 2))
    SRX_FAIL(SRX_E_UNSUPPORTED, "conv2d_bwd_data_add: stride-1 layers without accumulate only");
  if (accumulate && (d->stride != 1 || srx_thin_dgrad_applicable(d) || rt36))
    SRX_FAIL(SRX_E_UNSUPPORTED, "conv2d_bwd_data: accumulate is implemented for stride-1 layers on the generic kernel only");
  if (act_out && (srx_thin_dgrad_applicable(d) || (d->stride != 1 && accumulate)))
    SRX_FAIL(SRX_E_UNSUPPORTED, "conv2d_bwd_data_act: layers on the generic kernel only (strided ones without accumulate)");
  if (act_out) SRX_REQUIRE(c_lo >= 0 && c_lo < c_hi && c_lo % 4 == 0 && (c_hi % 4 == 0 || c_hi >= d->Cin) && c_lo < d->Cin_s,
                           "conv2d_bwd_data_act: the masked channel range must be made of whole quads inside the row");
  hipStream_t st = srx_stream(stream);
  if (srx_thin_dgrad_applicable(d)) return srx_thin_fwd(d, dy, wpk_bwd, nullptr, dx, d->Cin, st);
  const Geo g = fwd_geo(d);
  BwdClass cls[16];
  size_t total;
  const int nc = bwd_classes(d, cls, total);
  if (rt36)  // 3x3 / stride 1 / pad 1: one class, same geometry as the forward, flipped taps
    return srx_rt36_run(d, dy, wpk_bwd + cls[0].woff, nullptr, addend, dx, nullptr, SRX_ACT_NONE, 0.f, st);
  bool any_empty = false;
  for (int i = 0; i < nc; ++i) any_empty |= (cls[i].K == 0);
  if (any_empty) {
    if (hipMemsetAsync(dx, 0, (size_t)d->N * d->H * d->W * d->Cin_s * sizeof(float), st) != hipSuccess)
      SRX_FAIL(SRX_E_HIP, "conv2d_bwd_data: memset failed");
  }
  const size_t dy_bytes = (size_t)d->N * g.Ho * g.Wo * (d->shuffle ? 4 : 1) * d->Cout_s * sizeof(float);
  SRX_REQUIRE(small_enough(d) && dy_bytes < 0xfffffff0ull,
              "conv2d_bwd_data: more than 2^24 pixels or 4 GiB per tensor (data gradients keep 32-bit offsets: training crops, not whole frames)");

  GMulti multi{};
  for (int i = 0; i < nc; ++i) {
    const BwdClass& c = cls[i];
    if (c.K == 0 || c.Hm <= 0 || c.Wm <= 0) continue;
    GArgs a{};
    a.in = dy; a.w = wpk_bwd + c.woff; a.bias = nullptr; a.part = nullptr;
    set_mgrid(a, d->N, c.Hm, c.Wm);
    a.Hi = g.Ho; a.Wi = g.Wo; a.Ci = d->Cout_s;
    a.in_stride = 1; a.nth = c.nth; a.ntw = c.ntw; a.dh0 = c.dminh; a.dw0 = c.dminw;
    a.Ck = bwd_ck(d);
    a.K = c.K; a.Kp = c.Kp;
    a.in_shuffle = g.cps;
    a.Cn = d->Cin; a.Cs = (int)srx_roundup(d->Cin, 4);
    a.Ho = d->H; a.Wo = d->W; a.Co = d->Cin_s;
    a.out_stride = d->stride; a.oh_off = c.ph; a.ow_off = c.pw;
    a.out_shuffle = 0;
    a.act = SRX_ACT_NONE; a.slope = 1.f;
    a.linear_out = (d->stride == 1);
    a.out = dx;
    a.in_bytes = dy_bytes;
    a.big = 0;
    {  // the most negative tap of this class: dminh rows / dminw columns (in units of the gradient tensor's pixels, x2 when shuffled)
      const long long o = g.cps ? ((2LL * c.dminh) * (2 * a.Wi) + 2LL * c.dminw) * a.Ci : ((long long)c.dminh * a.Wi + c.dminw) * a.Ci;
      // A class's grid (Hm x Wm) may be larger than the gradient image (pad = 0: the last input rows / columns see no output):
      // such rows' "centre" lies past their image, so a LATER row (the next image's first) can sit lower in memory -- the
      // tile's base must leave room for that overshoot too
      const long long over_h = std::max(0, c.Hm - a.Hi), over_w = std::max(0, c.Wm - a.Wi);
      const long long over = g.cps ? ((2 * over_h) * (2LL * a.Wi) + 2 * over_w) * a.Ci : (over_h * a.Wi + over_w) * a.Ci;
      a.in_margin = (unsigned)(((o < 0 ? -o : 0) + over) * 4);
    }
    a.w_bytes = (unsigned)((size_t)pad_rows(d->Cin) * c.Kp * sizeof(float));
    a.add = accumulate ? dx : addend;
    a.oscale = out_scale;
    a.add_ld = accumulate ? a.Co : addend_ld; a.add_hi = accumulate ? 0x7fffffff : addend_channels;
    a.ascale = accumulate ? 1.f : addend_scale;
    if (act_out) { a.mask = act_out; a.mask_slope = act_slope; a.mask_lo = c_lo; a.mask_hi = c_hi; }
    if (d->stride == 1) {
      if (int rc = run_gconv(a, bwd_plan(d, c), ws, ws_floats, st, d->precision)) return rc;
    } else if (nc <= 4) {
      multi.g[multi.n++] = a;
    } else {  // stride > 2: one launch per class (with the layer's arithmetic: bf16 products under precision = 1)
      if (int rc = run_gconv(a, bwd_plan(d, c), ws, ws_floats, st, d->precision)) return rc;
    }
  }
  if (multi.n > 0) {
    int bm, bn, ks;
    const float t_multi = multi_tile(multi, pad_rows(d->Cin), bm, bn, ks, d->precision != 0);
    if (multi.n == nc && s2f_eligible(d, cls, nc)) {  // all classes of a tile in one workgroup, when that is the shorter launch
      int fbm, fbn;
      const float t_fused = s2f_tile(d, cls, nc, fbm, fbn);
      if (fbm > 0 && (0.97f * t_fused < t_multi || (srx_dev().s2_mode & 4)))  // (a tie goes to the fused launch: measured)
        return run_gconv_s2f(d, multi.g[0], cls, nc, wpk_bwd, total, fbm, fbn, st);
    }
    if (int rc = run_gconv_multi(multi, bm, bn, ks, st, d->precision)) return rc;
  }
  return SRX_OK;
}

extern "C" int srx_conv2d_bwd_data(const srx_conv2d_t* d, const float* dy, const float* wpk_bwd, float* dx,
                                   int accumulate, float* ws, size_t ws_floats, void* stream) {
  return conv_bwd_data_impl(d, dy, wpk_bwd, dx, accumulate, nullptr, 1.f, 0, 0, ws, ws_floats, stream);
}

extern "C" int srx_conv2d_bwd_data_add(const srx_conv2d_t* d, const float* dy, const float* wpk_bwd, const float* addend,
                                       float* dx, float* ws, size_t ws_floats, void* stream) {
  SRX_REQUIRE(addend && addend != dx, "conv2d_bwd_data_add: the addend must be a tensor of its own");
  return conv_bwd_data_impl(d, dy, wpk_bwd, dx, 0, nullptr, 1.f, 0, 0, ws, ws_floats, stream, addend);
}

extern "C" int srx_conv2d_bwd_data_act(const srx_conv2d_t* d, const float* dy, const float* wpk_bwd, const float* x,
                                       float slope, int c_lo, int c_hi, int accumulate, float* dx, float* ws,
                                       size_t ws_floats, void* stream) {
  SRX_REQUIRE(x, "conv2d_bwd_data_act: null activation tensor");
  return conv_bwd_data_impl(d, dy, wpk_bwd, dx, accumulate, x, slope, c_lo, c_hi, ws, ws_floats, stream);
}

// Rows of the BatchNorm-backward partial table srx_conv2d_bwd_data_bn writes for this layer; 0: the layer does not run on the
// kernel that has that epilogue (the 36-pixel row tile) and the caller keeps the separate srx_bn_act_bwd
extern "C" int srx_conv2d_bwd_data_bn_rows(const srx_conv2d_t* d) {
  if (check_desc(d) != SRX_OK || !srx_rt36_applicable(d) || d->Cin != 64) return 0;
  return srx_rt36_rows(d);
}

extern "C" int srx_conv2d_bwd_data_bn(const srx_conv2d_t* d, const float* dy, const float* wpk_bwd, const float* addend, float* dx,
                                      const float* bn_y, const float* bn_mean, const float* bn_invstd, const float* bn_gamma,
                                      const float* bn_beta, const float* bn_prelu, float* table, void* stream) {
  if (int rc = check_desc(d)) return rc;
  SRX_REQUIRE(dy && wpk_bwd && dx && bn_y && bn_mean && bn_invstd && bn_gamma && bn_beta && table, "conv2d_bwd_data_bn: null pointer");
  SRX_REQUIRE(!addend || addend != dx, "conv2d_bwd_data_bn: the addend must be a tensor of its own");
  if (srx_conv2d_bwd_data_bn_rows(d) == 0)
    SRX_FAIL(SRX_E_UNSUPPORTED, "conv2d_bwd_data_bn: 3x3 / 64 -> 64 / stride 1 layers on the 36-pixel row tile only (srx_conv2d_bwd_data_bn_rows)");
  BwdClass cls[16];
  size_t total;
  (void)bwd_classes(d, cls, total);
  const srx_rt36_bn_t bn{bn_y, bn_mean, bn_invstd, bn_gamma, bn_beta, bn_prelu, table};
  return srx_rt36_run(d, dy, wpk_bwd + cls[0].woff, nullptr, addend, dx, nullptr, SRX_ACT_NONE, 0.f, srx_stream(stream), &bn);
}

// srx_conv2d_bwd_data_bn whose INPUT is not yet the conv's output gradient but the gradient arriving at the output of the
// BatchNorm (+ PReLU) layer ABOVE the conv (srgan/residual.py:89-90 / :87-88 read backwards): the second pass of that layer's
// backward (sums finalised: srx_bn_act_bwd_finish with dy = NULL) runs while the patch is staged, dy_out receives the conv's
// output gradient for its weight gradient.  table == NULL: no BatchNorm below (plain data gradient + addend).
extern "C" int srx_conv2d_bwd_data_bn_in_ok(const srx_conv2d_t* d) {
  return (!srx_dev().no_bn_bwd_fuse && srx_conv2d_bwd_data_bn_rows(d) != 0) ? 1 : 0;
}

extern "C" int srx_conv2d_bwd_data_bn_in(const srx_conv2d_t* d, const float* dout, const float* in_y, const float* in_mean,
                                         const float* in_invstd, const float* in_gamma, const float* in_beta, const float* in_prelu,
                                         const float* in_sums, float* dy_out, const float* wpk_bwd, const float* addend, float* dx,
                                         const float* bn_y, const float* bn_mean, const float* bn_invstd, const float* bn_gamma,
                                         const float* bn_beta, const float* bn_prelu, float* table, void* stream) {
  if (int rc = check_desc(d)) return rc;
  SRX_REQUIRE(dout && in_y && in_mean && in_invstd && in_gamma && in_beta && in_sums && dy_out && wpk_bwd && dx,
              "conv2d_bwd_data_bn_in: null pointer");
  SRX_REQUIRE(dy_out != dout && dy_out != dx && dx != dout && (!addend || addend != dx), "conv2d_bwd_data_bn_in: dout, dy_out, dx and the addend must be tensors of their own");
  SRX_REQUIRE(!table || (bn_y && bn_mean && bn_invstd && bn_gamma && bn_beta), "conv2d_bwd_data_bn_in: a table needs the BatchNorm below");
  if (!srx_conv2d_bwd_data_bn_in_ok(d))
    SRX_FAIL(SRX_E_UNSUPPORTED, "conv2d_bwd_data_bn_in: 3x3 / 64 -> 64 / stride 1 layers on the 36-pixel row tile only (srx_conv2d_bwd_data_bn_in_ok)");
  BwdClass cls[16];
  size_t total;
  (void)bwd_classes(d, cls, total);
  const srx_rt36_bn_t bn{bn_y, bn_mean, bn_invstd, bn_gamma, bn_beta, bn_prelu, table};
  const srx_rt36_bnb_t bnb{in_y, in_mean, in_invstd, in_gamma, in_beta, in_prelu, in_sums,
                           1.0f / (float)((int64_t)d->N * d->H * d->W), dy_out};
  return srx_rt36_run(d, dout, wpk_bwd + cls[0].woff, nullptr, addend, dx, nullptr, SRX_ACT_NONE, 0.f, srx_stream(stream),
                      table ? &bn : nullptr, nullptr, &bnb);
}

// The forward of a conv whose INPUT is act(BatchNorm(y_in)) [+ residual] of the conv below (training mode, statistics finalised): the
// normalise + activate pass runs while the input patch is staged, and act_out receives the tensor that pass would have
// written.  1: this layer can (the 36-pixel row tile: 3x3, 64 -> 64, stride 1, few pixels); 0: keep srx_bn_act_fwd + srx_conv2d_fwd.
extern "C" int srx_conv2d_fwd_bn_in_ok(const srx_conv2d_t* d) {
  return (!srx_dev().no_bn_fwd_fuse && check_desc(d) == SRX_OK && srx_rt36_applicable(d)) ? 1 : 0;
}

extern "C" int srx_conv2d_fwd_bn_in(const srx_conv2d_t* d, const float* y_in, const float* bn_mean, const float* bn_invstd,
                                    const float* bn_gamma, const float* bn_beta, const float* bn_prelu, const float* bn_residual,
                                    float* act_out, const float* wpk, const float* bias, float* y, float* bn_partials, void* stream) {
  if (int rc = check_desc(d)) return rc;
  SRX_REQUIRE(y_in && bn_mean && bn_invstd && bn_gamma && bn_beta && act_out && wpk && y, "conv2d_fwd_bn_in: null pointer");
  SRX_REQUIRE(act_out != y_in && act_out != y && y != y_in && act_out != bn_residual && y != bn_residual,
              "conv2d_fwd_bn_in: y_in, act_out, y and the residual must be tensors of their own");
  if (!srx_conv2d_fwd_bn_in_ok(d))
    SRX_FAIL(SRX_E_UNSUPPORTED, "conv2d_fwd_bn_in: 3x3 / 64 -> 64 / stride 1 layers on the 36-pixel row tile only (srx_conv2d_fwd_bn_in_ok)");
  const srx_rt36_bnl_t bnl{bn_mean, bn_invstd, bn_gamma, bn_beta, bn_prelu, bn_residual, act_out};
  return srx_rt36_run(d, y_in, wpk, bias, nullptr, y, bn_partials, d->act, d->slope, srx_stream(stream), nullptr, &bnl);
}

extern "C" int srx_conv2d_bwd_data_ex(const srx_conv2d_t* d, const float* dy, const float* wpk_bwd, float* dx,
                                      const srx_dgrad_epilogue_t* e, float* ws, size_t ws_floats, void* stream) {
  SRX_REQUIRE(e, "conv2d_bwd_data_ex: null epilogue");
  SRX_REQUIRE(!e->addend || e->addend != dx, "conv2d_bwd_data_ex: the addend must be a tensor of its own (use accumulate)");
  return conv_bwd_data_impl(d, dy, wpk_bwd, dx, e->accumulate, e->act_out, e->act_slope, e->c_lo, e->c_hi, ws, ws_floats,
                            stream, e->addend, e->addend_ld, e->addend_channels, e->addend_scale == 0.f ? 1.f : e->addend_scale,
                            e->out_scale == 0.f ? 1.f : e->out_scale);
}

extern "C" int srx_colsum(const float* x, float* out, int64_t M, int C, int Cs, int accumulate, float* ws,
                          size_t ws_floats, void* stream);
extern "C" size_t srx_colsum_ws_floats(int64_t M, int C);

extern "C" int srx_conv2d_bwd_weight_multi_scaled(const srx_conv2d_t* d, int nprob, int per_out, const float* const* xs,
                                                  const float* const* dys, float* const* dws, int accumulate,
                                                  float* const* dbs, const float* out_scales, float* ws, size_t ws_floats,
                                                  void* stream);

extern "C" int srx_conv2d_bwd_weight_multi(const srx_conv2d_t* d, int nprob, int per_out, const float* const* xs,
                                           const float* const* dys, float* const* dws, int accumulate, float* const* dbs,
                                           float* ws, size_t ws_floats, void* stream) {
  return srx_conv2d_bwd_weight_multi_scaled(d, nprob, per_out, xs, dys, dws, accumulate, dbs, nullptr, ws, ws_floats, stream);
}

static int wgrad_multi_impl(const srx_conv2d_t* d, int nprob, int per_out, const float* const* xs, const float* const* dys,
                            float* const* dws, int accumulate, float* const* dbs, const float* out_scales,
                            float* const* dws_hi, float* const* dbs_hi, int cin_lo, float* ws, size_t ws_floats,
                            void* stream);

extern "C" int srx_conv2d_bwd_weight_multi_scaled(const srx_conv2d_t* d, int nprob, int per_out, const float* const* xs,
                                                  const float* const* dys, float* const* dws, int accumulate,
                                                  float* const* dbs, const float* out_scales, float* ws, size_t ws_floats,
                                                  void* stream) {
  return wgrad_multi_impl(d, nprob, per_out, xs, dys, dws, accumulate, dbs, out_scales, nullptr, nullptr, 0, ws, ws_floats,
                          stream);
}

extern "C" int srx_conv2d_bwd_weight_multi_pair(const srx_conv2d_t* d, int nprob, const float* const* xs,
                                                const float* const* dys, float* const* dws_lo, float* const* dws_hi,
                                                int cin_lo, int accumulate, float* const* dbs_lo, float* const* dbs_hi,
                                                float* ws, size_t ws_floats, void* stream) {
  SRX_REQUIRE(d && dws_hi, "conv2d_bwd_weight_multi_pair: null pointer");
  SRX_REQUIRE(d->Cout % 8 == 0 && !d->shuffle && d->up != 2 && cin_lo > 0 && cin_lo <= d->Cin && !srx_thin_wgrad_applicable(d),
              "conv2d_bwd_weight_multi_pair: two convs of Cout / 2 output channels each (a multiple of 4), no PixelShuffle, "
              "no fused upsample, 0 < cin_lo <= Cin");
  SRX_REQUIRE((dbs_lo == nullptr) == (dbs_hi == nullptr), "conv2d_bwd_weight_multi_pair: bias gradients for both convs or neither");
  SRX_REQUIRE(cin_lo % 4 == 0 || cin_lo == d->Cin, "conv2d_bwd_weight_multi_pair: cin_lo must be a whole number of quads");
  return wgrad_multi_impl(d, nprob, 1, xs, dys, dws_lo, accumulate, dbs_lo, nullptr, dws_hi, dbs_hi, cin_lo, ws, ws_floats, stream);
}

static int wgrad_multi_impl(const srx_conv2d_t* d, int nprob, int per_out, const float* const* xs, const float* const* dys,
                            float* const* dws, int accumulate, float* const* dbs, const float* out_scales,
                            float* const* dws_hi, float* const* dbs_hi, int cin_lo, float* ws, size_t ws_floats,
                            void* stream) {
  if (int rc = check_desc(d)) return rc;
  SRX_REQUIRE(nprob >= 1 && nprob <= WG_MAXP && per_out >= 1 && nprob % per_out == 0,
              "conv2d_bwd_weight_multi: 1..%d problems, a whole number of outputs", WG_MAXP);
  SRX_REQUIRE(xs && dys && dws && ws, "conv2d_bwd_weight: null pointer");
  SRX_REQUIRE(small_enough(d), "conv2d_bwd_weight: more than 2^24 pixels or 4 GiB of input per problem (the weight-gradient kernels "
                               "keep 32-bit tensor offsets: training crops, not whole frames)");
  if (d->up == 2) {
    const srx_conv2d_t h = upsampled_desc(d);
    const size_t tmp = upsampled_floats(d);
    SRX_REQUIRE(ws_floats >= nprob * tmp + srx_conv2d_bwd_weight_multi_ws_floats(&h, nprob), "conv2d_bwd_weight: workspace too small for up = 2");
    const float* up_x[WG_MAXP];
    for (int i = 0; i < nprob; ++i) {
      SRX_REQUIRE(xs[i], "conv2d_bwd_weight: null tensor in problem %d", i);
      up_x[i] = ws + (size_t)i * tmp;
      if (int rc = srx_upsample_nearest2x_fwd(xs[i], ws + (size_t)i * tmp, d->N, d->H, d->W, d->Cin_s, stream)) return rc;
    }
    return wgrad_multi_impl(&h, nprob, per_out, up_x, dys, dws, accumulate, dbs, out_scales, nullptr, nullptr, 0,
                            ws + nprob * tmp, ws_floats - nprob * tmp, stream);
  }
  const int nout = nprob / per_out;
  bool any_db = false;
  for (int i = 0; i < nprob; ++i) SRX_REQUIRE(xs[i] && dys[i], "conv2d_bwd_weight: null tensor in problem %d", i);
  for (int o = 0; o < nout; ++o) {
    SRX_REQUIRE(dws[o] && (!dws_hi || dws_hi[o]), "conv2d_bwd_weight: null gradient pointer for output %d", o);
    any_db |= (dbs && dbs[o]) || (dbs_hi && dbs_hi[o]);
  }
  hipStream_t st = srx_stream(stream);
  if (srx_thin_wgrad_applicable(d)) {  // 3-channel layers: their own kernel, one problem at a time
    for (int o = 0; out_scales && o < nout; ++o)
      if (out_scales[o] != 1.f) SRX_FAIL(SRX_E_UNSUPPORTED, "conv2d_bwd_weight: output scales on a 3-channel layer");
    const size_t thin_ws = srx_thin_wgrad_ws_floats(d);
    const int64_t m = (int64_t)d->N * d->H * d->W;  // thin layers are stride 1, same size
    for (int i = 0; i < nprob; ++i) {
      const int o = i / per_out;
      const int acc = accumulate || (i % per_out) > 0;
      if (int rc = srx_thin_wgrad(d, xs[i], dys[i], dws[o], acc, ws, ws_floats, st)) return rc;
      if (!(dbs && dbs[o])) continue;
      SRX_REQUIRE(ws_floats >= thin_ws + srx_colsum_ws_floats(m, d->Cout), "conv2d_bwd_weight: workspace too small");
      if (int rc = srx_colsum(dys[i], dbs[o], m, d->Cout, d->Cout_s, acc, ws + thin_ws, ws_floats - thin_ws, stream)) return rc;
    }
    return SRX_OK;
  }
  const Geo g = fwd_geo(d);
  WMulti mp{};
  WArgs& a = mp.a;
  a.in = xs[0]; a.dy = dys[0]; a.slab = ws;
  a.N = d->N; a.Hi = d->H; a.Wi = d->W; a.Ci = d->Cin_s;
  a.Hm = g.Ho; a.Wm = g.Wo; a.HmWm = g.Ho * g.Wo; a.M = d->N * g.Ho * g.Wo;
  a.inv_HmWm = 1.0f / (float)a.HmWm; a.inv_Wm = 1.0f / (float)g.Wo;
  a.in_stride = d->stride; a.nth = d->KH; a.ntw = d->KW; a.dh0 = -d->pad; a.dw0 = -d->pad;
  a.Ck = g.Ck; a.K = g.K;
  a.Kw = (int)srx_roundup(g.K, 64);
  a.Cnw = (int)srx_roundup(d->Cout, 64);
  a.Cd = d->Cout_s;
  a.dy_shuffle = g.cps;
  a.Cdv = bwd_ck(d);
  a.ktiles = a.Kw / 64;
  const size_t dyb = (size_t)d->N * g.Ho * g.Wo * (d->shuffle ? 4 : 1) * d->Cout_s * sizeof(float);
  SRX_REQUIRE(dyb < 0xfffffff0ull, "conv2d_bwd_weight: gradient tensor above 4 GiB; tile the image");
  a.in_bytes = (unsigned)((size_t)d->N * d->H * d->W * d->Cin_s * sizeof(float));
  a.dy_bytes = (unsigned)dyb;
  {
    const int s = d->stride, s_r = 32 / a.Wm, s_n = s_r / a.Hm;
    a.s_c = 32 % a.Wm; a.s_rm = s_r % a.Hm;
    a.dX0 = ((s_n * a.Hi + a.s_rm * s) * a.Wi + a.s_c * s) * a.Ci;
    a.dX1 = (s * a.Wi - a.Wm * s) * a.Ci;
    a.dX2 = (a.Hi - a.Hm * s) * a.Wi * a.Ci;
    if (a.dy_shuffle) {
      a.dD0 = ((s_n * 2 * a.Hm + 2 * a.s_rm) * (2 * a.Wm) + 2 * a.s_c) * a.Cd;
      a.dD1 = 2 * a.Wm * a.Cd;
    } else {
      a.dD0 = 32 * a.Cd;
      a.dD1 = 0;
    }
  }
  const int ntiles = a.Cnw / 64;
  const int64_t tiles = (int64_t)a.ktiles * ntiles;
  const bool rows_kernel = wgrad_rows_ok(d);
  const int nsplit = rows_kernel ? wgrad_rows_nsplit(d, nprob) : wgrad_nsplit(a.M, tiles, nprob, a.Cnw, a.Kw, d->precision);
  a.rows_per_split = rows_kernel ? (int)srx_cdiv(d->N * d->H, nsplit) * d->W : (int)srx_roundup(srx_cdiv(a.M, nsplit), 32);
  a.nsplit = nsplit;
  a.nprob = nprob;
  const size_t nslabs = (size_t)nsplit * nprob;
  const size_t need = nslabs * a.Cnw * a.Kw + (any_db ? nslabs * a.Cnw : 0);
  if (need > ws_floats) SRX_FAIL(SRX_E_WORKSPACE, "conv2d_bwd_weight: workspace %zu < %zu floats", ws_floats, need);
  a.bslab = any_db ? ws + nslabs * a.Cnw * a.Kw : nullptr;
  WReduce outs{};
  for (int i = 0; i < nprob; ++i) { mp.x[i] = xs[i]; mp.dy[i] = dys[i]; }
  for (int o = 0; o < nout; ++o) {
    outs.dw[o] = dws[o]; outs.db[o] = dbs ? dbs[o] : nullptr; outs.scale[o] = out_scales ? out_scales[o] : 1.f;
    SRX_REQUIRE(outs.scale[o] == outs.scale[o] && outs.scale[o] - outs.scale[o] == 0.f,
                "conv2d_bwd_weight_multi_scaled: output scale %d is not finite (out_scales is a HOST array of nprob / per_out floats)", o);
    outs.dw_hi[o] = dws_hi ? dws_hi[o] : nullptr; outs.db_hi[o] = dbs_hi ? dbs_hi[o] : nullptr;
  }
  outs.rows_lo = dws_hi ? d->Cout / 2 : 0;
  outs.cin_lo = cin_lo;
  dim3 grid((unsigned)(tiles * nprob * nsplit));
  const double wfl = 2.0 * a.M * d->Cout * a.K * nprob;
  char nm[112];
  if (rows_kernel) {
    grid = dim3((unsigned)((g.Ck / 32) * nprob * nsplit));
    if (srx_prof_on()) snprintf(nm, sizeof(nm), "wgrad_rows_bf16_kernel<%d> MxNxK=%dx%dx%d x%d", d->W, a.M, d->Cout, a.K, nprob);
    if (d->W == 32) SRX_LAUNCH_PROF(nm, wfl, wgrad_rows_bf16_kernel<32>, grid, dim3(256), 0, st, mp);
    else SRX_LAUNCH_PROF(nm, wfl, wgrad_rows_bf16_kernel<16>, grid, dim3(256), 0, st, mp);
    SRX_CHECK_LAUNCH("wgrad_rows_bf16_kernel");
  } else {
  if (srx_prof_on())
    snprintf(nm, sizeof(nm), "wgrad_kernel<%d> MxNxK=%dx%dx%d x%d", d->precision ? 1 : 0, a.M, d->Cout, a.K, nprob);
  if (d->precision) SRX_LAUNCH_PROF(nm, wfl, wgrad_kernel<1>, grid, dim3(256), 0, st, mp);
  else if (srx_dev().no_wgrad_dma) SRX_LAUNCH_PROF(nm, wfl, wgrad_kernel<0>, grid, dim3(256), 0, st, mp);
  else if (g.Ck % 64 == 0 && a.Cdv % 64 == 0 && (!a.dy_shuffle || a.dy_shuffle % 64 == 0)) {
    // LIN: offsets linear in the row, validity from a per-workgroup bit table (see wgrad_dma_kernel)
    const bool lin = !srx_dev().no_wgrad_lin && a.in_stride == 1 && a.Hi == a.Hm && a.Wi == a.Wm && !a.dy_shuffle && a.K % 64 == 0 &&
                     a.rows_per_split / 32 + 3 <= WG_MASKW && a.in_bytes < 0xfff00000u && a.dy_bytes < 0xfff00000u;
    if (lin) SRX_LAUNCH_PROF(nm, wfl, (wgrad_dma_kernel<true, true>), grid, dim3(256), 0, st, mp);
    else SRX_LAUNCH_PROF(nm, wfl, wgrad_dma_kernel<true>, grid, dim3(256), 0, st, mp);
  }
  else SRX_LAUNCH_PROF(nm, wfl, wgrad_dma_kernel<false>, grid, dim3(256), 0, st, mp);
  SRX_CHECK_LAUNCH("wgrad_kernel");
  }
  if (srx_dev().old_wgrad_reduce || (size_t)g.K * sizeof(float) > 48 * 1024) {
    const int64_t n = (int64_t)d->Cout * g.K;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)srx_cdiv(n, 256), (unsigned)nout), dim3(256), 0, st, ws,
                       nsplit * per_out, a.Cnw, a.Kw, g.K, g.Ck, d->Cout, d->Cin, d->KH, d->KW, g.cps, outs, accumulate,
                       a.bslab);
  } else {
    hipLaunchKernelGGL(wgrad_reduce_rows_kernel, dim3((unsigned)d->Cout, (unsigned)nout), dim3(256), (size_t)g.K * sizeof(float), st, ws,
                       nsplit * per_out, a.Cnw, a.Kw, g.K, g.Ck, d->Cout, d->Cin, d->KH, d->KW, g.cps, outs, accumulate,
                       a.bslab);
  }
  SRX_CHECK_LAUNCH("wgrad_reduce_kernel");
  return SRX_OK;
}

extern "C" int srx_conv2d_bwd_weight(const srx_conv2d_t* d, const float* x, const float* dy, float* dw, int accumulate,
                                     float* db, float* ws, size_t ws_floats, void* stream) {
  SRX_REQUIRE(x && dy && dw && ws, "conv2d_bwd_weight: null pointer");
  return srx_conv2d_bwd_weight_multi(d, 1, 1, &x, &dy, &dw, accumulate, db ? &db : nullptr, ws, ws_floats, stream);
}

// ------------------------------------------------------------------------------------------------------------------
// bf16 STORAGE of activations on the training path (round 5; round-4 review, item 2): 3x3 / stride 1 / pad 1 layers whose
// input AND output tensors are bf16 NHWC.  The arithmetic is the bf16-product recipe of precision = 1 to the letter -- every
// operand of such a layer was rounded to bf16 when it was staged; here the PRODUCER rounds it once instead of every consumer
// -- so the oracle and its tolerances do not change; what changes is the traffic (half the bytes, no fp32 -> bf16 conversion
// in the loader: 18.8 VALU instructions per MFMA on the 64-column layers, profiles/r04_pmc_esrgan.txt).  The kernel is
// gconv_kernel<..., PR = 2>: the bf16 tensors are described as fp32 tensors of HALF the channel count, so a 128-byte k-chunk
// is 64 channels of one tap and every gather offset is the fp32 code's; a 16-byte quad is one v_mfma_f32_32x32x16_bf16 operand.
// Used by the frozen VGG19 stack of the perceptual loss under autocast (esrgan/trainer.py:461-467, srgan/loss.py:52-53).
// ------------------------------------------------------------------------------------------------------------------
namespace {

bool bf16s_shape_ok(const srx_conv2d_t* d) {
  return d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && !d->shuffle && d->up != 2 && d->precision == 1 &&
         d->Cin % 64 == 0 && d->Cout % 64 == 0 && d->Cin_s == d->Cin && d->Cout_s == d->Cout &&
         (int64_t)d->N * d->H * d->W < (1 << 24) && (int64_t)d->N * d->H * d->W * std::max(d->Cin, d->Cout) < (1LL << 30);
}

// wf[co][(kh, kw, ci)] = bf16(w[co][ci][kh][kw]);  wb[ci][(kh, kw, co)] = bf16(w[co][ci][2 - kh][2 - kw])
__global__ void bf16s_pack_kernel(const float* __restrict__ w, __bf16* __restrict__ wf, __bf16* __restrict__ wb, int Cout, int Cin) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)Cout * Cin * 9) return;
  const int ci = (int)(idx % Cin);
  const int t = (int)((idx / Cin) % 9);
  const int co = (int)(idx / ((int64_t)Cin * 9));
  const __bf16 v = (__bf16)w[((size_t)co * Cin + ci) * 9 + t];
  wf[idx] = v;  // ((co * 9 + t) * Cin + ci)
  if (wb) wb[((size_t)ci * 9 + (8 - t)) * Cout + co] = v;
}

Plan bf16s_plan(const srx_conv2d_t* d, int which) {
  const int cc = which ? d->Cout : d->Cin, cn = which ? d->Cin : d->Cout;
  return make_plan(d->N * d->H * d->W, cn, 9 * cc / 64, true, true, false);
}

int bf16s_run(const srx_conv2d_t* d, int which, const void* in16, const void* wpk16, const float* bias, int relu,
              const void* mask16, void* out, int out_is_bf16, float* ws, size_t ws_floats, void* stream) {
  if (int rc = check_desc(d)) return rc;
  SRX_REQUIRE(in16 && wpk16 && out && in16 != out, "conv3x3_bf16s: null pointer / in place");
  if (!bf16s_shape_ok(d))
    SRX_FAIL(SRX_E_UNSUPPORTED, "conv3x3_bf16s: 3x3 / stride 1 / pad 1 layers with precision = 1 and channel counts that are multiples of 64 only");
  const int cc = which ? d->Cout : d->Cin, cn = which ? d->Cin : d->Cout;
  GArgs a{};
  a.in = static_cast<const float*>(in16); a.w = static_cast<const float*>(wpk16); a.bias = bias;
  set_mgrid(a, d->N, d->H, d->W);
  a.Hi = d->H; a.Wi = d->W; a.Ci = cc / 2;  // (floats = pairs of bf16 channels)
  a.in_stride = 1; a.nth = 3; a.ntw = 3; a.dh0 = -1; a.dw0 = -1;
  a.Ck = cc / 2; a.K = 9 * cc / 2; a.Kp = a.K;
  a.Cn = cn; a.Cs = cn; a.Ho = d->H; a.Wo = d->W; a.Co = cn;
  a.out_stride = 1; a.linear_out = 1;
  a.act = relu ? SRX_ACT_RELU : SRX_ACT_NONE;
  a.slope = relu ? 0.f : 1.f;
  a.out = static_cast<float*>(out);
  a.out16 = out_is_bf16 ? 1 : 0;
  a.oscale = 1.f; a.add_ld = cn; a.add_hi = 0x7fffffff; a.ascale = 1.f;
  if (mask16) { a.mask = static_cast<const float*>(mask16); a.mask16 = 1; a.mask_slope = 0.f; a.mask_lo = 0; a.mask_hi = cn; }
  a.in_bytes = (size_t)d->N * d->H * d->W * cc * 2;
  a.w_bytes = (unsigned)((size_t)cn * 9 * cc * 2);
  a.in_margin = 4u * (unsigned)((a.Wi + 1) * a.Ci);
  return run_gconv(a, bf16s_plan(d, which), ws, ws_floats, srx_stream(stream), 2);
}

}  // namespace

extern "C" int srx_conv3x3_bf16s_applicable(const srx_conv2d_t* d) { return d && check_desc(d) == SRX_OK && bf16s_shape_ok(d) ? 1 : 0; }

// bytes of ONE packed copy (forward or data gradient): Cout x 9 x Cin bf16 values
extern "C" size_t srx_conv3x3_bf16s_packed_bytes(const srx_conv2d_t* d) {
  return (d && bf16s_shape_ok(d)) ? (size_t)d->Cout * 9 * d->Cin * 2 : 0;
}

extern "C" int srx_conv3x3_bf16s_pack(const srx_conv2d_t* d, const float* w, void* wpk_fwd, void* wpk_bwd, void* stream) {
  if (int rc = check_desc(d)) return rc;
  SRX_REQUIRE(w && wpk_fwd, "conv3x3_bf16s_pack: null pointer");
  if (!bf16s_shape_ok(d)) SRX_FAIL(SRX_E_UNSUPPORTED, "conv3x3_bf16s_pack: not a bf16-storage layer");
  const int64_t n = (int64_t)d->Cout * d->Cin * 9;
  hipLaunchKernelGGL(bf16s_pack_kernel, dim3((unsigned)srx_cdiv(n, 256)), dim3(256), 0, srx_stream(stream), w,
                     static_cast<__bf16*>(wpk_fwd), static_cast<__bf16*>(wpk_bwd), d->Cout, d->Cin);
  SRX_CHECK_LAUNCH("bf16s_pack_kernel");
  return SRX_OK;
}

// workspace (floats) of the forward (which = 0) / data gradient (which = 1): K-split partial tiles of the last round
extern "C" size_t srx_conv3x3_bf16s_ws_floats(const srx_conv2d_t* d, int which) {
  if (!d || check_desc(d) != SRX_OK || !bf16s_shape_ok(d)) return 0;
  return plan_ws_floats(bf16s_plan(d, which));
}

// the 3 -> 64 first layer in front of such a stack (srx_conv2d_fwd's first3x3 kernel, precision = 1) writing a bf16 tensor:
// x fp32 [N][H][W][4], wpk: the layer's ordinary forward pack, y bf16 [N][H][W][64]
extern "C" int srx_conv2d_fwd_first3_to_bf16(const srx_conv2d_t* d, const float* x, const float* wpk_fwd, const float* bias, void* y,
                                             void* stream) {
  if (int rc = check_desc(d)) return rc;
  SRX_REQUIRE(x && wpk_fwd && y, "conv2d_fwd_first3_to_bf16: null pointer");
  if (!srx_first3_fwd_applicable(d) || d->precision != 1 || d->act == SRX_ACT_PRELU)
    SRX_FAIL(SRX_E_UNSUPPORTED, "conv2d_fwd_first3_to_bf16: 3x3 / stride 1 / pad 1 layers from <= 4 to 64 channels with precision = 1 only");
  return srx_first3_fwd(d, x, wpk_fwd, fwd_geo(d).Kp, bias, static_cast<float*>(y), srx_stream(stream), 1);
}

// y = [relu](conv(x) + bias): x bf16 [N][H][W][Cin]; y bf16 or fp32 [N][H][W][Cout]
extern "C" int srx_conv3x3_bf16s_fwd(const srx_conv2d_t* d, const void* x, const void* wpk_fwd, const float* bias, int relu,
                                     void* y, int y_is_bf16, float* ws, size_t ws_floats, void* stream) {
  return bf16s_run(d, 0, x, wpk_fwd, bias, relu, nullptr, y, y_is_bf16, ws, ws_floats, stream);
}

// dx = conv^T(dy) [* (relu_out > 0)]: dy bf16 [N][H][W][Cout]; relu_out: the bf16 OUTPUT of the ReLU that produced this layer's
// input (null: none) -- that activation's backward on the way out; dx bf16 or fp32 [N][H][W][Cin]
extern "C" int srx_conv3x3_bf16s_bwd_data(const srx_conv2d_t* d, const void* dy, const void* wpk_bwd, const void* relu_out,
                                          void* dx, int dx_is_bf16, float* ws, size_t ws_floats, void* stream) {
  return bf16s_run(d, 1, dy, wpk_bwd, nullptr, 0, relu_out, dx, dx_is_bf16, ws, ws_floats, stream);
}
