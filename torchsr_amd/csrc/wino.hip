// 3x3 / stride 1 / pad 1 convolutions of wide layers as Winograd F(2x2, 3x3) on the fp32 matrix cores (round 5).
//
// The direct gather-GEMM (gconv.hip) runs the VGG19 layers of the perceptual loss (srgan/loss.py:30-54: 344 of the SRGAN
// step's 692 GFLOP) at 0.95 of what v_mfma_f32_32x32x2_f32 gives at the clock the part holds under that load -- the only
// way to make them faster is to multiply less.  Winograd's minimal filtering computes a 2x2 output tile from a 4x4 input
// tile with 16 multiplications per (input channel, output channel) pair instead of 36:
//
//     Y = A^T [ (G g G^T) (.) (B^T d B) ] A
//     B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]   G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]   A^T = [1 1 1 0; 0 1 -1 -1]
//
// i.e. sixteen independent GEMMs  M_xi[tile][co] = sum_ci V_xi[tile][ci] U_xi[ci][co]  (xi = the 16 positions of the
// transformed 4x4 tile) with 2.25x fewer multiply-adds than the direct form, all in fp32: the result differs from the
// direct fp32 convolution by rounding only (measured 2-3x its own rounding error against fp64, i.e. ~5e-7 of the tensor's
// scale: tools/experiments/wino_error.py), far inside the 1e-3 the parity tests allow.
//
// One workgroup = 32 tiles (128 output pixels) x BN output channels x all 16 xi, 512 threads:
//   * waves 0..3 gather the 4x4 input patches of the 32 tiles for a chunk of 32 input channels (16 pixels x 16 bytes per
//     thread, padding pixels through an out-of-range descriptor offset), apply B^T d B in registers (32 float4 adds) and
//     write the sixteen V_xi rows of the chunk to a two-stage LDS ring (64 KB per stage, XOR-swizzled 128-byte rows);
//   * every wave owns two xi and multiplies them for all 32 tiles and BN channels: U fragments come STRAIGHT from global memory
//     into MFMA operand registers (each xi belongs to exactly one wave, so nothing is shared through LDS; srx_wino_pack lays
//     U out so that a wave's load is 1 KB contiguous), V fragments are one ds_read_b128 per eight MFMAs;
//   * epilogue: the accumulators (U as the A operand, so a lane holds four consecutive channels of one tile) go to LDS as
//     M[xi][tile][channel], every thread takes one (tile, channel quad), applies A^T M A, adds the bias, applies ReLU or the
//     ReLU mask of the layer below (data gradients: the fold of srx_conv2d_bwd_data_act) and stores four pixels x 16 bytes.
// The data gradient of such a layer IS such a layer (channels swapped, taps flipped): srx_wino_pack(..., transpose = 1).
// Small layers split the input channels over several workgroups (partial outputs + a streaming fix-up); layers that cut into a
// non-integer number of rounds of the chip's CUs run whole tiles for the full rounds and cut only the tiles of the last round along
// the input channels (tile-local partial outputs + wino_tail_fixup_kernel); the planner (wino_plan) picks BN (64 or 32) and the split
// from a cost model.  Inference adds LeakyReLU / PReLU, a skip addend and nn.PixelShuffle(2) in the store (srx_wino_fwd_act).
#include "srx_common.h"
#include <algorithm>
#include <mutex>

namespace {

constexpr int WT = 32;       // tiles per workgroup
constexpr int WKC = 32;      // input channels per chunk
constexpr int STAGE_BYTES = 16 * WT * WKC * 4;  // 65536: V of one chunk
constexpr int WINO_LDS = 2 * STAGE_BYTES;       // the epilogue's M[16][32][BN <= 64] reuses both stages

struct WinoArgs {
  const float* in; const float* upk; const float* bias; const float* mask; float* out; float* part;
  float* stats;  // training-mode BatchNorm partials of the pre-activation output: [tile block][Cout][2] = (sum, sum of squares), or null
  int N, H, W, Cin, Cout;
  int TH, TW, T, tblocks, ncb, nch, zsplit;
  int relu;
  // inference forms (FoldedConv, round 5): `lrelu` != 0: LeakyReLU / single-parameter PReLU with slope `slope` instead of ReLU;
  // `add`: a tensor laid out like `out`, added after the activation (the skip of a residual block, srgan/residual.py:86-91)
  int lrelu; float slope; const float* add;
  int shuffle;  // != 0: nn.PixelShuffle(2) in the store (forward only): the value is the channel count per sub-pixel, Cout / 4
  // Tail split (round 5): work items [0, full) are whole (tile block, channel block) tiles; the tiles of the last, partly filled
  // round of the chip -- item full + t is part t % tsplit of tile full + t / tsplit -- take 1 / tsplit of the input channels each and
  // write raw partial outputs to tpart[t][32 tiles][4 pixels][BN]; wino_tail_fixup_kernel finishes those tiles only.
  // (Without a tail: full = every item.)  2.25 rounds of whole tiles cost 3 rounds; 2 rounds + a quarter-length third cost ~2.4.
  int full, tsplit; float* tpart;
  unsigned in_bytes, upk_bytes;
  size_t out_elems;  // N * H * W * Cout (stride between the partial outputs of two splits)
};

template <int BN>
__global__ __launch_bounds__(512) void wino_kernel(const WinoArgs a) {
  constexpr int NJ = BN / 32;  // channel tiles per workgroup
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = srx_uniform(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  // Work item = (split, channel block, tile block), tile blocks fastest.  Workgroups are dealt round-robin over the 8 XCDs; this
  // remap gives every XCD a contiguous run of items (as gconv.hip's weight-gradient kernels do) -- the tile blocks of a run share
  // their U block and their halo rows in ONE L2 (a speed matter only: with the plain order every XCD pulled all of U through the
  // fabric, 142 MB of HBM reads for a 512 -> 512 layer whose operands are 22 MB; PMC, round 5)
  int b = (int)blockIdx.x;
  const bool tailw = b >= a.full;  // (workgroup-uniform) a part of a tail tile
  int tb, cb, z, zs;
  if (!tailw) {
    const int W = a.full, xcd = b & 7, slot = b >> 3, qq = W >> 3, rr = W & 7;
    b = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + slot;
    tb = b % a.tblocks; b /= a.tblocks;
    cb = b % a.ncb;
    z = b / a.ncb; zs = a.zsplit;
  } else {
    const int t = b - a.full, tile = a.full + t / a.tsplit;
    z = t % a.tsplit; zs = a.tsplit;
    tb = tile % a.tblocks; cb = tile / a.tblocks;
  }
  // this split's chunks of input channels: [kc0, kc1)
  const int kc0 = srx_uniform((int)((long long)z * a.nch / zs)), kc1 = srx_uniform((int)((long long)(z + 1) * a.nch / zs));
  const __amdgpu_buffer_rsrc_t rin = srx_rsrc(a.in, a.in_bytes);
  const __amdgpu_buffer_rsrc_t ru = srx_rsrc(a.upk, a.upk_bytes);

  // ---- loader state (waves 0..3): tile tl of the block, channel quad q of the chunk; byte offsets of the 16 patch pixels
  const bool loader = wave < 4;
  const int tl = tid >> 3, q = tid & 7;  // (tid < 256 for loaders)
  unsigned poff[16];
  {
    const int t = tb * WT + (tl & 31);
    const bool tok = loader && t < a.T;
    const int tw = t % a.TW, r = t / a.TW, th = r % a.TH, n = r / a.TH;
    const int ih0 = 2 * th - 1, iw0 = 2 * tw - 1;
    const int base = ((n * a.H + ih0) * a.W + iw0) * a.Cin + 4 * q;  // element offset of patch pixel (0, 0) (may be "negative": never used then)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool rok = tok && (unsigned)(ih0 + i) < (unsigned)a.H;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool ok = rok && (unsigned)(iw0 + j) < (unsigned)a.W;
        poff[4 * i + j] = ok ? 4u * (unsigned)(base + (i * a.W + j) * a.Cin) : 0xffffffffu;  // out of range reads 0
      }
    }
  }
  // ---- multiplier state: this wave's two xi; U fragments of a sub-step s: [xi][j] one float4 per lane
  //      packed U: (((xi * (Cout / 32) + jg) * nch + kc) * 4 + s) * 64 + lane, in float4 units
  const int xi0 = 2 * wave;
  const int cot = a.Cout >> 5;
  unsigned uoff[2][NJ];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      uoff[x][j] = 16u * (unsigned)((((xi0 + x) * cot + cb * NJ + j) * a.nch) * 256 + lane);

  f32x16 acc[2][NJ];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][j][r] = 0.f;

  f32x4 pd[16];        // the patch of the chunk being staged
  f32x4 uf[4][2][NJ];  // U fragments, one slot per sub-step
  // (measured: loads that are simply skipped past the last chunk beat gconv.hip's always-issued, out-of-range-pointed ones
  // here -- 1360 vs 1404 us over VGG19's forward -- the chunk loop is short and the last chunk's dummy stage costs more)
  auto load_patch = [&](int kc) {
#pragma unroll
    for (int p = 0; p < 16; ++p) pd[p] = srx_bload(rin, poff[p], (unsigned)srx_uniform(kc * (WKC * 4)));
  };
  auto load_u = [&](int kc, int s) {
    const unsigned so = (unsigned)srx_uniform((kc * 4 + s) * 1024);
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int j = 0; j < NJ; ++j) uf[s][x][j] = srx_bload(ru, uoff[x][j], so);
  };
  // B^T d B of this thread's patch, the sixteen results to row (xi, tl) of stage `st` at quad q (XOR swizzle as in gconv.hip)
  auto stage_patch = [&](int st) {
    // in place, so that no second copy of the patch is ever live: rows first (d B: every input row on its own), then the
    // columns (B^T .), each column's four results written as soon as they exist
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 r0 = pd[4 * i + 0] - pd[4 * i + 2], r1 = pd[4 * i + 1] + pd[4 * i + 2], r2 = pd[4 * i + 2] - pd[4 * i + 1],
                  r3 = pd[4 * i + 1] - pd[4 * i + 3];
      pd[4 * i + 0] = r0; pd[4 * i + 1] = r1; pd[4 * i + 2] = r2; pd[4 * i + 3] = r3;
    }
    float* base = reinterpret_cast<float*>(smem + st * STAGE_BYTES) + tl * WKC + ((q ^ ((tl >> 1) & 7)) << 2);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 v0 = pd[0 + j] - pd[8 + j], v1 = pd[4 + j] + pd[8 + j], v2 = pd[8 + j] - pd[4 + j], v3 = pd[4 + j] - pd[12 + j];
      *reinterpret_cast<f32x4*>(base + (0 + j) * (WT * WKC)) = v0;   // xi = 4 i + j
      *reinterpret_cast<f32x4*>(base + (4 + j) * (WT * WKC)) = v1;
      *reinterpret_cast<f32x4*>(base + (8 + j) * (WT * WKC)) = v2;
      *reinterpret_cast<f32x4*>(base + (12 + j) * (WT * WKC)) = v3;
    }
  };
  // sub-step s of the chunk in stage st: quads 2s (lanes 0..31) and 2s + 1 (lanes 32..63) of this wave's two xi
  const int vrow = l31 * WKC, vsw = (l31 >> 1) & 7;
  auto mma = [&](int st, int s) {
    const float* sv = reinterpret_cast<const float*>(smem + st * STAGE_BYTES) + vrow + (((2 * s + h) ^ vsw) << 2);
    f32x4 vf[2];
#pragma unroll
    for (int x = 0; x < 2; ++x) vf[x] = *reinterpret_cast<const f32x4*>(sv + (xi0 + x) * (WT * WKC));
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[x][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(uf[s][x][j][e], vf[x][e], acc[x][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };

  // ---- prologue: first chunk staged, its U fragments requested
  if (loader) load_patch(kc0);
#pragma unroll
  for (int s = 0; s < 4; ++s) load_u(kc0, s);
  if (loader) stage_patch(0);
  __syncthreads();
  // ---- chunk loop: the patch of chunk c + 1 is requested at the top, transformed and written between the second and third
  // sub-step (the partner wave on the SIMD keeps the matrix pipe busy meanwhile); a sub-step's U slot is refilled for
  // chunk c + 1 as soon as its MFMAs are issued
  int st = 0;
  for (int kc = kc0; kc < kc1; ++kc) {
    const bool more = kc + 1 < kc1;  // (workgroup-uniform)
    if (loader && more) load_patch(kc + 1);
    mma(st, 0);
    if (more) load_u(kc + 1, 0);
    mma(st, 1);
    if (more) load_u(kc + 1, 1);
    if (loader && more) stage_patch(st ^ 1);
    mma(st, 2);
    if (more) load_u(kc + 1, 2);
    mma(st, 3);
    if (more) load_u(kc + 1, 3);
    __syncthreads();
    st ^= 1;
  }

  // ---- epilogue: M[xi][tile][channel] through LDS (16-byte chunk c of row (xi, t) at chunk c ^ (t & 15): the 32 lanes of
  // a store hit 16 different chunk columns), then A^T M A per (tile, channel quad)
  constexpr int MROW = BN;  // floats per (xi, tile) row
  float* sM = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        // accumulator rows (channels) 8 g + 4 h + 0..3 of channel tile j, column (tile) l31
        const int c = (j * 32 + 8 * g + 4 * h) >> 2;  // 16-byte chunk of the row
        const f32x4 v = {acc[x][j][4 * g + 0], acc[x][j][4 * g + 1], acc[x][j][4 * g + 2], acc[x][j][4 * g + 3]};
        *reinterpret_cast<f32x4*>(sM + ((xi0 + x) * WT + l31) * MROW + ((c ^ (l31 & (BN / 4 - 1))) << 2)) = v;
      }
  __syncthreads();
  constexpr int CQ = BN / 4;            // channel quads per row
  constexpr int ITEMS = WT * CQ;        // 512 (BN = 64) or 256 (BN = 32)
  if (tid >= ITEMS) return;             // (BN = 32: waves 4..7 are done; a terminated wave does not hold up the barrier below)
  const int et = tid / CQ, cq = tid % CQ;
  const int t = tb * WT + et;
  const bool tvalid = t < a.T;
  if (!tvalid && !a.stats) return;
  f32x4 m[16];
#pragma unroll
  for (int x = 0; x < 16; ++x) m[x] = *reinterpret_cast<const f32x4*>(sM + (x * WT + et) * MROW + ((cq ^ (et & (CQ - 1))) << 2));
  f32x4 s0[4], s1[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    s0[j] = m[0 + j] + m[4 + j] + m[8 + j];
    s1[j] = m[4 + j] - m[8 + j] - m[12 + j];
  }
  f32x4 y[4];
  y[0] = s0[0] + s0[1] + s0[2];
  y[1] = s0[1] - s0[2] - s0[3];
  y[2] = s1[0] + s1[1] + s1[2];
  y[3] = s1[1] - s1[2] - s1[3];
  const int tw = t % a.TW, r = t / a.TW, th = r % a.TH, n = r / a.TH;
  const int co = cb * BN + 4 * cq;
  const size_t p00 = (((size_t)n * a.H + 2 * th) * a.W + 2 * tw) * a.Cout + co;
  size_t offs[4] = {p00, p00 + a.Cout, p00 + (size_t)a.W * a.Cout, p00 + (size_t)a.W * a.Cout + a.Cout};
  if (a.shuffle) {  // GEMM column co = (sub-pixel ij, channel cc): pixel (y, x) of the conv lands at (2y + ij / 2, 2x + ij % 2), channel cc
    const int ij = co / a.shuffle, cc = co - ij * a.shuffle;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const size_t oy = 2 * (size_t)(2 * th + (p >> 1)) + (ij >> 1), ox = 2 * (size_t)(2 * tw + (p & 1)) + (ij & 1);
      offs[p] = (((size_t)n * 2 * a.H + oy) * (2 * (size_t)a.W) + ox) * a.shuffle + cc;
    }
  }
  if (tailw) {  // a part of a tail tile: raw partial output, tile-local layout; finished by wino_tail_fixup_kernel
    float* o = a.tpart + (size_t)((int)blockIdx.x - a.full) * (WT * 4 * BN) + (size_t)(et * 4) * BN + 4 * cq;
#pragma unroll
    for (int p = 0; p < 4; ++p) *reinterpret_cast<f32x4*>(o + p * BN) = y[p];
    return;
  }
  if (a.part) {  // one of several splits of the input channels: the raw partial output; bias / activation in the fix-up pass
    float* o = a.part + (size_t)z * a.out_elems;
#pragma unroll
    for (int p = 0; p < 4; ++p) *reinterpret_cast<f32x4*>(o + offs[p]) = y[p];
    return;
  }
  f32x4 mk[4], ad[4];
  if (a.mask && tvalid) {  // (all loads before the first store)
#pragma unroll
    for (int p = 0; p < 4; ++p) mk[p] = *reinterpret_cast<const f32x4*>(a.mask + offs[p]);
  }
  if (a.add && tvalid) {
#pragma unroll
    for (int p = 0; p < 4; ++p) ad[p] = *reinterpret_cast<const f32x4*>(a.add + offs[p]);
  }
  if (a.bias) {
    f32x4 bv;
    if (a.shuffle) {  // (the bias is in the conv's own channel order)
      const int ij = co / a.shuffle, cc = co - ij * a.shuffle;
#pragma unroll
      for (int e = 0; e < 4; ++e) bv[e] = a.bias[(cc + e) * 4 + ij];
    } else {
      bv = *reinterpret_cast<const f32x4*>(a.bias + co);
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) y[p] += bv;
  }
  if (a.stats) {
    // per-channel sum / sum of squares of this tile block's 128 pixels (what gconv's epilogue writes per row tile): the
    // lanes of a wave that share a channel quad first (fixed butterfly), then the waves in order through LDS
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    if (tvalid) {
#pragma unroll
      for (int p = 0; p < 4; ++p) { s1 += y[p]; s2 += y[p] * y[p]; }
    }
#pragma unroll
    for (int o = CQ; o < 64; o <<= 1)
#pragma unroll
      for (int e = 0; e < 4; ++e) { s1[e] += __shfl_xor(s1[e], o, 64); s2[e] += __shfl_xor(s2[e], o, 64); }
    constexpr int NW = ITEMS / 64;  // waves that hold items
    f32x4* red = reinterpret_cast<f32x4*>(smem + WINO_LDS);  // [NW][CQ][2]
    if (lane < CQ) { red[(wave * CQ + lane) * 2 + 0] = s1; red[(wave * CQ + lane) * 2 + 1] = s2; }
    __syncthreads();
    if (tid < CQ) {
      f32x4 t1 = red[tid * 2], t2 = red[tid * 2 + 1];
#pragma unroll
      for (int wv = 1; wv < NW; ++wv) { t1 += red[(wv * CQ + tid) * 2]; t2 += red[(wv * CQ + tid) * 2 + 1]; }
      float* o = a.stats + ((size_t)tb * a.Cout + cb * BN + 4 * tid) * 2;
#pragma unroll
      for (int e = 0; e < 4; ++e) { o[2 * e] = t1[e]; o[2 * e + 1] = t2[e]; }
    }
    if (!tvalid) return;
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = y[p][e];
      if (a.relu) v = fmaxf(v, 0.f);
      if (a.lrelu) v = v > 0.f ? v : v * a.slope;
      if (a.mask) v = mk[p][e] > 0.f ? v : 0.f;
      if (a.add) v += ad[p][e];
      y[p][e] = v;
    }
    *reinterpret_cast<f32x4*>(a.out + offs[p]) = y[p];
  }
}

// out = act(sum_z part[z] + bias) [masked]: finishes a layer whose input channels were split over several workgroups
__global__ __launch_bounds__(256) void wino_fixup_kernel(const float* __restrict__ part, int zsplit, size_t n4, size_t stride,
                                                         const float* __restrict__ bias, const float* __restrict__ mask,
                                                         float* __restrict__ out, int cq, int relu) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    f32x4 v[8];
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int z0 = 0; z0 < zsplit; z0 += 8) {  // eight partials per trip, loads first, added in order
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(part + (size_t)min(z0 + u, zsplit - 1) * stride + i * 4);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (z0 + u < zsplit) s += v[u];
    }
    if (bias) s += *reinterpret_cast<const f32x4*>(bias + (i % cq) * 4);
    if (relu) {
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] = fmaxf(s[e], 0.f);
    }
    if (mask) {
      const f32x4 mk = *reinterpret_cast<const f32x4*>(mask + i * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] = mk[e] > 0.f ? s[e] : 0.f;
    }
    *reinterpret_cast<f32x4*>(out + i * 4) = s;
  }
}

// finishes the tail tiles of a launch (WinoArgs::full / tsplit): one workgroup per tile, one thread per (2x2 tile, channel quad) as in
// wino_kernel's epilogue; the parts are added in order, then bias, ReLU / mask, stores
template <int BN>
__global__ __launch_bounds__(512) void wino_tail_fixup_kernel(const WinoArgs a) {
  constexpr int CQ = BN / 4, ITEMS = WT * CQ;
  const int tid = threadIdx.x;
  if (tid >= ITEMS) return;
  const int tile = a.full + (int)blockIdx.x;
  const int tb = tile % a.tblocks, cb = tile / a.tblocks;
  const int et = tid / CQ, cq = tid % CQ;
  const int t = tb * WT + et;
  const bool tvalid = t < a.T;
  if (!tvalid && !a.stats) return;
  const float* src = a.tpart + (size_t)blockIdx.x * a.tsplit * (WT * 4 * BN) + (size_t)(et * 4) * BN + 4 * cq;
  f32x4 y[4];
  f32x4 v[4][4];
  for (int z0 = 0; z0 < a.tsplit; z0 += 4) {  // four parts per trip: loads first, added in order
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int p = 0; p < 4; ++p)
        v[u][p] = *reinterpret_cast<const f32x4*>(src + (size_t)min(z0 + u, a.tsplit - 1) * (WT * 4 * BN) + p * BN);
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (z0 + u < a.tsplit) {
#pragma unroll
        for (int p = 0; p < 4; ++p) y[p] = (z0 + u == 0) ? v[u][p] : y[p] + v[u][p];
      }
  }
  const int tw = t % a.TW, r = t / a.TW, th = r % a.TH, n = r / a.TH;
  const int co = cb * BN + 4 * cq;
  const size_t p00 = (((size_t)n * a.H + 2 * th) * a.W + 2 * tw) * a.Cout + co;
  const size_t offs[4] = {p00, p00 + a.Cout, p00 + (size_t)a.W * a.Cout, p00 + (size_t)a.W * a.Cout + a.Cout};
  f32x4 mk[4];
  if (a.mask && tvalid) {
#pragma unroll
    for (int p = 0; p < 4; ++p) mk[p] = *reinterpret_cast<const f32x4*>(a.mask + offs[p]);
  }
  if (a.bias) {
    const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + co);
#pragma unroll
    for (int p = 0; p < 4; ++p) y[p] += bv;
  }
  if (a.stats) {  // the tile block's BatchNorm partials, as wino_kernel's epilogue forms them
    __shared__ f32x4 red[(ITEMS / 64) * CQ * 2];
    const int lane = tid & 63, wave = tid >> 6;
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    if (tvalid) {
#pragma unroll
      for (int p = 0; p < 4; ++p) { s1 += y[p]; s2 += y[p] * y[p]; }
    }
#pragma unroll
    for (int o = CQ; o < 64; o <<= 1)
#pragma unroll
      for (int e = 0; e < 4; ++e) { s1[e] += __shfl_xor(s1[e], o, 64); s2[e] += __shfl_xor(s2[e], o, 64); }
    constexpr int NW = ITEMS / 64;
    if (lane < CQ) { red[(wave * CQ + lane) * 2 + 0] = s1; red[(wave * CQ + lane) * 2 + 1] = s2; }
    __syncthreads();
    if (tid < CQ) {
      f32x4 t1 = red[tid * 2], t2 = red[tid * 2 + 1];
#pragma unroll
      for (int wv = 1; wv < NW; ++wv) { t1 += red[(wv * CQ + tid) * 2]; t2 += red[(wv * CQ + tid) * 2 + 1]; }
      float* o = a.stats + ((size_t)tb * a.Cout + cb * BN + 4 * tid) * 2;
#pragma unroll
      for (int e = 0; e < 4; ++e) { o[2 * e] = t1[e]; o[2 * e + 1] = t2[e]; }
    }
    if (!tvalid) return;
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float w = y[p][e];
      if (a.relu) w = fmaxf(w, 0.f);
      if (a.mask) w = mk[p][e] > 0.f ? w : 0.f;
      y[p][e] = w;
    }
    *reinterpret_cast<f32x4*>(a.out + offs[p]) = y[p];
  }
}

// U = G g G^T of every (output channel, input channel) pair, laid out as wino_kernel's waves load it.
// transpose = 0: the layer itself (g = w[co][ci]); 1: its data gradient (output channels = the layer's inputs, g =
// w[ci][co] with both taps flipped).  One thread per (row channel, contraction channel) pair.
__global__ void wino_pack_kernel(const float* __restrict__ w, float* __restrict__ upk, int Cout, int Cin, int transpose) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)Cout * Cin) return;
  srx_wino_pack_one(w, upk, Cout, Cin, transpose, idx);
}

struct WinoPlan { int bn, zsplit; float cost; int tsplit; };  // tsplit > 1: the tiles of the last round are cut that many ways (zsplit = 1)

// rounds of the chip x (chunks per workgroup x time per chunk + fixed cost), plus the fix-up pass of a split
WinoPlan wino_plan(int T, int Cin, int Cout, bool no_split = false, bool no_tail = false) {
  const int P = srx_plan_cus();
  const int nch = Cin / WKC, tblocks = (int)srx_cdiv(T, WT);
  WinoPlan best{64, 1, 1e30f};
  for (int bn = 64; bn >= 32; bn -= 32) {
    if (Cout % bn) continue;
    const int ncb = Cout / bn;
    for (int zs = 1; zs <= (no_split ? 1 : nch) && zs <= 16; ++zs) {
      const int64_t wgs = (int64_t)tblocks * ncb * zs;
      const float rounds = (float)srx_cdiv(wgs, P);
      const float chunks = (float)srx_cdiv(nch, zs);
      const float t_chunk = bn == 64 ? 4.4f : 2.5f;  // us: 128 / 64 fp32 MFMAs per SIMD pair and chunk at ~2 GHz, 32-column tiles a little over half
      const float fixed = bn == 64 ? 4.5f : 3.5f;    // prologue (first patch + U round trip) and the epilogue through LDS
      float cost = rounds * (chunks * t_chunk + fixed);
      if (zs > 1) cost += 3.0f + (float)(zs + 1) * T * 4.0f * Cout * 4.0f / 4.0e6f;  // fix-up: (zs + 1) passes over the output at ~4 TB/s
      if (cost < best.cost) best = WinoPlan{bn, zs, cost, 1};
      // whole tiles for the full rounds, the tiles of the last round cut along the input channels so that it is a short one
      const int tail = (int)(wgs % P);
      if (zs == 1 && !no_tail && !srx_dev().wino_no_tail && wgs > P && tail > 0) {
        for (int ts = 2; ts <= nch && ts <= 8; ++ts) {
          if ((int64_t)tail * ts > P) break;
          const float c = (float)(wgs / P) * (nch * t_chunk + fixed) + (float)srx_cdiv(nch, ts) * t_chunk + fixed + 4.0f;  // (+ the fix-up launch)
          if (c < best.cost) best = WinoPlan{bn, 1, c, ts};
        }
      }
    }
  }
  if (const int v = srx_dev().wino_zsplit; v > 0 && v <= nch && !no_split) { best.zsplit = v; best.tsplit = 1; }
  (void)no_tail;
  if (const int v = srx_dev().wino_bn; (v == 32 || v == 64) && Cout % v == 0) best.bn = v;
  return best;
}

// shuffle_ok: also layers with nn.PixelShuffle(2) in their store (the forward of the sub-pixel convs, inference)
bool wino_shape_ok(const srx_conv2d_t* d, bool shuffle_ok = false) {
  const bool sh = d->shuffle == 2 && shuffle_ok && d->Cout % 128 == 0 && d->Cout_s * 4 == d->Cout;
  return d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && (!d->shuffle || sh) && d->up != 2 && d->precision == 0 &&
         d->Cin % WKC == 0 && d->Cout % 32 == 0 && d->Cin_s == d->Cin && (sh || d->Cout_s == d->Cout) && d->H % 2 == 0 && d->W % 2 == 0 &&
         // 32-bit byte offsets into the INPUT (the output is addressed with 64 bits); without the shuffle the layer may also run
         // as its own data gradient, whose input has Cout channels
         (int64_t)d->N * d->H * d->W * (sh ? d->Cin : std::max(d->Cin, d->Cout)) < (1LL << 30) && (int64_t)d->N * d->H * d->W < (1LL << 31);
}

}  // namespace

extern "C" int srx_wino_applicable(const srx_conv2d_t* d) {
  // (the small 64 -> 64 layers of the residual tower keep their row-tile kernel: 72 tile blocks would fill a quarter of the chip)
  if (srx_dev().no_wino || !d || !wino_shape_ok(d) || srx_rt36_applicable(d)) return 0;
  // ... and so does any layer too small to fill the chip with 32-tile blocks even without a channel split (the form the
  // BatchNorm-statistics epilogue needs): the direct kernel's K-split plans serve those better
  const double direct_us = 2.0 * d->N * d->H * d->W * (double)d->Cout * 9.0 * d->Cin / 110.0e6 + 5.0;
  return wino_plan(d->N * (d->H / 2) * (d->W / 2), d->Cin, d->Cout, true).cost < direct_us ? 1 : 0;
}

extern "C" size_t srx_wino_packed_floats(const srx_conv2d_t* d) {
  return (d && wino_shape_ok(d, true)) ? (size_t)16 * d->Cout * d->Cin : 0;
}

extern "C" int srx_wino_pack(const srx_conv2d_t* d, const float* w, float* upk, int transpose, void* stream) {
  SRX_REQUIRE(d && w && upk, "wino_pack: null pointer");
  if (!wino_shape_ok(d, !transpose)) SRX_FAIL(SRX_E_UNSUPPORTED, "wino_pack: 3x3 / stride 1 / pad 1 fp32 layers with Cin, Cout multiples of 32 and even H, W only");
  const int64_t n = (int64_t)d->Cout * d->Cin;
  hipLaunchKernelGGL(wino_pack_kernel, dim3((unsigned)srx_cdiv(n, 256)), dim3(256), 0, srx_stream(stream), w, upk, d->Cout, d->Cin,
                     transpose ? 1 : (d->shuffle ? 2 : 0));  // (a PixelShuffle layer's forward: rows in (sub-pixel, channel) order)
  SRX_CHECK_LAUNCH("wino_pack_kernel");
  return SRX_OK;
}

// workspace (floats) of a forward (which = 0) or data-gradient (which = 1) call: the partial outputs of a channel split
extern "C" size_t srx_wino_ws_floats(const srx_conv2d_t* d, int which) {
  if (!d || !wino_shape_ok(d, which != 1)) return 0;
  const int cin = which == 1 ? d->Cout : d->Cin, cout = which == 1 ? d->Cin : d->Cout;
  const int T = d->N * (d->H / 2) * (d->W / 2);
  // (which = 2: the forward with BatchNorm statistics -- whole outputs per tile, so no split of every tile; the last round's may be cut)
  const WinoPlan p = wino_plan(T, cin, cout, d->shuffle != 0 || which == 2, d->shuffle != 0);
  if (p.tsplit > 1) {
    const int64_t wgs = srx_cdiv(T, WT) * (cout / p.bn);
    return (size_t)(wgs % srx_plan_cus()) * p.tsplit * WT * 4 * p.bn;
  }
  return p.zsplit > 1 ? (size_t)p.zsplit * d->N * d->H * d->W * cout : 0;
}

// out[6] = {BN, channel splits, workgroups, tile blocks, chunks per workgroup (max), 0}
extern "C" int srx_wino_plan(const srx_conv2d_t* d, int which, int* out) {
  SRX_REQUIRE(d && out && wino_shape_ok(d), "wino_plan: not a Winograd layer");
  const int cin = which ? d->Cout : d->Cin, cout = which ? d->Cin : d->Cout;
  const int T = d->N * (d->H / 2) * (d->W / 2);
  const WinoPlan p = wino_plan(T, cin, cout);
  out[0] = p.bn; out[1] = p.zsplit; out[3] = (int)srx_cdiv(T, WT); out[2] = out[3] * (cout / p.bn) * p.zsplit;
  out[4] = (int)srx_cdiv(cin / WKC, p.zsplit); out[5] = p.tsplit;
  if (p.tsplit > 1) { const int tail = out[2] % srx_plan_cus(); out[2] += tail * (p.tsplit - 1); }
  return SRX_OK;
}

static int wino_run(const srx_conv2d_t* d, int which, const float* x, const float* upk, const float* bias, const float* mask,
                    int relu, float* y, float* ws, size_t ws_floats, void* stream, float* stats = nullptr, const float* add = nullptr,
                    int lrelu = 0, float slope = 0.f) {
  SRX_REQUIRE(d && x && upk && y, "wino: null pointer");
  if (!wino_shape_ok(d, which == 0)) SRX_FAIL(SRX_E_UNSUPPORTED, "wino: 3x3 / stride 1 / pad 1 fp32 layers with Cin, Cout multiples of 32 and even H, W only");
  SRX_REQUIRE(x != y && (!mask || mask != y), "wino: in place is not possible (neighbouring tiles read their halo)");
  SRX_REQUIRE(!d->shuffle || (!stats && !mask), "wino: PixelShuffle layers: plain forward only");
  WinoArgs a{};
  a.in = x; a.upk = upk; a.bias = bias; a.mask = mask; a.out = y;
  a.N = d->N; a.H = d->H; a.W = d->W;
  a.Cin = which ? d->Cout : d->Cin;
  a.Cout = which ? d->Cin : d->Cout;
  a.TH = d->H / 2; a.TW = d->W / 2; a.T = d->N * a.TH * a.TW;
  a.tblocks = (int)srx_cdiv(a.T, WT);
  a.nch = a.Cin / WKC;
  a.relu = relu;
  a.in_bytes = (unsigned)((size_t)d->N * d->H * d->W * a.Cin * sizeof(float));
  a.upk_bytes = (unsigned)((size_t)16 * a.Cin * a.Cout * sizeof(float));
  a.out_elems = (size_t)d->N * d->H * d->W * a.Cout;
  SRX_REQUIRE(!add || (add != y && add != x), "wino: the addend must be a tensor of its own");
  a.add = add; a.lrelu = lrelu; a.slope = slope;
  // (statistics come from whole outputs, and the fix-up pass knows neither an addend nor a LeakyReLU: no channel split then)
  a.shuffle = d->shuffle ? d->Cout / 4 : 0;
  const bool plain = add == nullptr && lrelu == 0 && d->shuffle == 0;  // (the fix-up passes know bias, ReLU, mask and statistics)
  const WinoPlan p = wino_plan(a.T, a.Cin, a.Cout, stats != nullptr || !plain, !plain);
  a.zsplit = p.zsplit;
  a.ncb = a.Cout / p.bn;
  a.stats = stats;
  if (p.zsplit > 1) {
    if (!ws || ws_floats < (size_t)p.zsplit * a.out_elems) SRX_FAIL(SRX_E_WORKSPACE, "wino: workspace %zu < %zu floats", ws_floats, (size_t)p.zsplit * a.out_elems);
    a.part = ws;
  }
  hipStream_t st = srx_stream(stream);
  int64_t items = (int64_t)a.tblocks * a.ncb * a.zsplit;
  a.full = (int)items; a.tsplit = 1;
  int tail = 0;
  if (p.tsplit > 1) tail = (int)(items % srx_plan_cus());
  if (tail > 0) {  // (0: a developer override of BN left no partly filled round)
    const size_t need = (size_t)tail * p.tsplit * WT * 4 * p.bn;
    if (!ws || ws_floats < need) SRX_FAIL(SRX_E_WORKSPACE, "wino: workspace %zu < %zu floats", ws_floats, need);
    a.full = (int)items - tail; a.tsplit = p.tsplit; a.tpart = ws;
    items = a.full + (int64_t)tail * p.tsplit;
  }
  const dim3 grid((unsigned)items);
  static std::once_flag once;
  std::call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  char nm[112];
  if (srx_prof_on()) snprintf(nm, sizeof(nm), "wino_kernel<%d> MxNxK=%dx%dx%d", p.bn, a.N * a.H * a.W, a.Cout, 9 * a.Cin);
  const double fl = 2.0 * a.N * a.H * a.W * (double)a.Cout * 9.0 * a.Cin;  // algorithmic FLOPs of the convolution (the direct form's)
  const size_t lds = WINO_LDS + (stats ? 4096 : 0);
  if (p.bn == 64) SRX_LAUNCH_PROF(nm, fl, wino_kernel<64>, grid, dim3(512), lds, st, a);
  else SRX_LAUNCH_PROF(nm, fl, wino_kernel<32>, grid, dim3(512), lds, st, a);
  SRX_CHECK_LAUNCH("wino_kernel");
  if (p.zsplit > 1) {
    const size_t n4 = a.out_elems / 4;
    const unsigned blocks = (unsigned)std::min<size_t>(srx_cdiv((int64_t)n4, 256), 4096);
    hipLaunchKernelGGL(wino_fixup_kernel, dim3(blocks), dim3(256), 0, st, ws, p.zsplit, n4, a.out_elems, bias, mask, y, a.Cout / 4, relu);
    SRX_CHECK_LAUNCH("wino_fixup_kernel");
  }
  if (tail > 0) {
    if (p.bn == 64) hipLaunchKernelGGL(wino_tail_fixup_kernel<64>, dim3((unsigned)tail), dim3(512), 0, st, a);
    else hipLaunchKernelGGL(wino_tail_fixup_kernel<32>, dim3((unsigned)tail), dim3(256), 0, st, a);
    SRX_CHECK_LAUNCH("wino_tail_fixup_kernel");
  }
  return SRX_OK;
}

extern "C" int srx_wino_fwd(const srx_conv2d_t* d, const float* x, const float* upk, const float* bias, float* y, float* ws,
                            size_t ws_floats, void* stream) {
  SRX_REQUIRE(d && (d->act == SRX_ACT_NONE || d->act == SRX_ACT_RELU), "wino_fwd: no activation or ReLU");
  return wino_run(d, 0, x, upk, bias, nullptr, d->act == SRX_ACT_RELU, y, ws, ws_floats, stream);
}

// inference forms (functional.FoldedConv: conv with the eval-mode BatchNorm folded in, srgan/residual.py:86-91): any of no
// activation / ReLU / LeakyReLU(d->slope) (a single-parameter PReLU is that), then `+ residual` (may be NULL)
extern "C" int srx_wino_fwd_act(const srx_conv2d_t* d, const float* x, const float* upk, const float* bias, const float* residual,
                                float* y, float* ws, size_t ws_floats, void* stream) {
  SRX_REQUIRE(d && (d->act == SRX_ACT_NONE || d->act == SRX_ACT_RELU || d->act == SRX_ACT_LRELU), "wino_fwd_act: no activation, ReLU or LeakyReLU");
  return wino_run(d, 0, x, upk, bias, nullptr, d->act == SRX_ACT_RELU, y, ws, ws_floats, stream, nullptr, residual,
                  d->act == SRX_ACT_LRELU, d->slope);
}
// Is the Winograd form of this layer's forward worth it at inference?  As srx_wino_applicable, for layers the training path
// leaves to the 36-pixel row tile as well (3x3 64 -> 64 on a frame: thousands of tile blocks)
extern "C" int srx_wino_infer_applicable(const srx_conv2d_t* d) {
  if (srx_dev().no_wino || !d || !wino_shape_ok(d, true)) return 0;
  const double direct_us = 2.0 * d->N * d->H * d->W * (double)d->Cout * 9.0 * d->Cin / 110.0e6 + 5.0;
  return wino_plan(d->N * (d->H / 2) * (d->W / 2), d->Cin, d->Cout, true).cost < direct_us ? 1 : 0;
}

// forward of a layer followed by a training-mode BatchNorm2d (srgan/discriminator.py:35-61: conv without bias, BatchNorm,
// LeakyReLU): stats [srx_wino_stat_rows(d)][Cout][2] receives per tile block (128 output pixels: 32 tiles of 2x2) the
// per-channel sum and sum of squares of the output, as srx_conv2d_fwd's bn_partials does per row tile
extern "C" int srx_wino_stat_rows(const srx_conv2d_t* d) {
  return (d && wino_shape_ok(d)) ? (int)srx_cdiv((int64_t)d->N * (d->H / 2) * (d->W / 2), WT) : 0;
}
extern "C" int srx_wino_fwd_stats(const srx_conv2d_t* d, const float* x, const float* upk, const float* bias, float* y, float* stats,
                                  float* ws, size_t ws_floats, void* stream) {
  SRX_REQUIRE(d && d->act == SRX_ACT_NONE && stats, "wino_fwd_stats: a linear layer and a statistics table");
  return wino_run(d, 0, x, upk, bias, nullptr, 0, y, ws, ws_floats, stream, stats);
}

extern "C" int srx_wino_bwd_data(const srx_conv2d_t* d, const float* dy, const float* upk_t, const float* relu_out, float* dx,
                                 float* ws, size_t ws_floats, void* stream) {
  return wino_run(d, 1, dy, upk_t, nullptr, relu_out, 0, dx, ws, ws_floats, stream);
}
