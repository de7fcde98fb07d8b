// Convolutions with a 3-channel side (image in / image out): generator conv3 64->3 9x9
// (srgan/generator.py:58), generator conv1 3->64 9x9 (:38), discriminator / VGG19 first layers 3->64
// 3x3 (srgan/discriminator.py:32, VGG cfg 'E'), ESRGAN conv1 / conv4 (esrgan/generator.py:36,52).
//
// On the generic 32-wide MFMA tile these waste 8-10x of the matrix pipe (N = 3 padded to 32).  Here
// they run on v_mfma_f32_4x4x1_16b_f32: 16 independent 4x4 outer products per instruction,
//     D[reg i] on lane l  +=  A(lane 4*(l/4)+i) * B(lane l)          (probed on gfx950)
// so the 4-wide operand A holds the thin side (3 channels + 1 zero, identical in all 16 blocks) and
// the 64 lanes of B hold the wide side: 75 % of the pipe does useful work and the thin operand is a
// 4-address broadcast.  Same 64 FLOP/clk/SIMD rate as the 32x32 MFMA, exact fp32.
#include "srx_common.h"
#include <algorithm>
#include <cstdio>
#include <mutex>
#include <cstdlib>

namespace {

struct ThinW {
  const float* wide;   // [N][H][W][64]
  const float* thin;   // [N][H][W][4]
  float* slab;         // [ranges][4][taps][64]
  int N, H, W;
  int nseg, segs_per_range, wsegs, nranges;  // segments = (n, row, 16-column block)
};

// Weight gradient.  acc[c][tap][ch] = sum_q wide[q][ch] * thin[q + SGN*(tap - pad)][c].
//   SGN = -1: thin side is the conv OUTPUT gradient (conv3):     dW[co=c][ci=ch][tap]
//   SGN = +1: thin side is the conv INPUT image (first layers):  dW[co=ch][ci=c][tap]
// One wave = one (segment range, group of kernel rows).  Per segment (16 pixels of one image row) the wave
// holds the 16 wide rows in registers (lane = channel), and per kernel row the 16+KW-1 thin values it
// slides over, so all 16*KW MFMAs of that row run on registers with static indices: no LDS at all.
template <int KH, int KW, int GROUPS, int SGN>
__global__ __launch_bounds__(256) void thin_wgrad_kernel(const ThinW a) {
  constexpr int RPG = KH / GROUPS;  // kernel rows per wave
  constexpr int PAD = (KH - 1) / 2;
  constexpr int NT = 16 + KW - 1;
  // the four waves of a workgroup share the group of kernel rows and take four consecutive segment
  // ranges; their accumulators are folded through LDS before the slab is written (the slabs are
  // 4 x taps x 64 floats each -- 83 KB for 9x9 -- and used to cost more traffic than the operands)
  __shared__ float fold[2][RPG * KW * 4 * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int group = blockIdx.x % GROUPS, rb = blockIdx.x / GROUPS;
  const int range = rb * 4 + wave;
  const int sbeg = min(a.nseg, range * a.segs_per_range);  // ranges past the end are empty
  const int send = min(a.nseg, sbeg + a.segs_per_range);
  const int tc = lane & 3;

  f32x4 acc[RPG * KW];
#pragma unroll
  for (int t = 0; t < RPG * KW; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // Round 6: every address of this loop is WAVE-UNIFORM up to the lane's channel (wide: lane, thin: lane & 3) -- the segment, the
  // kernel row and the column are the wave's, not the lane's -- so both operands come through raw buffer descriptors with the pixel
  // offset (or, for a padding pixel / a column past the row, an offset past the tensor: reads 0; the range check includes the scalar
  // offset, tools/probe/buf_range.hip) in the instruction's SCALAR offset: no per-lane compare, select or multiply is left.  The
  // f32 MFMAs run on the vector ALUs (tools/probe/mfma_valu.hip): the ~340 VALU instructions per segment of the select form cost
  // 0.7 of its 432 MFMAs' time (PMC, round 5: 35 % MFMA-busy at 24 M VALU instructions per launch).
  const unsigned wide_bytes = (unsigned)((size_t)a.N * a.H * a.W * 256), thin_bytes = (unsigned)((size_t)a.N * a.H * a.W * 16);
  const __amdgpu_buffer_rsrc_t rwd = srx_rsrc(a.wide, wide_bytes), rth = srx_rsrc(a.thin, thin_bytes);
  const unsigned wlane = 4u * (unsigned)lane, tlane = 4u * (unsigned)tc;
  const int s0 = srx_uniform(sbeg), s1 = srx_uniform(send);
  for (int s = s0; s < s1; ++s) {
    const int cb = s % a.wsegs;
    const int row = (s / a.wsegs) % a.H;
    const int n = s / (a.wsegs * a.H);
    const int c0 = cb * 16;
    float wv[16];
    const unsigned wrow = (unsigned)(((n * a.H + row) * a.W + c0) * 256);
#pragma unroll
    for (int j = 0; j < 16; ++j)
      wv[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rwd, (int)wlane, (int)((c0 + j < a.W) ? wrow + 256u * j : wide_bytes), 0));
#pragma unroll
    for (int dr = 0; dr < RPG; ++dr) {
      const int kh = group * RPG + dr;
      const int tr = row + SGN * (kh - PAD);
      const bool rok = (unsigned)tr < (unsigned)a.H;
      float tv[NT];
      const unsigned trow = (unsigned)(((n * a.H + tr) * a.W) * 16);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int col = c0 - PAD + t;
        const bool ok = rok && (unsigned)col < (unsigned)a.W;
        tv[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rth, (int)tlane, (int)(ok ? trow + 16u * (unsigned)col : thin_bytes), 0));
      }
#pragma unroll
      for (int j = 0; j < 16; ++j)
#pragma unroll
        for (int kw = 0; kw < KW; ++kw) {
          // thin column = (c0+j) + SGN*(kw-PAD)  ->  index into tv (which starts at c0-PAD)
          const int ti = SGN > 0 ? j + kw : j + (KW - 1 - kw);
          acc[dr * KW + kw] = __builtin_amdgcn_mfma_f32_4x4x1f32(tv[ti], wv[j], acc[dr * KW + kw], 0, 0, 0);
        }
    }
  }
  // fold: waves 2,3 -> LDS, waves 0,1 add; wave 1 -> LDS, wave 0 adds and writes the slab
  auto put = [&](int slot) {
#pragma unroll
    for (int t = 0; t < RPG * KW; ++t)
#pragma unroll
      for (int c = 0; c < 4; ++c) fold[slot][(t * 4 + c) * 64 + lane] = acc[t][c];
  };
  auto add = [&](int slot) {
#pragma unroll
    for (int t = 0; t < RPG * KW; ++t)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[t][c] += fold[slot][(t * 4 + c) * 64 + lane];
  };
  if (wave >= 2) put(wave - 2);
  __syncthreads();
  if (wave < 2) add(wave);
  __syncthreads();
  if (wave == 1) put(0);
  __syncthreads();
  if (wave != 0) return;
  add(0);
  float* o = a.slab + (size_t)rb * (4 * KH * KW * 64);
#pragma unroll
  for (int dr = 0; dr < RPG; ++dr)
#pragma unroll
    for (int kw = 0; kw < KW; ++kw) {
      const int tap = (group * RPG + dr) * KW + kw;
#pragma unroll
      for (int c = 0; c < 4; ++c) o[((size_t)c * (KH * KW) + tap) * 64 + lane] = acc[dr * KW + kw][c];
    }
}

// slab sums -> OIHW.  thin_is_out: dw[c][ch][tap] else dw[ch][c][tap];  Cthin real thin channels (3).
// One workgroup per (c, tap): 64 channels x 4 range lanes, independent loads, LDS fold.
__global__ __launch_bounds__(256) void thin_wgrad_reduce_kernel(const float* __restrict__ slab, int ranges, int taps,
                                                                int Cthin, int Cwide, int thin_is_out,
                                                                float* __restrict__ dw, int accumulate) {
  __shared__ float red[4][64];
  const int ch = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int tap = blockIdx.x % taps, c = blockIdx.x / taps;
  const float* p = slab + ((size_t)c * taps + tap) * 64 + ch;
  const size_t stride = (size_t)4 * taps * 64;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int r = part;
  for (; r + 12 < ranges; r += 16) {
    s0 += p[(size_t)r * stride];
    s1 += p[(size_t)(r + 4) * stride];
    s2 += p[(size_t)(r + 8) * stride];
    s3 += p[(size_t)(r + 12) * stride];
  }
  for (; r < ranges; r += 4) s0 += p[(size_t)r * stride];
  red[part][ch] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (part == 0 && ch < Cwide) {
    const float s = (red[0][ch] + red[1][ch]) + (red[2][ch] + red[3][ch]);
    float* o = thin_is_out ? dw + ((size_t)c * Cwide + ch) * taps + tap : dw + ((size_t)ch * Cthin + c) * taps + tap;
    *o = accumulate ? *o + s : s;
  }
}

// ---------------------------------------------------------------------------------------------
// Thin OUTPUT convolution: out[p][c<4] = sum_{tap,ch} in[p+tap-pad][ch] * w[c][tap][ch] (+bias).
// Used for the forward of conv3 / conv4 (in = activations) and for the data gradient of the
// 3-channel first layers (in = dy, w = flipped / transposed taps).  lane = output pixel; a workgroup
// owns a TH x 32 block of output pixels, stages the (TH+KH-1) x (32+KW-1) input patch 16 channels at a
// time in LDS (row stride 20 floats: conflict-free ds_read_b128) together with that channel slice of
// the weights, and every wave sweeps 2 x 64 pixels so one weight read feeds two MFMA groups.
// ---------------------------------------------------------------------------------------------
struct ThinF {
  const float* in;    // [N][H][W][64]
  const float* w;     // packed [4][taps][64] (thin channel, tap, wide channel), zero padded
  const float* bias;  // [Cout] or null
  float* out;         // [N][H][W][4]
  int N, H, W, Cout;
  int tiles_w, tiles_h;
  unsigned in_bytes;
  int in_bf16;  // thin_fwd2_bf16_kernel: `in` holds bf16 (128 bytes per pixel), the bf16-native inference chain (c64.hip)
};

// Forward kernel (also the data gradient of the 3-channel-input layers, with flipped taps).  A first
// version read three LDS b128 operands for every eight MFMAs and was LDS-bandwidth bound (136 B/clk per CU
// asked, 128 available; 168 us for conv3 at batch 16).  Here a lane owns R output rows of one column
// and one half (4) of the 8 channels staged per round:
//   * for a fixed kernel column kw the nine (KH) weight fragments live in registers,
//   * every input pixel fragment read from LDS (rows r .. r+R+KH-2 of column c+kw) feeds all the
//     output rows it touches (kh = input row - output row),
// i.e. (K + R + K - 1) b128 reads per 4 R K MFMAs (21 per 144 at R = 4, where the first version needed 54).  Lanes 0-31 / 32-63 hold the two channel halves of the
// same 32 pixels and are summed at the end; pixels are stored 32 B apart so a wave's read is one
// contiguous 2 KB block.  Output tile 4R x 32 pixels per workgroup (4 waves x R rows); 84 us for conv3.
template <int K, int R>
__global__ __launch_bounds__(256) void thin_fwd2_kernel(const ThinF a) {
  constexpr int TH = 4 * R, TW = 32, CS = 8;  // CS: channels staged per round
  constexpr int PH = TH + K - 1, PW = TW + K - 1, PAD = (K - 1) / 2, TAPS = K * K;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sx = reinterpret_cast<float*>(smem);  // [PH][PW][CS]
  float* sw = sx + PH * PW * CS;               // [4][TAPS][CS]
  const int tid = threadIdx.x, lane = tid & 63, wave = srx_uniform(tid >> 6);
  int b = blockIdx.x;
  const int tw_i = b % a.tiles_w; b /= a.tiles_w;
  const int th_i = b % a.tiles_h;
  const int n = b / a.tiles_h;
  const int h0 = th_i * TH, w0 = tw_i * TW;
  const int pc = lane & 31, hh = lane >> 5, r0 = wave * R, wc = lane & 3;
  // the descriptor covers the image rows this tile's patch can touch, based at the first of them (a 64-bit base; offsets
  // inside it are small): tensors above 4 GiB / 2^24 pixels -- the 8K frame -- need no tiling (round 4)
  const int hb = max(h0 - PAD, 0), he = min(h0 - PAD + PH, a.H);
  const __amdgpu_buffer_rsrc_t rin = srx_rsrc(a.in + ((size_t)n * a.H + hb) * (size_t)a.W * 64, (unsigned)((size_t)(he - hb) * a.W * 256));
  const __amdgpu_buffer_rsrc_t rw = srx_rsrc(a.w, 4 * TAPS * 64 * 4);

  f32x4 acc[R];
#pragma unroll
  for (int j = 0; j < R; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  constexpr int NPX = PH * PW * 2;                      // b128 units of the patch (pixel, channel quad)
  constexpr int LP = (NPX + 255) / 256;                 // loads per thread (all issued before any is written)
  constexpr int NWQ = 4 * TAPS * 2, LW = (NWQ + 255) / 256;
  // the patch / weight offsets of this thread do not depend on the channel round: computed ONCE (round 6: ~20 VALU instructions per
  // load and round were matrix time lost -- the f32 MFMA shares the vector ALUs), the round's channel offset is the load's scalar
  // offset; a slot outside the image keeps an offset that stays out of range with any round's offset added (no 32-bit wrap)
  constexpr unsigned OOR = 0x7ffffff0u;
  unsigned poff[LP], woff[LW];
#pragma unroll
  for (int u = 0; u < LP; ++u) {
    const int i = u * 256 + tid;
    const int q = i & 1, pix = i >> 1;
    const int pw = pix % PW, ph = pix / PW;
    const int ih = h0 - PAD + ph, iw = w0 - PAD + pw;
    const bool ok = i < NPX && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
    poff[u] = ok ? 4u * (unsigned)(((ih - hb) * a.W + iw) * 64 + q * 4) : OOR;
  }
#pragma unroll
  for (int u = 0; u < LW; ++u) {
    const int i = u * 256 + tid;  // (c*TAPS + tap) * 2 + q
    woff[u] = i < NWQ ? 4u * (unsigned)((i >> 1) * 64 + (i & 1) * 4) : OOR;
  }
  for (int cc = 0; cc < 64; cc += CS) {
    f32x4 vp[LP], vw[LW];
#pragma unroll
    for (int u = 0; u < LP; ++u) vp[u] = srx_bload(rin, poff[u], (unsigned)(cc * 4));
#pragma unroll
    for (int u = 0; u < LW; ++u) vw[u] = srx_bload(rw, woff[u], (unsigned)(cc * 4));
    __syncthreads();  // the previous round's reads are done
#pragma unroll
    for (int u = 0; u < LP; ++u) {
      const int i = u * 256 + tid;
      if (i < NPX) *reinterpret_cast<f32x4*>(sx + i * 4) = vp[u];
    }
#pragma unroll
    for (int u = 0; u < LW; ++u) {
      const int i = u * 256 + tid;
      if (i < NWQ) *reinterpret_cast<f32x4*>(sw + i * 4) = vw[u];
    }
    __syncthreads();
#pragma unroll 1
    for (int kw = 0; kw < K; ++kw) {
      f32x4 wr[K];
#pragma unroll
      for (int kh = 0; kh < K; ++kh)
        wr[kh] = *reinterpret_cast<const f32x4*>(sw + ((wc * TAPS + kh * K + kw) * CS + 4 * hh));
      const float* xp = sx + ((r0 * PW) + pc + kw) * CS + 4 * hh;
#pragma unroll
      for (int ir = 0; ir < R + K - 1; ++ir) {  // input row r0 + ir of the patch
        const f32x4 xv = *reinterpret_cast<const f32x4*>(xp + ir * PW * CS);
#pragma unroll
        for (int j = 0; j < R; ++j) {
          const int kh = ir - j;
          if (kh >= 0 && kh < K) {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[kh][e], xv[e], acc[j], 0, 0, 0);
          }
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < R; ++j) {
    f32x4 v = acc[j];
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] += __shfl_xor(v[c], 32, 64);  // the other channel half
    const int oh = h0 + r0 + j, ow = w0 + pc;
    if (hh == 0 && oh < a.H && ow < a.W) {
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = c < a.Cout ? v[c] + (a.bias ? a.bias[c] : 0.f) : 0.f;
      *reinterpret_cast<f32x4*>(a.out + ((size_t)(n * a.H + oh) * a.W + ow) * 4) = v;
    }
  }
}

// The same forward kernel with bf16 products (srx_conv2d_t::precision = 2: inference with bf16 products in EVERY conv; the
// training paths keep the 3-channel layers exact, precision 0 / 1).  Same tiles, same LDS byte layout -- a pixel is 32 bytes,
// now 16 bf16 channels instead of 8 floats, so a round covers 16 channels (four rounds) and the two lane halves hold channels
// 0-7 / 8-15 -- and v_mfma_f32_4x4x4_16b_bf16 contracts four channels per instruction (probed on gfx950,
// tools/probe/mfma4x4_bf16.hip: D[reg i] on lane l += sum_k A(lane 4 (l / 4) + i)[k] B(lane l)[k]): a quarter of the MFMAs
// and half the LDS reads of the fp32 kernel, which the 9x9 64 -> 3 output conv at 8K resolution (33 M pixels) was bound by.
// Operands are rounded to bf16 (nearest even) as they are staged; accumulation is fp32.
template <int K, int R, bool IN16 = false>
__global__ __launch_bounds__(256) void thin_fwd2_bf16_kernel(const ThinF a) {
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  struct Frag { s16x4 lo, hi; };
  constexpr int TH = 4 * R, TW = 32, CS = 16;  // CS: channels staged per round
  constexpr int PH = TH + K - 1, PW = TW + K - 1, PAD = (K - 1) / 2, TAPS = K * K;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned char* sx = reinterpret_cast<unsigned char*>(smem);  // [PH][PW][CS] bf16
  unsigned char* sw = sx + PH * PW * CS * 2;                   // [4][TAPS][CS] bf16
  const int tid = threadIdx.x, lane = tid & 63, wave = srx_uniform(tid >> 6);
  int b = blockIdx.x;
  const int tw_i = b % a.tiles_w; b /= a.tiles_w;
  const int th_i = b % a.tiles_h;
  const int n = b / a.tiles_h;
  const int h0 = th_i * TH, w0 = tw_i * TW;
  const int pc = lane & 31, hh = lane >> 5, r0 = wave * R, wc = lane & 3;
  const int hb = max(h0 - PAD, 0), he = min(h0 - PAD + PH, a.H);  // (64-bit row base: see thin_fwd2_kernel)
  constexpr unsigned PXB = IN16 ? 128u : 256u;                    // bytes per input pixel
  const __amdgpu_buffer_rsrc_t rin = srx_rsrc(reinterpret_cast<const unsigned char*>(a.in) + ((size_t)n * a.H + hb) * (size_t)a.W * PXB,
                                              (unsigned)((size_t)(he - hb) * a.W * PXB));
  const __amdgpu_buffer_rsrc_t rw = srx_rsrc(a.w, 4 * TAPS * 64 * 4);

  f32x4 acc[R];
#pragma unroll
  for (int j = 0; j < R; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  constexpr int NPX = PH * PW * 2;       // 16-byte units of the patch (pixel, 8-channel half)
  constexpr int LP = (NPX + 255) / 256;  // units per thread (all loads issued before any is written)
  constexpr int NWQ = 4 * TAPS * 2, LW = (NWQ + 255) / 256;
  auto pack = [](const f32x4& u, const f32x4& v) {
    return bf16x8{(__bf16)u[0], (__bf16)u[1], (__bf16)u[2], (__bf16)u[3], (__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
  };
  for (int cc = 0; cc < 64; cc += CS) {
    f32x4 vp[LP][IN16 ? 1 : 2], vw[LW][2];
#pragma unroll
    for (int u = 0; u < LP; ++u) {
      const int i = u * 256 + tid;
      const int q = i & 1, pix = i >> 1;
      const int pw = pix % PW, ph = pix / PW;
      const int ih = h0 - PAD + ph, iw = w0 - PAD + pw;
      const bool ok = i < NPX && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
      if constexpr (IN16) {  // the unit (pixel, 8-channel half) is 16 bytes of the bf16 tensor as it stands
        vp[u][0] = srx_bload(rin, ok ? 2u * (unsigned)(((ih - hb) * a.W + iw) * 64 + cc + q * 8) : 0xffffffffu, 0);
      } else {
        const unsigned off = ok ? 4u * (unsigned)(((ih - hb) * a.W + iw) * 64 + cc + q * 8) : 0xffffffffu;
        vp[u][0] = srx_bload(rin, off, 0);
        vp[u][1] = srx_bload(rin, off, 16);
      }
    }
#pragma unroll
    for (int u = 0; u < LW; ++u) {
      const int i = u * 256 + tid;  // (c*TAPS + tap) * 2 + q
      const unsigned off = i < NWQ ? 4u * (unsigned)((i >> 1) * 64 + cc + (i & 1) * 8) : 0xffffffffu;
      vw[u][0] = srx_bload(rw, off, 0);
      vw[u][1] = srx_bload(rw, off, 16);
    }
    __syncthreads();  // the previous round's reads are done
#pragma unroll
    for (int u = 0; u < LP; ++u) {
      const int i = u * 256 + tid;
      if constexpr (IN16) { if (i < NPX) *reinterpret_cast<f32x4*>(sx + i * 16) = vp[u][0]; }
      else if (i < NPX) *reinterpret_cast<bf16x8*>(sx + i * 16) = pack(vp[u][0], vp[u][IN16 ? 0 : 1]);
    }
#pragma unroll
    for (int u = 0; u < LW; ++u) {
      const int i = u * 256 + tid;
      if (i < NWQ) *reinterpret_cast<bf16x8*>(sw + i * 16) = pack(vw[u][0], vw[u][1]);
    }
    __syncthreads();
#pragma unroll 1
    for (int kw = 0; kw < K; ++kw) {
      Frag wr[K];
#pragma unroll
      for (int kh = 0; kh < K; ++kh)
        wr[kh] = *reinterpret_cast<const Frag*>(sw + ((wc * TAPS + kh * K + kw) * CS + 8 * hh) * 2);
      const unsigned char* xp = sx + (((r0 * PW) + pc + kw) * CS + 8 * hh) * 2;
#pragma unroll
      for (int ir = 0; ir < R + K - 1; ++ir) {  // input row r0 + ir of the patch
        const Frag xv = *reinterpret_cast<const Frag*>(xp + ir * PW * CS * 2);
#pragma unroll
        for (int j = 0; j < R; ++j) {
          const int kh = ir - j;
          if (kh >= 0 && kh < K) {
            acc[j] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wr[kh].lo, xv.lo, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wr[kh].hi, xv.hi, acc[j], 0, 0, 0);
          }
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < R; ++j) {
    f32x4 v = acc[j];
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] += __shfl_xor(v[c], 32, 64);  // the other channel half
    const int oh = h0 + r0 + j, ow = w0 + pc;
    if (hh == 0 && oh < a.H && ow < a.W) {
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = c < a.Cout ? v[c] + (a.bias ? a.bias[c] : 0.f) : 0.f;
      *reinterpret_cast<f32x4*>(a.out + ((size_t)(n * a.H + oh) * a.W + ow) * 4) = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Thin INPUT convolution, forward: 3x3 / stride 1 / pad 1, image (<= 4 channels, stored as 4) -> 64 channels + bias +
// ReLU / LeakyReLU -- the first layers of the discriminators and of VGG19 (srgan/discriminator.py:32, VGG cfg 'E').
// K is only 36 (9 taps x 4 stored channels): on the generic tile that is two 32-wide k-chunks, one of them padding, and a
// pipeline that never fills (45 us for 295 k pixels, 30 TFLOP/s).  Here nothing is staged: a wave owns 32 pixels x 64
// columns (two 32x32x2 accumulators), keeps the WHOLE weight matrix in 36 registers per lane (lane = column, lane half =
// k parity, as the MFMA's B operand wants it; fetched through LDS) and reads each pixel's nine 16-byte neighbours straight into registers
// (lane = pixel; both lane halves load the same quad and pick their k parity's channel).  36 MFMAs per 32 pixels;
// the kernel is a write stream of 256 bytes per pixel.  The packed weights are the generic forward layout
// ([64][Kp], k = tap * 4 + channel), so nothing else changes.  PR: operands rounded to bf16 (products stay fp32).
// ---------------------------------------------------------------------------------------------
struct First3 {
  const float* in; const float* w; const float* bias; float* out;
  int H, W, HW, M, Kp, ntiles;
  float inv_HW, inv_W, slope;
  unsigned in_bytes, out_bytes;
};

template <int PR, bool OUT16 = false>  // OUT16: the output tensor is bf16 (bf16-storage conv stacks, gconv.hip, round 5)
__global__ __launch_bounds__(256) void first3x3_fwd_kernel(const First3 a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i31 = lane & 31, h2 = lane >> 5;
  const __amdgpu_buffer_rsrc_t rin = srx_rsrc(a.in, a.in_bytes);
  const __amdgpu_buffer_rsrc_t rout = srx_rsrc(a.out, a.out_bytes);
  auto rnd = [](float v) { return PR ? (float)(__bf16)v : v; };
  // the 64 x 36 weights go through LDS once per workgroup (rows of 144 contiguous bytes; a lane reading its own column's
  // row straight from memory touched 64 lines per load instruction: 41 us of L1 line requests per launch, measured 48.7 us)
  __shared__ float sw[64][37];
  for (int i = threadIdx.x; i < 64 * 36; i += 256) sw[i / 36][i % 36] = a.w[(size_t)(i / 36) * a.Kp + i % 36];
  __syncthreads();
  float b[18][2], bv[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    bv[j] = a.bias ? a.bias[32 * j + i31] : 0.f;
#pragma unroll
    for (int s = 0; s < 18; ++s) b[s][j] = rnd(sw[32 * j + i31][2 * s + h2]);
  }
  for (int tile = blockIdx.x * 4 + wave; tile < a.ntiles; tile += gridDim.x * 4) {
    const int p = tile * 32 + i31;
    int n, rem, ih, iw;
    srx_divmod(min(p, a.M - 1), a.HW, a.inv_HW, n, rem);
    srx_divmod(rem, a.W, a.inv_W, ih, iw);
    f32x4 v[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int y = ih + t / 3 - 1, x = iw + t % 3 - 1;
      const bool ok = p < a.M && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
      v[t] = srx_bload(rin, ok ? (unsigned)((n * a.H + y) * a.W + x) * 16u : 0xffffffffu, 0);
    }
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {  // k = 4 t + 2 jj + h2
        const float av = rnd(h2 ? v[t][2 * jj + 1] : v[t][2 * jj]);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[2 * t + jj][0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[2 * t + jj][1], acc1, 0, 0, 0);
      }
    // 32x32 accumulator: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5); rows past M fall outside the descriptor
    const unsigned obase = ((unsigned)tile * 32u * 64u + (unsigned)i31 + 256u * h2) * 4u;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float o0 = acc0[r] + bv[0], o1 = acc1[r] + bv[1];
      o0 = o0 > 0.f ? o0 : o0 * a.slope;
      o1 = o1 > 0.f ? o1 : o1 * a.slope;
      const int roff = ((r & 3) + 8 * (r >> 2)) * 256;
      if (OUT16) {
        __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (__bf16)o0), rout, (int)(obase >> 1), roff >> 1, 0);
        __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (__bf16)o1), rout, (int)(obase >> 1), (roff + 128) >> 1, 0);
      } else {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o0), rout, (int)obase, roff, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o1), rout, (int)obase, roff + 128, 0);
      }
    }
  }
}

// packs OIHW weights for thin_fwd_kernel: p[c][tap][ch].
//  mode 0 (forward, thin = Cout): p[c][kh*KW+kw][ch] = w[c][ch][kh][kw]
//  mode 1 (data gradient of a Cin<=4 layer, thin = Cin, wide = Cout, taps flipped):
//          p[c][kh*KW+kw][ch] = w[ch][c][KH-1-kh][KW-1-kw]
__global__ void thin_pack_kernel(const float* __restrict__ w, float* __restrict__ p, int Cthin, int Cwide, int KH,
                                 int KW, int mode) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int taps = KH * KW;
  if (idx >= 4 * taps * 64) return;
  const int ch = idx & 63, tap = (idx >> 6) % taps, c = idx / (64 * taps);
  float v = 0.f;
  if (c < Cthin && ch < Cwide) {
    const int kh = tap / KW, kw = tap - kh * KW;
    v = mode == 0 ? w[(((size_t)c * Cwide + ch) * KH + kh) * KW + kw]
                  : w[(((size_t)ch * Cthin + c) * KH + (KH - 1 - kh)) * KW + (KW - 1 - kw)];
  }
  p[idx] = v;
}

int thin_ranges(int nseg, int groups) {
  const int cus = srx_plan_cus();
  int ranges = (cus * 8) / groups;  // about two waves per SIMD: one's loads hide under the other's MFMAs
  if (ranges > nseg) ranges = nseg;
  if (ranges < 1) ranges = 1;
  return ranges;
}

}  // namespace

// ---- internal entry points used by gconv.hip's C ABI functions ------------------------------------
bool srx_thin_wgrad_applicable(const srx_conv2d_t* d) {  // (the 3-channel layers are fp32 at either precision)
  if (d->stride != 1 || d->shuffle || d->KH != d->KW || (d->KH != 3 && d->KH != 9) || d->pad != (d->KH - 1) / 2)
    return false;
  const bool thin_out = d->Cout <= 4 && d->Cout_s == 4 && d->Cin == 64 && d->Cin_s == 64;
  const bool thin_in = d->Cin <= 4 && d->Cin_s == 4 && d->Cout == 64 && d->Cout_s == 64;
  return thin_out || thin_in;
}

size_t srx_thin_wgrad_ws_floats(const srx_conv2d_t* d) {
  const int nseg = d->N * d->H * (int)srx_cdiv(d->W, 16);
  const int groups = d->KH == 9 ? 3 : 1;
  return (size_t)srx_cdiv(thin_ranges(nseg, groups), 4) * 4 * d->KH * d->KW * 64;  // one slab per workgroup
}

int srx_thin_wgrad(const srx_conv2d_t* d, const float* x, const float* dy, float* dw, int accumulate, float* ws,
                   size_t ws_floats, hipStream_t st) {
  const bool thin_out = d->Cout <= 4 && d->Cout_s == 4;
  ThinW a;
  a.wide = thin_out ? x : dy;
  a.thin = thin_out ? dy : x;
  a.slab = ws;
  a.N = d->N; a.H = d->H; a.W = d->W;
  SRX_REQUIRE((size_t)d->N * d->H * d->W * 256 < 0xfffffff0ull, "conv2d_bwd_weight(thin): activation tensor above 4 GiB (32-bit buffer offsets)");
  a.wsegs = (int)srx_cdiv(d->W, 16);
  a.nseg = d->N * d->H * a.wsegs;
  const int groups = d->KH == 9 ? 3 : 1;
  const int ranges = thin_ranges(a.nseg, groups);
  a.segs_per_range = (int)srx_cdiv(a.nseg, ranges);
  const int nranges = (int)srx_cdiv(a.nseg, a.segs_per_range);
  const int nslabs = (int)srx_cdiv(nranges, 4);
  if ((size_t)nslabs * 4 * d->KH * d->KW * 64 > ws_floats)
    SRX_FAIL(SRX_E_WORKSPACE, "conv2d_bwd_weight(thin): workspace too small");
  a.nranges = nranges;
  const unsigned blocks = (unsigned)(nslabs * groups);
  char nm[64];
  if (srx_prof_on()) snprintf(nm, sizeof(nm), "thin_wgrad_kernel<%d, %d, %d, %d>", d->KH, d->KW, groups, thin_out ? -1 : 1);
  const double fl = 2.0 * d->N * d->H * d->W * d->KH * d->KW * 64 * (thin_out ? d->Cout : d->Cin);
  if (d->KH == 9) {
    if (thin_out) SRX_LAUNCH_PROF(nm, fl, (thin_wgrad_kernel<9, 9, 3, -1>), dim3(blocks), dim3(256), 0, st, a);
    else SRX_LAUNCH_PROF(nm, fl, (thin_wgrad_kernel<9, 9, 3, +1>), dim3(blocks), dim3(256), 0, st, a);
  } else {
    if (thin_out) SRX_LAUNCH_PROF(nm, fl, (thin_wgrad_kernel<3, 3, 1, -1>), dim3(blocks), dim3(256), 0, st, a);
    else SRX_LAUNCH_PROF(nm, fl, (thin_wgrad_kernel<3, 3, 1, +1>), dim3(blocks), dim3(256), 0, st, a);
  }
  SRX_CHECK_LAUNCH("thin_wgrad_kernel");
  const int taps = d->KH * d->KW;
  const int cthin = thin_out ? d->Cout : d->Cin;
  hipLaunchKernelGGL(thin_wgrad_reduce_kernel, dim3((unsigned)(cthin * taps)), dim3(256), 0, st, ws,
                     nslabs, taps, cthin, 64, thin_out ? 1 : 0, dw, accumulate);
  SRX_CHECK_LAUNCH("thin_wgrad_reduce_kernel");
  return SRX_OK;
}

// (the 3-channel layers are fp32 at either precision: their packed layout must not depend on srx_conv2d_t::precision,
// which a trainer switches between its pre-training and GAN phases)
static bool thin_geom_ok(const srx_conv2d_t* d) {
  return d->stride == 1 && !d->shuffle && d->KH == d->KW && (d->KH == 3 || d->KH == 9) && d->pad == (d->KH - 1) / 2 &&
         d->up <= 1;
}
bool srx_thin_fwd_applicable(const srx_conv2d_t* d) {
  return thin_geom_ok(d) && d->Cout <= 4 && d->Cout_s == 4 && d->Cin == 64 && d->Cin_s == 64 && d->act == SRX_ACT_NONE;
}
bool srx_thin_dgrad_applicable(const srx_conv2d_t* d) {
  return thin_geom_ok(d) && d->Cin <= 4 && d->Cin_s == 4 && d->Cout == 64 && d->Cout_s == 64;
}

int srx_thin_pack(const srx_conv2d_t* d, const float* w, float* p, int mode, hipStream_t st) {
  const int taps = d->KH * d->KW;
  hipLaunchKernelGGL(thin_pack_kernel, dim3((unsigned)srx_cdiv(4 * taps * 64, 256)), dim3(256), 0, st, w, p,
                     mode == 0 ? d->Cout : d->Cin, 64, d->KH, d->KW, mode);
  SRX_CHECK_LAUNCH("thin_pack_kernel");
  return SRX_OK;
}

template <int K, int R>
static int launch_thin_fwd2(const ThinF& a, hipStream_t st) {
  constexpr int PH = 4 * R + K - 1, PW = 32 + K - 1;
  const size_t lds = (size_t)(PH * PW * 8 + 4 * K * K * 8) * sizeof(float);
  char nm[64];
  if (srx_prof_on()) snprintf(nm, sizeof(nm), "thin_fwd2_kernel<%d, %d>", K, R);
  SRX_LAUNCH_PROF(nm, 2.0 * a.N * a.H * a.W * K * K * 64 * a.Cout, (thin_fwd2_kernel<K, R>),
                  dim3((unsigned)(a.N * a.tiles_h * a.tiles_w)), dim3(256), lds, st, a);
  SRX_CHECK_LAUNCH("thin_fwd2_kernel");
  return SRX_OK;
}

// in: [N][H][W][64], out: [N][H][W][4]; wpk from srx_thin_pack; n_out = real output channels
bool srx_first3_fwd_applicable(const srx_conv2d_t* d) {
  const bool off = srx_dev().no_first3;  // developer switch (A/B runs), read at load time
  return !off && thin_geom_ok(d) && d->KH == 3 && d->Cin <= 4 && d->Cin_s == 4 && d->Cout == 64 && d->Cout_s == 64 &&
         (int64_t)d->N * d->H * d->W < (1 << 24);
}

int srx_first3_fwd(const srx_conv2d_t* d, const float* in, const float* wpk, int Kp, const float* bias, float* out,
                   hipStream_t st, int out_bf16) {
  First3 a;
  a.in = in; a.w = wpk; a.bias = bias; a.out = out;
  a.H = d->H; a.W = d->W; a.HW = d->H * d->W; a.M = d->N * a.HW; a.Kp = Kp;
  a.ntiles = (int)srx_cdiv(a.M, 32);
  a.inv_HW = 1.0f / (float)a.HW; a.inv_W = 1.0f / (float)a.W;
  a.slope = d->act == SRX_ACT_RELU ? 0.f : (d->act == SRX_ACT_LRELU ? d->slope : 1.f);  // v > 0 ? v : v * slope
  a.in_bytes = (unsigned)((size_t)a.M * 4 * sizeof(float));
  a.out_bytes = (unsigned)((size_t)a.M * 64 * (out_bf16 ? 2 : sizeof(float)));
  // three workgroups per CU, each wave walking its share of the tiles: the weight prologue is paid 768 times, not 2304
  const int cus = srx_plan_cus(), gdev = srx_dev().first3_wgs_per_cu;
  const unsigned grid = (unsigned)std::min<int64_t>(srx_cdiv(a.ntiles, 4), (int64_t)cus * (gdev > 0 ? gdev : 3));
  char nm[64];
  if (srx_prof_on()) snprintf(nm, sizeof(nm), "first3x3_fwd_kernel<%d> MxNxK=%dx64x36", d->precision ? 1 : 0, a.M);
  if (out_bf16) {
    if (!d->precision) SRX_FAIL(SRX_E_UNSUPPORTED, "first3_fwd: a bf16 output needs precision = 1");
    SRX_LAUNCH_PROF(nm, 2.0 * a.M * 64 * 27, (first3x3_fwd_kernel<1, true>), dim3(grid), dim3(256), 0, st, a);
  } else if (d->precision) SRX_LAUNCH_PROF(nm, 2.0 * a.M * 64 * 27, first3x3_fwd_kernel<1>, dim3(grid), dim3(256), 0, st, a);
  else SRX_LAUNCH_PROF(nm, 2.0 * a.M * 64 * 27, first3x3_fwd_kernel<0>, dim3(grid), dim3(256), 0, st, a);
  SRX_CHECK_LAUNCH("first3x3_fwd_kernel");
  return SRX_OK;
}

template <int K, int R, bool IN16 = false>
static int launch_thin_fwd2_bf16(const ThinF& a, hipStream_t st) {
  constexpr int PH = 4 * R + K - 1, PW = 32 + K - 1;
  const size_t lds = (size_t)(PH * PW * 16 + 4 * K * K * 16) * 2;
  static std::once_flag once;
  std::call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&thin_fwd2_bf16_kernel<K, R, IN16>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              96 * 1024);
  });
  char nm[64];
  if (srx_prof_on()) snprintf(nm, sizeof(nm), "thin_fwd2_bf16_kernel<%d, %d, %d>", K, R, (int)IN16);
  SRX_LAUNCH_PROF(nm, 2.0 * a.N * a.H * a.W * K * K * 64 * a.Cout, (thin_fwd2_bf16_kernel<K, R, IN16>),
                  dim3((unsigned)(a.N * a.tiles_h * a.tiles_w)), dim3(256), lds, st, a);
  SRX_CHECK_LAUNCH("thin_fwd2_bf16_kernel");
  return SRX_OK;
}

int srx_thin_fwd(const srx_conv2d_t* d, const float* in, const float* wpk, const float* bias, float* out, int n_out,
                 hipStream_t st, int in_bf16) {
  ThinF a;
  a.in = in; a.w = wpk; a.bias = bias; a.out = out;
  a.N = d->N; a.H = d->H; a.W = d->W; a.Cout = n_out;
  a.tiles_w = (int)srx_cdiv(d->W, 32);
  a.in_bf16 = in_bf16;
  a.in_bytes = (unsigned)((size_t)d->N * d->H * d->W * 64 * (in_bf16 ? 2 : sizeof(float)));
  if (in_bf16 && d->precision != 2) SRX_FAIL(SRX_E_UNSUPPORTED, "thin_fwd: a bf16 input needs precision = 2 (bf16 products)");
  // rows per lane: 6 reads LDS least (23 b128 per 216 MFMAs) and wins whenever there are plenty of tiles; small
  // images balance better over the CUs with 12-row tiles (R = 3).  SRX_THIN_FWD_ROWS overrides (3, 4, 6).
  const int dev = srx_dev().thin_fwd_rows, cus = srx_plan_cus();
  const int64_t big_tiles = (int64_t)d->N * srx_cdiv(d->H, 24) * a.tiles_w;
  const int R = dev > 0 ? dev : (big_tiles >= 4 * cus ? 6 : 3);
  a.tiles_h = (int)srx_cdiv(d->H, 4 * R);
  if (d->precision == 2) {  // bf16 products in the 3-channel layer too (inference)
    if (in_bf16) {
      if (d->KH == 9) return R == 3 ? launch_thin_fwd2_bf16<9, 3, true>(a, st) : launch_thin_fwd2_bf16<9, 6, true>(a, st);
      return R == 3 ? launch_thin_fwd2_bf16<3, 3, true>(a, st) : launch_thin_fwd2_bf16<3, 6, true>(a, st);
    }
    if (d->KH == 9) return R == 3 ? launch_thin_fwd2_bf16<9, 3>(a, st) : launch_thin_fwd2_bf16<9, 6>(a, st);
    return R == 3 ? launch_thin_fwd2_bf16<3, 3>(a, st) : launch_thin_fwd2_bf16<3, 6>(a, st);
  }
  if (d->KH == 9) return R == 3 ? launch_thin_fwd2<9, 3>(a, st) : (R == 6 ? launch_thin_fwd2<9, 6>(a, st) : launch_thin_fwd2<9, 4>(a, st));
  return R == 3 ? launch_thin_fwd2<3, 3>(a, st) : (R == 6 ? launch_thin_fwd2<3, 6>(a, st) : launch_thin_fwd2<3, 4>(a, st));
}
