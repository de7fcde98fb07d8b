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
#include <cstdio>

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
  const int lane = threadIdx.x & 63;
  const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int group = gw % GROUPS, range = gw / GROUPS;
  if (range >= a.nranges) return;  // grid is rounded up to whole workgroups
  const int sbeg = range * a.segs_per_range;
  const int send = min(a.nseg, sbeg + a.segs_per_range);
  const int tc = lane & 3;

  f32x4 acc[RPG * KW];
#pragma unroll
  for (int t = 0; t < RPG * KW; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int s = sbeg; s < send; ++s) {
    const int cb = s % a.wsegs;
    const int row = (s / a.wsegs) % a.H;
    const int n = s / (a.wsegs * a.H);
    const int c0 = cb * 16;
    float wv[16];
    const float* wp = a.wide + ((size_t)(n * a.H + row) * a.W + c0) * 64 + lane;
#pragma unroll
    for (int j = 0; j < 16; ++j) wv[j] = (c0 + j < a.W) ? wp[j * 64] : 0.f;
#pragma unroll
    for (int dr = 0; dr < RPG; ++dr) {
      const int kh = group * RPG + dr;
      const int tr = row + SGN * (kh - PAD);
      const bool rok = (unsigned)tr < (unsigned)a.H;
      float tv[NT];
      const float* tp = a.thin + ((size_t)(n * a.H + (rok ? tr : 0)) * a.W) * 4 + tc;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int col = c0 - PAD + t;
        const bool ok = rok && (unsigned)col < (unsigned)a.W;
        tv[t] = ok ? tp[(ok ? col : 0) * 4] : 0.f;
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
  float* o = a.slab + (size_t)range * (4 * KH * KW * 64);
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
};

template <int KH, int KW>
__global__ __launch_bounds__(256) void thin_fwd_kernel(const ThinF a) {
  constexpr int TH = 16, TW = 32;  // output pixels per workgroup: 4 waves x (2 groups of 64 = 2 rows x 32)
  constexpr int PH = TH + KH - 1, PW = TW + KW - 1;
  constexpr int PAD = (KH - 1) / 2;
  constexpr int CS = 20;  // floats per pixel slot in LDS (16 channels + 4 pad)
  constexpr int TAPS = KH * KW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sx = reinterpret_cast<float*>(smem);         // [PH][PW][CS]
  float* sw = sx + PH * PW * CS;                      // [4][TAPS][16]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int b = blockIdx.x;
  const int tw_i = b % a.tiles_w; b /= a.tiles_w;
  const int th_i = b % a.tiles_h;
  const int n = b / a.tiles_h;
  const int h0 = th_i * TH, w0 = tw_i * TW;
  // this lane's two pixels: group g -> rows (wave*4 + 2g + lane/32), column lane%32
  const int pc = lane & 31, pr0 = wave * 4 + (lane >> 5);
  const int wc = lane & 3;

  f32x4 acc[2];
  acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
  acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int cc = 0; cc < 64; cc += 16) {
    __syncthreads();
    // stage the input patch slice: PH*PW pixels x 16 channels = 4 float4 per pixel
    for (int i = tid; i < PH * PW * 4; i += 256) {
      const int q = i & 3, pix = i >> 2;
      const int pw = pix % PW, ph = pix / PW;
      const int ih = h0 - PAD + ph, iw = w0 - PAD + pw;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if ((unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W)
        v = *reinterpret_cast<const f32x4*>(a.in + ((size_t)(n * a.H + ih) * a.W + iw) * 64 + cc + q * 4);
      *reinterpret_cast<f32x4*>(sx + pix * CS + q * 4) = v;
    }
    for (int i = tid; i < 4 * TAPS * 4; i += 256) {
      const int q = i & 3, ct = i >> 2;  // ct = c*TAPS + tap
      *reinterpret_cast<f32x4*>(sw + ct * 16 + q * 4) =
          *reinterpret_cast<const f32x4*>(a.w + (size_t)ct * 64 + cc + q * 4);
    }
    __syncthreads();
#pragma unroll 1
    for (int kh = 0; kh < KH; ++kh) {
#pragma unroll
      for (int kw = 0; kw < KW; ++kw) {
        const float* wp = sw + (wc * TAPS + kh * KW + kw) * 16;
        const float* x0 = sx + ((pr0 + kh) * PW + pc + kw) * CS;
        const float* x1 = x0 + 2 * PW * CS;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 wv = *reinterpret_cast<const f32x4*>(wp + q * 4);
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(x0 + q * 4);
          const f32x4 v1 = *reinterpret_cast<const f32x4*>(x1 + q * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[e], v0[e], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[e], v1[e], acc[1], 0, 0, 0);
          }
        }
      }
    }
  }
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int oh = h0 + pr0 + 2 * g, ow = w0 + pc;
    if (oh < a.H && ow < a.W) {
      f32x4 v = acc[g];
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = c < a.Cout ? v[c] + (a.bias ? a.bias[c] : 0.f) : 0.f;
      *reinterpret_cast<f32x4*>(a.out + ((size_t)(n * a.H + oh) * a.W + ow) * 4) = v;
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
  int cus = srx_device_cus();
  if (cus <= 0) cus = 256;
  int ranges = (cus * 8) / groups;  // about two waves per SIMD: one's loads hide under the other's MFMAs
  if (ranges > nseg) ranges = nseg;
  if (ranges < 1) ranges = 1;
  return ranges;
}

}  // namespace

// ---- internal entry points used by gconv.hip's C ABI functions ------------------------------------
bool srx_thin_wgrad_applicable(const srx_conv2d_t* d) {
  if (d->stride != 1 || d->shuffle || d->KH != d->KW || (d->KH != 3 && d->KH != 9) || d->pad != (d->KH - 1) / 2)
    return false;
  const bool thin_out = d->Cout <= 4 && d->Cout_s == 4 && d->Cin == 64 && d->Cin_s == 64;
  const bool thin_in = d->Cin <= 4 && d->Cin_s == 4 && d->Cout == 64 && d->Cout_s == 64;
  return thin_out || thin_in;
}

size_t srx_thin_wgrad_ws_floats(const srx_conv2d_t* d) {
  const int nseg = d->N * d->H * (int)srx_cdiv(d->W, 16);
  const int groups = d->KH == 9 ? 3 : 1;
  return (size_t)thin_ranges(nseg, groups) * 4 * d->KH * d->KW * 64;
}

int srx_thin_wgrad(const srx_conv2d_t* d, const float* x, const float* dy, float* dw, int accumulate, float* ws,
                   size_t ws_floats, hipStream_t st) {
  const bool thin_out = d->Cout <= 4 && d->Cout_s == 4;
  ThinW a;
  a.wide = thin_out ? x : dy;
  a.thin = thin_out ? dy : x;
  a.slab = ws;
  a.N = d->N; a.H = d->H; a.W = d->W;
  a.wsegs = (int)srx_cdiv(d->W, 16);
  a.nseg = d->N * d->H * a.wsegs;
  const int groups = d->KH == 9 ? 3 : 1;
  const int ranges = thin_ranges(a.nseg, groups);
  a.segs_per_range = (int)srx_cdiv(a.nseg, ranges);
  const int nranges = (int)srx_cdiv(a.nseg, a.segs_per_range);
  if ((size_t)nranges * 4 * d->KH * d->KW * 64 > ws_floats)
    SRX_FAIL(SRX_E_WORKSPACE, "conv2d_bwd_weight(thin): workspace too small");
  a.nranges = nranges;
  const unsigned blocks = (unsigned)srx_cdiv((int64_t)nranges * groups, 4);
  if (srx_prof_on()) {
    char nm[64];
    snprintf(nm, sizeof(nm), "thin_wgrad_kernel<%d, %d, %d, %d>", d->KH, d->KW, groups, thin_out ? -1 : 1);
    srx_prof_begin_launch(nm, 2.0 * d->N * d->H * d->W * d->KH * d->KW * 64 * (thin_out ? d->Cout : d->Cin), st);
  }
  if (d->KH == 9) {
    if (thin_out) hipLaunchKernelGGL((thin_wgrad_kernel<9, 9, 3, -1>), dim3(blocks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((thin_wgrad_kernel<9, 9, 3, +1>), dim3(blocks), dim3(256), 0, st, a);
  } else {
    if (thin_out) hipLaunchKernelGGL((thin_wgrad_kernel<3, 3, 1, -1>), dim3(blocks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((thin_wgrad_kernel<3, 3, 1, +1>), dim3(blocks), dim3(256), 0, st, a);
  }
  if (srx_prof_on()) srx_prof_end_launch(st);
  SRX_CHECK_LAUNCH("thin_wgrad_kernel");
  const int taps = d->KH * d->KW;
  const int cthin = thin_out ? d->Cout : d->Cin;
  hipLaunchKernelGGL(thin_wgrad_reduce_kernel, dim3((unsigned)(cthin * taps)), dim3(256), 0, st, ws,
                     nranges, taps, cthin, 64, thin_out ? 1 : 0, dw, accumulate);
  SRX_CHECK_LAUNCH("thin_wgrad_reduce_kernel");
  return SRX_OK;
}

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

template <int K>
static int launch_thin_fwd(const ThinF& a, hipStream_t st) {
  constexpr int PH = 16 + K - 1, PW = 32 + K - 1;
  const size_t lds = (size_t)(PH * PW * 20 + 4 * K * K * 16) * sizeof(float);
  static bool once = false;
  if (!once) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&thin_fwd_kernel<K, K>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    once = true;
  }
  if (srx_prof_on()) {
    char nm[64];
    snprintf(nm, sizeof(nm), "thin_fwd_kernel<%d, %d>", K, K);
    srx_prof_begin_launch(nm, 2.0 * a.N * a.H * a.W * K * K * 64 * a.Cout, st);
  }
  hipLaunchKernelGGL((thin_fwd_kernel<K, K>), dim3((unsigned)(a.N * a.tiles_h * a.tiles_w)), dim3(256), lds, st, a);
  if (srx_prof_on()) srx_prof_end_launch(st);
  SRX_CHECK_LAUNCH("thin_fwd_kernel");
  return SRX_OK;
}

// in: [N][H][W][64], out: [N][H][W][4]; wpk from srx_thin_pack; n_out = real output channels
int srx_thin_fwd(const srx_conv2d_t* d, const float* in, const float* wpk, const float* bias, float* out, int n_out,
                 hipStream_t st) {
  ThinF a;
  a.in = in; a.w = wpk; a.bias = bias; a.out = out;
  a.N = d->N; a.H = d->H; a.W = d->W; a.Cout = n_out;
  a.tiles_h = (int)srx_cdiv(d->H, 16);
  a.tiles_w = (int)srx_cdiv(d->W, 32);
  return d->KH == 9 ? launch_thin_fwd<9>(a, st) : launch_thin_fwd<3>(a, st);
}
