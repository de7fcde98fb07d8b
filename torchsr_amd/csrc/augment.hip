// On-device training-data pipeline: what TrainData.__getitem__ does on the
// host with PIL for every sample -- RandomCrop, RandomHorizontalFlip, RandomVerticalFlip, bicubic x1/4
// (torchsr/dataset.py:88-99,121-125) -- as two HBM-bound kernels over a batch of decoded uint8 images
// that already live in device memory.
#include "srx_common.h"

namespace {

// out[n][c][y][x] = img_n[top + (vflip ? crop-1-y : y)][left + (hflip ? crop-1-x : x)][c] / 255
// meta[n] = {H, W, top, left, hflip, vflip}; images are HWC uint8 with 3 channels
__global__ void crop_flip_u8_kernel(const uint8_t* const* __restrict__ imgs, const int* __restrict__ meta,
                                    float* __restrict__ out, int N, int crop) {
  const int64_t total = (int64_t)N * 3 * crop * crop;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % crop);
    int64_t t = i / crop;
    const int y = (int)(t % crop);
    t /= crop;
    const int c = (int)(t % 3);
    const int n = (int)(t / 3);
    const int* m = meta + 6 * n;
    const int W = m[1];
    const int sy = m[2] + (m[5] ? crop - 1 - y : y);
    const int sx = m[3] + (m[4] ? crop - 1 - x : x);
    out[i] = (float)imgs[n][((size_t)sy * W + sx) * 3 + c] / 255.0f;  // ToTensor: .div(255)
  }
}

// Keys bicubic, a = -0.5 (PIL's BICUBIC and torch's antialiased bicubic)
__device__ __forceinline__ float bicubic_w(float x) {
  x = fabsf(x);
  if (x < 1.f) return ((1.5f * x - 2.5f) * x) * x + 1.f;
  if (x < 2.f) return (((x - 5.f) * x + 8.f) * x - 4.f) * -0.5f;
  return 0.f;
}

// Antialiased bicubic reduction by an integer factor s (PIL ImagingResample / torch antialias=True):
// output pixel o averages the inputs in [centre - 2s, centre + 2s), centre = (o + 0.5) s, with weights
// w((x + 0.5 - centre) / s), taps outside the image dropped and the rest renormalised.
__global__ void bicubic_down_kernel(const float* __restrict__ in, float* __restrict__ out, int planes, int H, int W,
                                    int s, int quantize) {
  const int Ho = H / s, Wo = W / s;
  const int64_t total = (int64_t)planes * Ho * Wo;
  const float inv_s = 1.0f / (float)s;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ox = (int)(i % Wo);
    int64_t t = i / Wo;
    const int oy = (int)(t % Ho);
    const int p = (int)(t / Ho);
    const float cy = (oy + 0.5f) * s, cx = (ox + 0.5f) * s;
    const int y0 = max(0, (int)(cy - 2.f * s + 0.5f)), y1 = min(H, (int)(cy + 2.f * s + 0.5f));
    const int x0 = max(0, (int)(cx - 2.f * s + 0.5f)), x1 = min(W, (int)(cx + 2.f * s + 0.5f));
    const float* src = in + (size_t)p * H * W;
    float acc = 0.f, wsum_y = 0.f, wsum_x = 0.f;
    for (int x = x0; x < x1; ++x) wsum_x += bicubic_w((x + 0.5f - cx) * inv_s);
    for (int y = y0; y < y1; ++y) {
      const float wy = bicubic_w((y + 0.5f - cy) * inv_s);
      wsum_y += wy;
      float row = 0.f;
      for (int x = x0; x < x1; ++x) row += bicubic_w((x + 0.5f - cx) * inv_s) * src[(size_t)y * W + x];
      acc += wy * row;
    }
    float v = acc / (wsum_x * wsum_y);
    if (quantize) v = rintf(fminf(fmaxf(v, 0.f), 1.f) * 255.f) * (1.0f / 255.0f);  // the 8-bit LR image PIL returns
    out[i] = v;
  }
}

unsigned grid_for(int64_t n) {
  int64_t b = (n + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > 65535 ? 65535 : b));
}

}  // namespace

extern "C" int srx_crop_flip_u8(const void* const* imgs, const int32_t* meta, float* out_nchw, int N, int crop,
                                void* stream) {
  SRX_REQUIRE(imgs && meta && out_nchw && N > 0 && crop > 0, "crop_flip_u8: bad argument");
  hipLaunchKernelGGL(crop_flip_u8_kernel, dim3(grid_for((int64_t)N * 3 * crop * crop)), dim3(256), 0, srx_stream(stream),
                     reinterpret_cast<const uint8_t* const*>(imgs), meta, out_nchw, N, crop);
  SRX_CHECK_LAUNCH("crop_flip_u8_kernel");
  return SRX_OK;
}

extern "C" int srx_bicubic_down(const float* in_nchw, float* out_nchw, int N, int C, int H, int W, int scale,
                                int quantize, void* stream) {
  SRX_REQUIRE(in_nchw && out_nchw && N > 0 && C > 0 && scale >= 1 && H >= scale && W >= scale,
              "bicubic_down: bad argument");
  hipLaunchKernelGGL(bicubic_down_kernel, dim3(grid_for((int64_t)N * C * (H / scale) * (W / scale))), dim3(256), 0,
                     srx_stream(stream), in_nchw, out_nchw, N * C, H, W, scale, quantize);
  SRX_CHECK_LAUNCH("bicubic_down_kernel");
  return SRX_OK;
}
