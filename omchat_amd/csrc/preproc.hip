// Image front-end on the GPU: the step immediately before the ViT.
//
// Replaces, bit for bit, what the reference does on one CPU thread per image:
//   process_anyres_image (omchat/mm_utils.py:119-158): select_best_resolution (:12-39), resize_and_pad_image (:42-74,
//   PIL Image.resize = BICUBIC, pasted centred on a black canvas), divide_to_patches (:77-96), plus the 448x448 thumbnail;
//   CLIPImageProcessor(size=448, crop=448, rescale 1/255, normalize mean/std) per tile (internVIT_encoder.py:25-29) -- on a
//   448x448 tile resize and centre-crop are identities, so what is left is rescale + normalize + HWC->CHW + dtype cast.
// Pillow's 8-bit resampler (src/libImaging/Resample.c; third-party, not vendored in the reference) is restated here:
//   coefficients in double exactly as precompute_coeffs / normalize_coeffs_8bpc (fixed point, 22 fractional bits), two
//   separable passes (horizontal, then vertical; a pass whose size does not change is skipped) with int32 accumulation,
//   +2^21 rounding and clip to [0, 255].  Integer arithmetic -> results are identical to Pillow's, not merely close.
// rescale + normalize collapse into a 3 x 256 table built with the reference's float32/float64 rounding order.
//
// Byte work, HBM/latency bound and tiny (a 12 MP photo is 36 MB): one thread per output pixel, coalesced over x; the
// whole front-end is < 100 us on the device versus tens of ms in PIL.
#include "kernels.h"
#include "../../include/omchat_hip.h"
#include <math.h>
#include <vector>

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

double bicubic(double x) {            // Resample.c bicubic_filter, a = -0.5
  const double a = -0.5;
  if (x < 0.0) x = -x;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}

struct Coeffs {
  int ksize = 0;
  std::vector<int> bounds;   // [out][2] = first source index, tap count
  std::vector<int> kk;       // [out][ksize] fixed-point taps
};

void precompute(int in_size, int out_size, Coeffs& c) {
  double scale, filterscale;
  scale = filterscale = (double)in_size / out_size;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = 2.0 * filterscale;
  c.ksize = (int)ceil(support) * 2 + 1;
  c.bounds.assign((size_t)out_size * 2, 0);
  c.kk.assign((size_t)out_size * c.ksize, 0);
  std::vector<double> k(c.ksize);
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = (xx + 0.5) * scale;
    double ww = 0.0;
    const double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    for (int x = 0; x < c.ksize; ++x) k[x] = 0.0;
    for (int x = 0; x < xmax; ++x) {
      const double w = bicubic((x + xmin - center + 0.5) * ss);
      k[x] = w;
      ww += w;
    }
    for (int x = 0; x < xmax; ++x)
      if (ww != 0.0) k[x] /= ww;
    c.bounds[2 * xx] = xmin;
    c.bounds[2 * xx + 1] = xmax;
    for (int x = 0; x < c.ksize; ++x) {
      const double v = k[x] * (1 << PRECISION_BITS);
      c.kk[(size_t)xx * c.ksize + x] = k[x] < 0 ? (int)(-0.5 + v) : (int)(0.5 + v);
    }
  }
}

__device__ __forceinline__ unsigned char clip8(int v) {
  v >>= PRECISION_BITS;
  return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// horizontal pass: in [H][W][3] -> out [H][OW][3]
__global__ __launch_bounds__(256) void resample_h_kernel(const unsigned char* __restrict__ in, unsigned char* __restrict__ out, int H, int W, int OW,
                                                         const int* __restrict__ bounds, const int* __restrict__ kk, int ksize) {
  const int ox = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (ox >= OW) return;
  const int xmin = bounds[2 * ox], n = bounds[2 * ox + 1];
  const int* k = kk + (size_t)ox * ksize;
  const unsigned char* row = in + ((size_t)y * W + xmin) * 3;
  int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
  for (int x = 0; x < n; ++x) {
    const int w = k[x];
    s0 += row[3 * x] * w; s1 += row[3 * x + 1] * w; s2 += row[3 * x + 2] * w;
  }
  unsigned char* o = out + ((size_t)y * OW + ox) * 3;
  o[0] = clip8(s0); o[1] = clip8(s1); o[2] = clip8(s2);
}

// vertical pass: in [H][W][3] -> out [OH][W][3]
__global__ __launch_bounds__(256) void resample_v_kernel(const unsigned char* __restrict__ in, unsigned char* __restrict__ out, int H, int W, int OH,
                                                         const int* __restrict__ bounds, const int* __restrict__ kk, int ksize) {
  const int xb = blockIdx.x * 256 + threadIdx.x, oy = blockIdx.y;     // xb indexes bytes of a row (3 per pixel): fully coalesced
  if (xb >= W * 3) return;
  const int ymin = bounds[2 * oy], n = bounds[2 * oy + 1];
  const int* k = kk + (size_t)oy * ksize;
  int s = 1 << (PRECISION_BITS - 1);
  for (int y = 0; y < n; ++y) s += in[(size_t)(ymin + y) * W * 3 + xb] * k[y];
  out[(size_t)oy * W * 3 + xb] = clip8(s);
}

// tiles: t = 0 thumbnail [tile][tile][3]; t >= 1: tile (ty, tx) of the best-resolution canvas = resized image [nh][nw][3]
// pasted at (x0, y0) on black.  out [n][3][tile][tile] in the model dtype through the normalisation table lut[3][256].
template <typename T>
__global__ __launch_bounds__(256) void tiles_kernel(const unsigned char* __restrict__ thumb, const unsigned char* __restrict__ img, int nw, int nh,
                                                    int x0, int y0, int tiles_x, int tile, const float* __restrict__ lut, T* __restrict__ out) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, t = blockIdx.z;
  if (x >= tile) return;
  unsigned char px[3] = {0, 0, 0};
  const int first = thumb ? 1 : 0;          // no thumbnail: tile t is canvas tile t
  if (t < first) {
    const unsigned char* p = thumb + ((size_t)y * tile + x) * 3;
    px[0] = p[0]; px[1] = p[1]; px[2] = p[2];
  } else {
    const int ty = (t - first) / tiles_x, tx = (t - first) % tiles_x;
    const int cx = tx * tile + x - x0, cy = ty * tile + y - y0;
    if (cx >= 0 && cx < nw && cy >= 0 && cy < nh) {
      const unsigned char* p = img + ((size_t)cy * nw + cx) * 3;
      px[0] = p[0]; px[1] = p[1]; px[2] = p[2];
    }
  }
  const size_t plane = (size_t)tile * tile;
  T* o = out + (size_t)t * 3 * plane + (size_t)y * tile + x;
#pragma unroll
  for (int c = 0; c < 3; ++c) o[c * plane] = fromf<T>(lut[c * 256 + px[c]]);
}

struct DevBuf {       // stream-ordered scratch, released on the same stream when the call returns
  void* p = nullptr; hipStream_t s;
  explicit DevBuf(hipStream_t s_) : s(s_) {}
  ~DevBuf() { if (p) (void)hipFreeAsync(p, s); }
  hipError_t alloc(size_t n) { return hipMallocAsync(&p, n ? n : 1, s); }
};

// PIL Image.resize((ow, oh)) of a device image [H][W][3]; result in `dst` [oh][ow][3] (device).  Identity sizes copy.
int resize_device(const unsigned char* src, int W, int H, unsigned char* dst, int ow, int oh, hipStream_t s) {
  const unsigned char* cur = src;
  int cw = W;
  DevBuf tmp(s), bh(s), kh(s), bv(s), kv(s);
  Coeffs ch, cv;      // host tables must outlive the async copies: pageable hipMemcpyAsync stages before returning
  if (W != ow) {
    precompute(W, ow, ch);
    unsigned char* hout = dst;
    if (H != oh) { OM_HIP(tmp.alloc((size_t)H * ow * 3)); hout = (unsigned char*)tmp.p; }
    OM_HIP(bh.alloc(ch.bounds.size() * 4)); OM_HIP(kh.alloc(ch.kk.size() * 4));
    OM_HIP(hipMemcpyAsync(bh.p, ch.bounds.data(), ch.bounds.size() * 4, hipMemcpyHostToDevice, s));
    OM_HIP(hipMemcpyAsync(kh.p, ch.kk.data(), ch.kk.size() * 4, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(resample_h_kernel, dim3(cdiv(ow, 256), H), dim3(256), 0, s, cur, hout, H, W, ow, (const int*)bh.p, (const int*)kh.p, ch.ksize);
    OM_LAUNCH_CHECK();
    cur = hout; cw = ow;
  }
  if (H != oh) {
    precompute(H, oh, cv);
    OM_HIP(bv.alloc(cv.bounds.size() * 4)); OM_HIP(kv.alloc(cv.kk.size() * 4));
    OM_HIP(hipMemcpyAsync(bv.p, cv.bounds.data(), cv.bounds.size() * 4, hipMemcpyHostToDevice, s));
    OM_HIP(hipMemcpyAsync(kv.p, cv.kk.data(), cv.kk.size() * 4, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(resample_v_kernel, dim3(cdiv(cw * 3, 256), oh), dim3(256), 0, s, cur, dst, H, cw, oh, (const int*)bv.p, (const int*)kv.p, cv.ksize);
    OM_LAUNCH_CHECK();
  } else if (W == ow) {
    OM_HIP(hipMemcpyAsync(dst, src, (size_t)H * W * 3, hipMemcpyDeviceToDevice, s));
  }
  // the tables are pageable host memory: make sure the staged copies are done before the vectors die
  OM_HIP(hipStreamSynchronize(s));
  return 0;
}

void padded_size(int ow, int oh, int tw, int th, int* nw, int* nh) {     // mm_utils.py:55-66
  const double sw = (double)tw / ow, sh = (double)th / oh;
  if (sw < sh) { *nw = tw; int h = (int)ceil(oh * sw); *nh = h < th ? h : th; }
  else { *nh = th; int w = (int)ceil(ow * sh); *nw = w < tw ? w : tw; }
}

}  // namespace

extern "C" int omchat_resample_coeffs(int in_size, int out_size, int* ksize, int* bounds, int* kk, int kk_cap) {
  OM_CHECK(in_size > 0 && out_size > 0, "sizes must be positive");
  Coeffs c;
  precompute(in_size, out_size, c);
  *ksize = c.ksize;
  if (bounds) for (size_t i = 0; i < c.bounds.size(); ++i) bounds[i] = c.bounds[i];
  if (kk) {
    OM_CHECK((size_t)kk_cap >= c.kk.size(), "kk buffer too small");
    for (size_t i = 0; i < c.kk.size(); ++i) kk[i] = c.kk[i];
  }
  return 0;
}

extern "C" int omchat_normalize_lut(const float mean[3], const float std_[3], float lut[768]) {
  // transformers image_transforms: rescale = float32(float64(u) * (1/255)); normalize = (x - mean32) / std32 in float32
  for (int c = 0; c < 3; ++c)
    for (int u = 0; u < 256; ++u) {
      const float x = (float)((double)u * (1.0 / 255));
      lut[c * 256 + u] = (x - mean[c]) / std_[c];
    }
  return 0;
}

extern "C" int omchat_preproc_plan(int W, int H, const int* pinpoints, int n_pin, int tile, int* best_w, int* best_h, int* n_tiles) {
  OM_CHECK(W > 0 && H > 0 && n_pin > 0 && tile > 0, "bad arguments");
  // select_best_resolution (mm_utils.py:12-39): maximise the effective resolution, then minimise waste; first wins ties
  int bw = 0, bh = 0;
  long long best_eff = 0, best_waste = -1;
  for (int i = 0; i < n_pin; ++i) {
    const int w = pinpoints[2 * i], h = pinpoints[2 * i + 1];
    const double sw = (double)w / W, sh = (double)h / H, sc = sw < sh ? sw : sh;
    const long long dw = (long long)(W * sc), dh = (long long)(H * sc);
    long long eff = dw * dh; const long long orig = (long long)W * H;
    if (orig < eff) eff = orig;
    const long long waste = (long long)w * h - eff;
    if (eff > best_eff || (eff == best_eff && (best_waste < 0 || waste < best_waste))) { bw = w; bh = h; best_eff = eff; best_waste = waste; }
  }
  OM_CHECK(bw > 0 && bw % tile == 0 && bh % tile == 0, "grid pinpoints must be multiples of the tile edge");
  *best_w = bw; *best_h = bh; *n_tiles = 1 + (bw / tile) * (bh / tile);
  return 0;
}

// resize to (rw, rh), paste at (x0, y0) on a black (cw x ch) canvas, cut row-major tiles, optional thumbnail first
static int preproc_tiles(int dtype, const void* rgb, int rgb_on_device, int W, int H, int cw, int ch, int rw, int rh, int x0, int y0, int tile,
                         int thumbnail, const float mean[3], const float std_[3], void* pixels_out, hipStream_t s) {
  OM_CHECK(rgb && pixels_out && W > 0 && H > 0, "null image");
  OM_CHECK(cw > 0 && ch > 0 && cw % tile == 0 && ch % tile == 0, "canvas must be a multiple of the tile edge");
  OM_CHECK(rw > 0 && rh > 0 && x0 >= 0 && y0 >= 0 && x0 + rw <= cw && y0 + rh <= ch, "resized image must fit the canvas");
  OM_CHECK(dtype == OMCHAT_F16 || dtype == OMCHAT_BF16 || dtype == OMCHAT_F32, "bad dtype");
  const int tiles_x = cw / tile, n = (thumbnail ? 1 : 0) + tiles_x * (ch / tile);
  DevBuf src(s), thumb(s), img(s), lut(s);
  const unsigned char* d_src = (const unsigned char*)rgb;
  if (!rgb_on_device) {
    OM_HIP(src.alloc((size_t)W * H * 3));
    OM_HIP(hipMemcpyAsync(src.p, rgb, (size_t)W * H * 3, hipMemcpyHostToDevice, s));
    d_src = (const unsigned char*)src.p;
  }
  OM_HIP(img.alloc((size_t)rw * rh * 3));
  OM_HIP(lut.alloc(768 * 4));
  float h_lut[768];
  omchat_normalize_lut(mean, std_, h_lut);
  OM_HIP(hipMemcpyAsync(lut.p, h_lut, sizeof(h_lut), hipMemcpyHostToDevice, s));
  if (thumbnail) {
    OM_HIP(thumb.alloc((size_t)tile * tile * 3));
    if (int rc = resize_device(d_src, W, H, (unsigned char*)thumb.p, tile, tile, s)) return rc;
  }
  if (int rc = resize_device(d_src, W, H, (unsigned char*)img.p, rw, rh, s)) return rc;
  OM_HIP(hipStreamSynchronize(s));       // h_lut / staged host image are consumed
  dim3 grid(cdiv(tile, 256), tile, n);
  const unsigned char* th = thumbnail ? (const unsigned char*)thumb.p : nullptr;
  if (dtype == OMCHAT_F16)
    hipLaunchKernelGGL(tiles_kernel<f16>, grid, dim3(256), 0, s, th, (const unsigned char*)img.p, rw, rh, x0, y0, tiles_x, tile, (const float*)lut.p, (f16*)pixels_out);
  else if (dtype == OMCHAT_BF16)
    hipLaunchKernelGGL(tiles_kernel<bf16>, grid, dim3(256), 0, s, th, (const unsigned char*)img.p, rw, rh, x0, y0, tiles_x, tile, (const float*)lut.p, (bf16*)pixels_out);
  else
    hipLaunchKernelGGL(tiles_kernel<float>, grid, dim3(256), 0, s, th, (const unsigned char*)img.p, rw, rh, x0, y0, tiles_x, tile, (const float*)lut.p, (float*)pixels_out);
  OM_LAUNCH_CHECK();
  return 0;
}

extern "C" int omchat_preproc_anyres(int dtype, const void* rgb, int rgb_on_device, int W, int H, int best_w, int best_h, int tile,
                                     const float mean[3], const float std_[3], void* pixels_out, void* stream) {
  OM_CHECK(W > 0 && H > 0 && best_w > 0 && best_h > 0 && tile > 0 && best_w % tile == 0 && best_h % tile == 0,
           "best resolution must be a positive multiple of the tile edge");
  int nw, nh;
  padded_size(W, H, best_w, best_h, &nw, &nh);
  return preproc_tiles(dtype, rgb, rgb_on_device, W, H, best_w, best_h, nw, nh, (best_w - nw) / 2, (best_h - nh) / 2, tile, 1, mean, std_, pixels_out,
                       (hipStream_t)stream);
}

extern "C" int omchat_preproc_dynamic(int dtype, const void* rgb, int rgb_on_device, int W, int H, int grid_w, int grid_h, int tile, int thumbnail,
                                      const float mean[3], const float std_[3], void* pixels_out, void* stream) {
  OM_CHECK(grid_w > 0 && grid_h > 0 && tile > 0, "bad grid");
  // dynamic_preprocess (mm_utils.py:276-312): plain resize to the grid (aspect NOT preserved), row-major tiles, thumbnail first
  return preproc_tiles(dtype, rgb, rgb_on_device, W, H, grid_w * tile, grid_h * tile, grid_w * tile, grid_h * tile, 0, 0, tile, thumbnail ? 1 : 0, mean, std_,
                       pixels_out, (hipStream_t)stream);
}
