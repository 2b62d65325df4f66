// blur2d.hip — 2-D blur with reflective ('reflect' = half-sample symmetric) boundary for gfx950.
//
// Replaces scipy.ndimage.convolve(X.reshape(nx,ny), PSF, mode='reflect') and its flipped-PSF "transpose"
// (trips/test_problems/Deblurring2D.py:66-73):
//     y[i,j] = sum_{a,b} PSF[a,b] * xr[i + kh/2 - a, j + kw/2 - b]
// written here as a correlation  y[i,j] = sum_{a',b'} c[a',b'] * xr[i - T + a', j - L + b'],
//     c[a',b'] = PSF[kh-1-a', kw-1-b'],  T = kh-1-kh/2,  L = kw-1-kw/2.
//
// Kernel plan (HBM-bound: 8 bytes of traffic per pixel):
//   * k_blur_tile<KH,KW,SEP>: one workgroup (256 threads = 4 waves) per 32 x 128 output tile.  The tile plus halo is
//     staged in LDS with 16-byte coalesced row loads (scalar reflect path only on border tiles).  Rank-1 PSFs (every
//     Gaussian PSF) run a separable row pass LDS->LDS then a column pass; general PSFs run the direct KH*KW form.  Each
//     thread owns a 4 x 4 register block of outputs and reads LDS only with 16-byte ds_read_b128.  Stores are 16-byte
//     coalesced.  Optionally the tile's sum(y^2) is reduced (wave64 shuffles, fp64) into one partial per workgroup.
//   * workgroup -> tile map is XCD-aware: ids are dealt round-robin over the 8 XCDs, so id%8 picks one of 8 contiguous
//     bands of tiles and vertically adjacent tiles (which share halo rows) stay in one XCD's L2.
//   * k_blur_generic: any PSF size (even, rectangular, longer than the image: repeated reflection), no tiling.
#include "trk_internal.h"

#include <cmath>
#include <vector>

using namespace trk;

namespace {

constexpr int NT = 256;
constexpr int TW = 128;  // output tile width  (32 lanes x float4)
constexpr int TH = 32;   // output tile height (8 thread rows x 4)

__host__ __device__ constexpr int rup4(int v) { return (v + 3) & ~3; }

__device__ __forceinline__ int reflect(int i, int n) {
  // half-sample symmetric extension, any distance
  const int p = 2 * n;
  i %= p;
  if (i < 0) i += p;
  return (i >= n) ? (p - 1 - i) : i;
}

struct BlurImpl {
  int nx, ny, kh, kw;
  bool separable;
  bool tiled;          // a k_blur_tile instantiation exists for (kh,kw)
  float* w_dev[2];     // [kh*kw] correlation weights: 0 forward, 1 "transpose" (flipped PSF)
  float* sep_dev[2];   // [kw row weights | kh column weights]
};

// ------------------------------------------------------------------------------------------------ tiled kernel
template <int KH, int KW, bool SEP, bool SUMSQ>
__global__ __launch_bounds__(NT) void k_blur_tile(const float* __restrict__ x, int64_t ldx, float* __restrict__ y,
                                                  int64_t ldy, int nx, int ny, const float* __restrict__ wts,
                                                  double* __restrict__ partials, int tiles_x, int tiles_y) {
  constexpr int T = KH - 1 - KH / 2;   // halo above
  constexpr int L = KW - 1 - KW / 2;   // halo left
  constexpr int R = KW / 2;            // halo right
  constexpr int LP = rup4(L), RP = rup4(R);
  constexpr int SW = TW + LP + RP;     // staged row length (floats, multiple of 4)
  constexpr int SH = TH + KH - 1;      // staged rows
  constexpr int OFF = LP - L;          // first needed column inside the 16-byte aligned read
  constexpr int NV = (OFF + KW + 3 + 3) / 4;  // float4s covering 4 outputs' taps
  constexpr int SW4 = SW / 4;

  __shared__ __attribute__((aligned(16))) float S[SH * SW];
  __shared__ __attribute__((aligned(16))) float H[SEP ? SH * TW : 4];
  __shared__ double red[NT / 64];

  // XCD-aware tile id
  const int ntile = tiles_x * tiles_y;
  int bid = blockIdx.x;
  if ((ntile & 7) == 0) bid = (bid & 7) * (ntile >> 3) + (bid >> 3);
  const int ti = bid / tiles_x, tj = bid - ti * tiles_x;
  const int i0 = ti * TH, j0 = tj * TW;
  x += (int64_t)blockIdx.y * ldx;
  y += (int64_t)blockIdx.y * ldy;

  // ---- stage tile + halo
  const bool interior = (i0 - T >= 0) && (i0 + TH + KH / 2 <= nx) && (j0 - LP >= 0) && (j0 + TW + RP <= ny) &&
                        ((ny & 3) == 0) && ((reinterpret_cast<uintptr_t>(x) & 15u) == 0);
  if (interior) {
    const float* src = x + (int64_t)(i0 - T) * ny + (j0 - LP);
    for (int idx = threadIdx.x; idx < SH * SW4; idx += NT) {
      const int r = idx / SW4, c4 = idx - r * SW4;
      const float4 v = *reinterpret_cast<const float4*>(src + (int64_t)r * ny + 4 * c4);
      *reinterpret_cast<float4*>(&S[r * SW + 4 * c4]) = v;
    }
  } else {
    for (int idx = threadIdx.x; idx < SH * SW; idx += NT) {
      const int r = idx / SW, c = idx - r * SW;
      const int gi = reflect(i0 - T + r, nx), gj = reflect(j0 - LP + c, ny);
      S[idx] = x[(int64_t)gi * ny + gj];
    }
  }
  __syncthreads();

  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8 threads; thread block of 4 rows x 4 cols
  float acc[4][4];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[rr][c] = 0.f;

  if (SEP) {
    // row pass: H[r][c] = sum_b wr[b] * S[r][c + OFF + b]
    float wr[KW];
#pragma unroll
    for (int b = 0; b < KW; ++b) wr[b] = wts[b];
    for (int r = ty; r < SH; r += 8) {
      float v[NV * 4];
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        const float4 t = *reinterpret_cast<const float4*>(&S[r * SW + 4 * tx + 4 * q]);
        v[4 * q] = t.x;
        v[4 * q + 1] = t.y;
        v[4 * q + 2] = t.z;
        v[4 * q + 3] = t.w;
      }
      float h[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int b = 0; b < KW; ++b)
#pragma unroll
        for (int c = 0; c < 4; ++c) h[c] = fmaf(wr[b], v[OFF + c + b], h[c]);
      *reinterpret_cast<float4*>(&H[r * TW + 4 * tx]) = make_float4(h[0], h[1], h[2], h[3]);
    }
    __syncthreads();
    // column pass
    float wc[KH];
#pragma unroll
    for (int a = 0; a < KH; ++a) wc[a] = wts[KW + a];
#pragma unroll
    for (int a = 0; a < KH + 3; ++a) {
      const float4 hv = *reinterpret_cast<const float4*>(&H[(4 * ty + a) * TW + 4 * tx]);
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        if (a - rr >= 0 && a - rr < KH) {
          const float w = wc[a - rr];
          acc[rr][0] = fmaf(w, hv.x, acc[rr][0]);
          acc[rr][1] = fmaf(w, hv.y, acc[rr][1]);
          acc[rr][2] = fmaf(w, hv.z, acc[rr][2]);
          acc[rr][3] = fmaf(w, hv.w, acc[rr][3]);
        }
      }
    }
  } else {
#pragma unroll
    for (int a = 0; a < KH + 3; ++a) {
      float v[NV * 4];
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        const float4 t = *reinterpret_cast<const float4*>(&S[(4 * ty + a) * SW + 4 * tx + 4 * q]);
        v[4 * q] = t.x;
        v[4 * q + 1] = t.y;
        v[4 * q + 2] = t.z;
        v[4 * q + 3] = t.w;
      }
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        if (a - rr >= 0 && a - rr < KH) {
#pragma unroll
          for (int b = 0; b < KW; ++b) {
            const float w = wts[(a - rr) * KW + b];
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[rr][c] = fmaf(w, v[OFF + c + b], acc[rr][c]);
          }
        }
      }
    }
  }

  // ---- store (16-byte coalesced when the row is aligned and fully inside)
  double ss = 0.0;
  const int gj = j0 + 4 * tx;
  const bool vec_ok = ((ny & 3) == 0) && ((reinterpret_cast<uintptr_t>(y) & 15u) == 0) && (gj + 3 < ny);
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int gi = i0 + 4 * ty + rr;
    if (gi < nx) {
      float* dst = y + (int64_t)gi * ny + gj;
      if (vec_ok) {
        *reinterpret_cast<float4*>(dst) = make_float4(acc[rr][0], acc[rr][1], acc[rr][2], acc[rr][3]);
        if (SUMSQ)
          ss += (double)acc[rr][0] * acc[rr][0] + (double)acc[rr][1] * acc[rr][1] + (double)acc[rr][2] * acc[rr][2] +
                (double)acc[rr][3] * acc[rr][3];
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (gj + c < ny) {
            dst[c] = acc[rr][c];
            if (SUMSQ) ss += (double)acc[rr][c] * acc[rr][c];
          }
      }
    }
  }
  if (SUMSQ) {
    ss = block_sum<NT>(ss, red);
    if (threadIdx.x == 0) partials[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = ss;
  }
}

// ------------------------------------------------------------------------------------------------ generic kernel
template <bool SUMSQ>
__global__ __launch_bounds__(NT) void k_blur_generic(const float* __restrict__ x, int64_t ldx, float* __restrict__ y,
                                                     int64_t ldy, int nx, int ny, int kh, int kw,
                                                     const float* __restrict__ w, double* __restrict__ partials) {
  __shared__ double red[NT / 64];
  const int T = kh - 1 - kh / 2, L = kw - 1 - kw / 2;
  x += (int64_t)blockIdx.y * ldx;
  y += (int64_t)blockIdx.y * ldy;
  const int64_t npix = (int64_t)nx * ny;
  double ss = 0.0;
  for (int64_t idx = (int64_t)blockIdx.x * NT + threadIdx.x; idx < npix; idx += (int64_t)gridDim.x * NT) {
    const int i = (int)(idx / ny), j = (int)(idx - (int64_t)i * ny);
    float acc = 0.f;
    for (int a = 0; a < kh; ++a) {
      const int gi = reflect(i - T + a, nx);
      const float* row = x + (int64_t)gi * ny;
      for (int b = 0; b < kw; ++b) acc = fmaf(w[a * kw + b], row[reflect(j - L + b, ny)], acc);
    }
    y[idx] = acc;
    if (SUMSQ) ss += (double)acc * acc;
  }
  if (SUMSQ) {
    ss = block_sum<NT>(ss, red);
    if (threadIdx.x == 0) partials[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = ss;
  }
}

// ------------------------------------------------------------------------------------------------ dispatch
template <int K>
int launch_tile(const BlurImpl* im, int tr, const float* x, int64_t ldx, float* y, int64_t ldy, int batch,
                double* part, int tiles_x, int tiles_y, hipStream_t s) {
  dim3 grid(tiles_x * tiles_y, batch), block(NT);
  const float* w = im->separable ? im->sep_dev[tr] : im->w_dev[tr];
#define BL(SEP, SS) \
  hipLaunchKernelGGL((k_blur_tile<K, K, SEP, SS>), grid, block, 0, s, x, ldx, y, ldy, im->nx, im->ny, w, part, tiles_x, tiles_y)
  if (im->separable) { if (part) BL(true, true); else BL(true, false); }
  else               { if (part) BL(false, true); else BL(false, false); }
#undef BL
  TRK_LAUNCH_CHECK();
  return TRK_OK;
}

int blur_apply(trk_op* op, int tr, const float* x, int64_t ldx, float* y, int64_t ldy, int batch, double* sumsq,
               hipStream_t s) {
  auto* im = static_cast<BlurImpl*>(op->impl);
  double* part = nullptr;
  int nblk;
  TimerScope tm(op->timer, op->timer_which, tr, s);
  if (im->tiled) {
    const int tiles_x = ceil_div(im->ny, TW), tiles_y = ceil_div(im->nx, TH);
    nblk = tiles_x * tiles_y;
    if (sumsq)
      if (int rc = scratch_doubles(s, (size_t)nblk * batch, &part)) return rc;
    int rc;
    switch (im->kh) {
      case 3: rc = launch_tile<3>(im, tr, x, ldx, y, ldy, batch, part, tiles_x, tiles_y, s); break;
      case 5: rc = launch_tile<5>(im, tr, x, ldx, y, ldy, batch, part, tiles_x, tiles_y, s); break;
      case 7: rc = launch_tile<7>(im, tr, x, ldx, y, ldy, batch, part, tiles_x, tiles_y, s); break;
      case 9: rc = launch_tile<9>(im, tr, x, ldx, y, ldy, batch, part, tiles_x, tiles_y, s); break;
      case 11: rc = launch_tile<11>(im, tr, x, ldx, y, ldy, batch, part, tiles_x, tiles_y, s); break;
      case 13: rc = launch_tile<13>(im, tr, x, ldx, y, ldy, batch, part, tiles_x, tiles_y, s); break;
      case 15: rc = launch_tile<15>(im, tr, x, ldx, y, ldy, batch, part, tiles_x, tiles_y, s); break;
      default: return fail(TRK_EUNSUPPORTED, "blur2d: no tiled kernel for %dx%d", im->kh, im->kw);
    }
    if (rc) return rc;
  } else {
    const int64_t npix = (int64_t)im->nx * im->ny;
    int64_t want = (npix + NT - 1) / NT;
    if (want > 4096) want = 4096;
    nblk = (int)(want < 1 ? 1 : want);
    if (sumsq)
      if (int rc = scratch_doubles(s, (size_t)nblk * batch, &part)) return rc;
    dim3 grid(nblk, batch);
    if (part)
      hipLaunchKernelGGL((k_blur_generic<true>), grid, dim3(NT), 0, s, x, ldx, y, ldy, im->nx, im->ny, im->kh, im->kw, im->w_dev[tr], part);
    else
      hipLaunchKernelGGL((k_blur_generic<false>), grid, dim3(NT), 0, s, x, ldx, y, ldy, im->nx, im->ny, im->kh, im->kw, im->w_dev[tr], part);
    TRK_LAUNCH_CHECK();
  }
  tm.stop();
  if (sumsq) return finalize_sums(part, nblk * batch, 1, 1, sumsq, s);
  return TRK_OK;
}

void blur_destroy(trk_op* op) {
  auto* im = static_cast<BlurImpl*>(op->impl);
  for (int t = 0; t < 2; ++t) {
    if (im->w_dev[t]) (void)hipFree(im->w_dev[t]);
    if (im->sep_dev[t]) (void)hipFree(im->sep_dev[t]);
  }
  delete im;
}

}  // namespace

extern "C" int trk_blur2d_create(const double* psf, int kh, int kw, int nx, int ny, trk_op** out) {
  TRK_REQUIRE(psf && out, "trk_blur2d_create: NULL argument");
  TRK_REQUIRE(kh >= 1 && kw >= 1 && nx >= 1 && ny >= 1, "trk_blur2d_create: sizes must be >= 1");
  TRK_REQUIRE((int64_t)kh * kw <= (1 << 24), "trk_blur2d_create: PSF too large");
  auto* im = new BlurImpl{nx, ny, kh, kw, false, false, {nullptr, nullptr}, {nullptr, nullptr}};

  // correlation weights: forward c[a'][b'] = psf[kh-1-a'][kw-1-b'];  "transpose" convolves with flip(psf): c = psf
  std::vector<float> wf((size_t)kh * kw), wt((size_t)kh * kw);
  for (int a = 0; a < kh; ++a)
    for (int b = 0; b < kw; ++b) {
      wf[(size_t)a * kw + b] = (float)psf[(size_t)(kh - 1 - a) * kw + (kw - 1 - b)];
      wt[(size_t)a * kw + b] = (float)psf[(size_t)a * kw + b];
    }

  // rank-1 test in double: psf[a][b] == col[a] * row[b]  (pivot = largest |entry|)
  int pa = 0, pb = 0;
  double pmax = 0.0;
  for (int a = 0; a < kh; ++a)
    for (int b = 0; b < kw; ++b)
      if (std::fabs(psf[(size_t)a * kw + b]) > pmax) {
        pmax = std::fabs(psf[(size_t)a * kw + b]);
        pa = a;
        pb = b;
      }
  std::vector<double> col(kh), row(kw);
  bool sep = pmax > 0.0;
  if (sep) {
    const double piv = psf[(size_t)pa * kw + pb];
    for (int a = 0; a < kh; ++a) col[a] = psf[(size_t)a * kw + pb];
    for (int b = 0; b < kw; ++b) row[b] = psf[(size_t)pa * kw + b] / piv;
    for (int a = 0; a < kh && sep; ++a)
      for (int b = 0; b < kw; ++b)
        if (std::fabs(psf[(size_t)a * kw + b] - col[a] * row[b]) > 1e-12 * pmax) {
          sep = false;
          break;
        }
  }
  im->separable = sep;
  im->tiled = (kh == kw) && (kh & 1) && kh >= 3 && kh <= 15;

  auto upload = [](const std::vector<float>& h, float** d) -> int {
    TRK_HIP(hipMalloc(d, h.size() * sizeof(float)));
    TRK_HIP(hipMemcpy(*d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    return TRK_OK;
  };
  int rc = upload(wf, &im->w_dev[0]);
  if (!rc) rc = upload(wt, &im->w_dev[1]);
  if (!rc && sep) {
    std::vector<float> sf(kw + kh), st(kw + kh);
    for (int b = 0; b < kw; ++b) {
      sf[b] = (float)row[kw - 1 - b];
      st[b] = (float)row[b];
    }
    for (int a = 0; a < kh; ++a) {
      sf[kw + a] = (float)col[kh - 1 - a];
      st[kw + a] = (float)col[a];
    }
    rc = upload(sf, &im->sep_dev[0]);
    if (!rc) rc = upload(st, &im->sep_dev[1]);
  }
  if (rc) {
    trk_op tmp{1, 0, 0, im, nullptr, nullptr, nullptr, 0};
    blur_destroy(&tmp);
    return rc;
  }
  const int64_t n = (int64_t)nx * ny;
  *out = new trk_op{1, n, n, im, blur_apply, blur_destroy, nullptr, 0};
  return TRK_OK;
}
