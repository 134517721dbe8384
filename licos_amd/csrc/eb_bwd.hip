// Backward of the entropy models' likelihoods (the training step of /root/reference/licos/train.py:186-200
// differentiates RateDistortionLoss through EntropyBottleneck._likelihood and, for the scale hyperprior, through
// GaussianConditional._likelihood - [CAI] entropy_models/entropy_models.py, ops/bound_ops.py), plus the point-wise
// ReLU / |x| gradient masks of the hyper transforms.
//
// EntropyBottleneck: lik = max(sigmoid(F(v + .5)) - sigmoid(F(v - .5)), bound), F the per-channel MLP
//   h_0 = v -+ .5;   z_i = softplus(M_i) h_i + b_i;   h_(i+1) = z_i + tanh(f_i) tanh(z_i)   (no gate on the last layer).
// Given g = dL/dlik the kernel returns dL/dv and dL/d(M_i, b_i, f_i) per channel, w.r.t. the RAW parameters (the
// softplus' and tanh' factors are applied here).  LowerBound passes the gradient where lik >= bound or g < 0.
// One wave per (channel, slice of the channel's B x HW elements); per 64 elements:
//   (1) lane = element: forward of both branches, h_i and tanh(z_i) kept in LDS, one row per (branch, element) of odd
//       length (conflict-free for these lane-per-element accesses and for the reads of (3));
//   (2) lane = element: backward through the layers; dL/dz_i replaces tanh(z_i), dL/dh_(i+1) tanh(z_i) goes beside it;
//   (3) lane = PARAMETER: every lane owns a few entries of the packed record and adds (delta x h) over the 64 elements
//       from LDS into registers: no atomics, no cross-lane reduction, a fixed summation order.
// Slices write partial sums [slice][channel][per_channel]; the caller adds the slices.
#include "common.hpp"

namespace licos {

constexpr int EBB_MAX_LAYERS = 8, EBB_MAX_WIDTH = 16, EBB_MAX_OWN = 8;

struct EbbShape {
  int n_layers;
  int f[EBB_MAX_LAYERS + 1];     // (1, filters..., 1)
  int pbase[EBB_MAX_LAYERS];     // offset of layer i's matrix in the packed record (then bias, then tanh(factor))
  int hoff[EBB_MAX_LAYERS];      // offset of h_i (f[i] values) in a row
  int zoff[EBB_MAX_LAYERS];      // offset of z_i (f[i+1] values) in the row's second / third part
  int per_channel, units, row;   // units = sum f[0..n-1] = sum f[1..n];  row = 3 * units rounded up to odd
};

__device__ inline float ebb_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// VARIANT 1..3: the filter tuples LICOS instantiates (/root/reference/licos/model_utils.py:25-29: filters = (in, in, 3, 3)
// for 3 / 1 / 13 input channels) with compile-time extents - every per-element array stays in registers; 0 = any shape
// (runtime extents: the per-element arrays live in scratch).
template <int V> struct EbbCt { static constexpr int f[6] = {1, 1, 1, 1, 1, 1}; };
template <> struct EbbCt<1> { static constexpr int f[6] = {1, 3, 3, 3, 3, 1}; };
template <> struct EbbCt<2> { static constexpr int f[6] = {1, 1, 1, 3, 3, 1}; };
template <> struct EbbCt<3> { static constexpr int f[6] = {1, 13, 13, 3, 3, 1}; };
template <int V>
__device__ __forceinline__ int ebb_f(const EbbShape &s, int i) {
  if constexpr (V == 0) return s.f[i];
  else return EbbCt<V>::f[i];
}

template <int V>
__global__ __launch_bounds__(64) void eb_likelihood_bwd_kernel(const float *__restrict__ v, const float *__restrict__ g,
                                                               const float *__restrict__ packed, EbbShape s, float bound,
                                                               int form, float *__restrict__ dv, float *__restrict__ dparams,
                                                               int B, int C, int HW, int nslice) {
  extern __shared__ float smem[];
  const int c = blockIdx.x, slice = blockIdx.y, lane = threadIdx.x;
  const int U = s.units, UP = s.row;
  const int NL = V ? 5 : s.n_layers;
  float *s_p = smem;                                  // the channel's packed record
  float *s_act = smem + ((s.per_channel + 3) & ~3);   // [branch][element][UP]: h | dz (tanh z first) | dh * tanh z
  for (int e = lane; e < s.per_channel; e += 64) s_p[e] = packed[(size_t)c * s.per_channel + e];

  // (3)'s ownership: entry p = lane + 64 k of the record -> LDS offsets (A, Bq): acc += row[A] * (Bq >= 0 ? row[Bq] : 1)
  int own_a[EBB_MAX_OWN], own_b[EBB_MAX_OWN];
  float acc[EBB_MAX_OWN];
#pragma unroll
  for (int k = 0; k < EBB_MAX_OWN; ++k) {
    acc[k] = 0.f;
    own_a[k] = -1;
    own_b[k] = -1;
    const int p = lane + 64 * k;
    if (p < s.per_channel) {
      for (int i = 0; i < s.n_layers; ++i) {
        const int rows = s.f[i + 1], cols = s.f[i], rel = p - s.pbase[i];
        const int span = rows * cols + rows + (i < s.n_layers - 1 ? rows : 0);
        if (rel < 0 || rel >= span) continue;
        if (rel < rows * cols) {
          own_a[k] = U + s.zoff[i] + rel / cols;
          own_b[k] = s.hoff[i] + rel % cols;
        } else if (rel < rows * cols + rows) {
          own_a[k] = U + s.zoff[i] + (rel - rows * cols);
        } else {
          own_a[k] = 2 * U + s.zoff[i] + (rel - rows * cols - rows);
        }
      }
    }
  }
  __syncthreads();

  const long n_el = (long)B * HW;
  const long per_slice = (n_el + nslice - 1) / nslice;
  const long e_begin = slice * per_slice, e_end = (e_begin + per_slice < n_el) ? e_begin + per_slice : n_el;

  for (long e0 = e_begin; e0 < e_end; e0 += 64) {
    const long e = e0 + lane;
    const bool live = e < e_end;
    size_t gi = 0;
    float x = 0.f, gl = 0.f;
    if (live) {
      gi = ((size_t)(e / HW) * C + c) * HW + (size_t)(e % HW);
      x = v[gi];
      gl = g[gi];
    }
    // ---- (1) forward, both branches ---------------------------------------------------------------------
    float outv[2];
    for (int br = 0; br < 2; ++br) {
      float *a = s_act + ((size_t)br * 64 + lane) * UP;
      float cur[EBB_MAX_WIDTH], nxt[EBB_MAX_WIDTH];
      cur[0] = x + (br ? 0.5f : -0.5f);
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        const int rows = ebb_f<V>(s, i + 1), cols = ebb_f<V>(s, i), base = s.pbase[i];
#pragma unroll
        for (int q = 0; q < cols; ++q) a[s.hoff[i] + q] = cur[q];
#pragma unroll
        for (int r = 0; r < rows; ++r) {
          float t = 0.f;
#pragma unroll
          for (int q = 0; q < cols; ++q) t = fmaf(s_p[base + r * cols + q], cur[q], t);
          nxt[r] = t + s_p[base + rows * cols + r];
        }
        if (i < NL - 1) {
#pragma unroll
          for (int r = 0; r < rows; ++r) {
            const float th = tanhf(nxt[r]);
            a[U + s.zoff[i] + r] = th;
            nxt[r] = nxt[r] + s_p[base + rows * cols + rows + r] * th;
          }
        }
#pragma unroll
        for (int r = 0; r < rows; ++r) cur[r] = nxt[r];
      }
      outv[br] = cur[0];
    }
    // ---- likelihood, LowerBound rule, dL/d(out) of each branch -----------------------------------------------
    const float lo = outv[0], up = outv[1];
    float d_lo, d_up, raw;
    if (form == 0) {
      const float su = ebb_sigmoid(up), sl = ebb_sigmoid(lo);
      raw = su - sl;
      d_up = su * (1.f - su);
      d_lo = -sl * (1.f - sl);
    } else {
      const float t = lo + up;
      const float sg = (t > 0.f) ? -1.f : ((t < 0.f) ? 1.f : 0.f);
      const float su = ebb_sigmoid(sg * up), sl = ebb_sigmoid(sg * lo);
      const float a = su - sl, sa = (a > 0.f) ? 1.f : ((a < 0.f) ? -1.f : 0.f);
      raw = fabsf(a);
      d_up = sa * sg * su * (1.f - su);
      d_lo = -sa * sg * sl * (1.f - sl);
    }
    const float gp = (live && (raw >= bound || gl < 0.f)) ? gl : 0.f;
    // ---- (2) backward, both branches --------------------------------------------------------------------
    float dx = 0.f;
    for (int br = 0; br < 2; ++br) {
      float *a = s_act + ((size_t)br * 64 + lane) * UP;
      float dh[EBB_MAX_WIDTH], dprev[EBB_MAX_WIDTH];
      dh[0] = gp * (br ? d_up : d_lo);
#pragma unroll
      for (int ii = 0; ii < NL; ++ii) {
        const int i = NL - 1 - ii;
        const int rows = ebb_f<V>(s, i + 1), cols = ebb_f<V>(s, i), base = s.pbase[i];
        if (i < NL - 1) {
#pragma unroll
          for (int r = 0; r < rows; ++r) {
            const float th = a[U + s.zoff[i] + r], tf = s_p[base + rows * cols + rows + r];
            a[2 * U + s.zoff[i] + r] = dh[r] * th;                 // -> d tanh(f_i)
            dh[r] = dh[r] * (1.f + tf * (1.f - th * th));          // dL/dz_i
          }
        }
#pragma unroll
        for (int r = 0; r < rows; ++r) a[U + s.zoff[i] + r] = dh[r];
#pragma unroll
        for (int q = 0; q < cols; ++q) {
          float t = 0.f;
#pragma unroll
          for (int r = 0; r < rows; ++r) t = fmaf(s_p[base + r * cols + q], dh[r], t);
          dprev[q] = t;
        }
#pragma unroll
        for (int q = 0; q < cols; ++q) dh[q] = dprev[q];
      }
      dx += dh[0];
    }
    if (live) dv[gi] = dx;
    __syncthreads();
    // ---- (3) lane = parameter ----------------------------------------------------------------------------
    const int cnt = (int)((e_end - e0 < 64) ? (e_end - e0) : 64);
    for (int br = 0; br < 2; ++br)
      for (int el = 0; el < cnt; ++el) {
        const float *row = s_act + ((size_t)br * 64 + el) * UP;
#pragma unroll
        for (int k = 0; k < EBB_MAX_OWN; ++k)
          if (own_a[k] >= 0) acc[k] = fmaf(row[own_a[k]], own_b[k] >= 0 ? row[own_b[k]] : 1.f, acc[k]);
      }
    __syncthreads();
  }
  // chain factors of the raw parameters: d softplus(m)/dm = sigmoid(m) = 1 - exp(-softplus(m)); d tanh(f)/df = 1 - tanh(f)^2
  float *out = dparams + ((size_t)slice * C + c) * s.per_channel;
#pragma unroll
  for (int k = 0; k < EBB_MAX_OWN; ++k) {
    const int p = lane + 64 * k;
    if (p >= s.per_channel) continue;
    float val = acc[k];
    if (own_b[k] >= 0) val *= 1.f - expf(-s_p[p]);
    else if (own_a[k] >= 2 * U) val *= 1.f - s_p[p] * s_p[p];
    out[p] = val;
  }
}

// ---- Gaussian conditional -------------------------------------------------------------------------------------
// lik = max(Phi((.5 - |v|) / s) - Phi((-.5 - |v|) / s), lik_bound), s = max(scale, scale_bound); both bounds with
// LowerBound's gradient rule.  Element-wise: dL/dv and dL/dscale.
__global__ void gc_likelihood_bwd_kernel(const float *__restrict__ v, const float *__restrict__ scales, const float *__restrict__ g,
                                         float scale_bound, float lik_bound, float *__restrict__ dv, float *__restrict__ dscale,
                                         long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float sc = scales[i], s = fmaxf(sc, scale_bound), x = v[i], a = fabsf(x), gl = g[i];
    const float u1 = (0.5f - a) / s, u2 = (-0.5f - a) / s;
    const float raw = 0.5f * erfcf(-0.70710678118654752440f * u1) - 0.5f * erfcf(-0.70710678118654752440f * u2);
    const float gp = (raw >= lik_bound || gl < 0.f) ? gl : 0.f;
    const float p1 = 0.3989422804014327f * expf(-0.5f * u1 * u1), p2 = 0.3989422804014327f * expf(-0.5f * u2 * u2);
    const float sgn = (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f);
    dv[i] = gp * (p2 - p1) / s * sgn;
    const float ds = gp * (p2 * u2 - p1 * u1) / s;   // dL/d(bounded scale)
    dscale[i] = (sc >= scale_bound || ds < 0.f) ? ds : 0.f;
  }
}

// out = g * (ref > 0) (mode 0: ReLU, ref = the ReLU's input or output) or g * sign(ref) (mode 1: |x|)
__global__ void mask_mul_kernel(const float *__restrict__ g, const float *__restrict__ ref, float *__restrict__ out, long n,
                                int mode) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float r = ref[i];
    const float m = mode == 0 ? (r > 0.f ? 1.f : 0.f) : (r > 0.f ? 1.f : (r < 0.f ? -1.f : 0.f));
    out[i] = g[i] * m;
  }
}

static int ebb_make_shape(const int *filters, int nfilt, EbbShape *s) {
  if (!filters || nfilt < 1 || nfilt + 1 > EBB_MAX_LAYERS) return -1;
  s->n_layers = nfilt + 1;
  s->f[0] = 1;
  for (int i = 0; i < nfilt; ++i) {
    if (filters[i] < 1 || filters[i] > EBB_MAX_WIDTH) return -1;
    s->f[i + 1] = filters[i];
  }
  s->f[nfilt + 1] = 1;
  int n = 0, h = 0, z = 0;
  for (int i = 0; i < s->n_layers; ++i) {
    s->pbase[i] = n;
    s->hoff[i] = h;
    s->zoff[i] = z;
    n += s->f[i + 1] * s->f[i] + s->f[i + 1];
    if (i < s->n_layers - 1) n += s->f[i + 1];
    h += s->f[i];
    z += s->f[i + 1];
  }
  s->per_channel = n;
  s->units = h;
  s->row = (3 * h) | 1;
  return (n <= 64 * EBB_MAX_OWN) ? 0 : -1;
}

}  // namespace licos

using namespace licos;

extern "C" {

int licos_eb_likelihood_bwd_slices(int B, int HW) {
  const long n = (long)B * HW;
  long s = n / 256;  // >= 4 wave-iterations per slice; 192 channels x 16 slices fill the chip's 2048 single-wave slots
  return (int)(s < 1 ? 1 : (s > 64 ? 64 : s));
}

int licos_eb_likelihood_bwd(const float *v, const float *g_lik, const float *packed, const int *filters, int nfilt, float bound,
                            int form, float *dv, float *dparams_slices, int B, int C, int HW, void *stream) {
  LICOS_REQUIRE(v && g_lik && packed && dv && dparams_slices && B > 0 && C > 0 && HW > 0, "eb_likelihood_bwd: bad arguments");
  EbbShape s;
  LICOS_REQUIRE(ebb_make_shape(filters, nfilt, &s) == 0, "eb_likelihood_bwd: unsupported filters (max 7 layers of width 16, 512 parameters per channel)");
  const int nslice = licos_eb_likelihood_bwd_slices(B, HW);
  const size_t lds = ((size_t)((s.per_channel + 3) & ~3) + (size_t)2 * 64 * s.row) * sizeof(float);
  LICOS_REQUIRE(lds <= 64 * 1024, "eb_likelihood_bwd: filters too wide for the LDS working set");
  int variant = 0;
  if (nfilt == 4 && filters[2] == 3 && filters[3] == 3) {
    if (filters[0] == 3 && filters[1] == 3) variant = 1;
    else if (filters[0] == 1 && filters[1] == 1) variant = 2;
    else if (filters[0] == 13 && filters[1] == 13) variant = 3;
  }
#define LICOS_EBB_CASE(V)                                                                                                     \
  case V:                                                                                                                     \
    hipLaunchKernelGGL((eb_likelihood_bwd_kernel<V>), dim3(C, nslice), dim3(64), lds, as_stream(stream), v, g_lik, packed, s, \
                       bound, form, dv, dparams_slices, B, C, HW, nslice);                                                    \
    break;
  switch (variant) {
    LICOS_EBB_CASE(0)
    LICOS_EBB_CASE(1)
    LICOS_EBB_CASE(2)
    LICOS_EBB_CASE(3)
  }
#undef LICOS_EBB_CASE
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

int licos_gc_likelihood_bwd(const float *v, const float *scales, const float *g_lik, float scale_bound, float lik_bound, float *dv,
                            float *dscale, long n, void *stream) {
  LICOS_REQUIRE(v && scales && g_lik && dv && dscale && n > 0, "gc_likelihood_bwd: bad arguments");
  const int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
  hipLaunchKernelGGL(gc_likelihood_bwd_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), v, scales, g_lik, scale_bound,
                     lik_bound, dv, dscale, n);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

int licos_mask_mul_f32(const float *g, const float *ref, float *out, long n, int mode, void *stream) {
  LICOS_REQUIRE(g && ref && out && n > 0 && (mode == 0 || mode == 1), "mask_mul_f32: bad arguments");
  const int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
  hipLaunchKernelGGL(mask_mul_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), g, ref, out, n, mode);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

}  // extern "C"
