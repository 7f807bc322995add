/*
 * aa_oracle.c — CPU ORACLE (test infrastructure only; see aa_oracle.h for the rules and the pinning).
 *
 * Plain-C restatement of the reference's separable PIL-style antialiased resample.  "s2.2" below is
 * step_two_dot_two/aa_interpolation_impl.h in the reference checkout; the step_three separable variant
 * (step_three/aa_separable_single_dim_loop2d_impl.h:38-76,303-377) has the same arithmetic (SURVEY §8a K4:
 * outputs bit-identical), so one restatement serves both.
 *
 * Build: gcc -O2 -ffp-contract=off (oracle/Makefile) — products and sums must round separately, exactly as
 * the reference's non-FMA x86 build does, so fp32 results are bit-comparable.
 */
#include "aa_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int aao_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ------------------------------------------------------------------------------------------------
 * Filters.  Each takes scalar_t and evaluates with the reference's literal (double) constants, i.e. in
 * double, then narrows to scalar_t on return.
 * ---------------------------------------------------------------------------------------------- */

/* s2.2:292-300  HelperInterpLinear::_filter (triangle) */
static float filt_linear_f32(float x) {
  if (x < 0.0) x = -x;
  if (x < 1.0) return (float)(1.0 - x);
  return 0.0f;
}
static double filt_linear_f64(double x) {
  if (x < 0.0) x = -x;
  if (x < 1.0) return 1.0 - x;
  return 0.0;
}
/* s2.2:410-424  HelperInterpCubic::_filter (Keys, a=-0.5) */
static float filt_cubic_f32(float x) {
  const double a = -0.5;
  if (x < 0.0) x = -x;
  if (x < 1.0) return (float)(((a + 2.0) * x - (a + 3.0)) * x * x + 1);
  if (x < 2.0) return (float)((((x - 5) * x + 8) * x - 4) * a);
  return 0.0f;
}
static double filt_cubic_f64(double x) {
  const double a = -0.5;
  if (x < 0.0) x = -x;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}
/* s2.2:367-372  HelperInterpNearest::_filter ("it's not nearest but box", extension_interpolate.cpp:48) */
static float filt_box_f32(float x) { return (x > -0.5 && x <= 0.5) ? 1.0f : 0.0f; }
static double filt_box_f64(double x) { return (x > -0.5 && x <= 0.5) ? 1.0 : 0.0; }

static int filter_interp_size(int filter) {
  switch (filter) {
    case AAO_FILTER_LINEAR: return 2; /* s2.2:287 */
    case AAO_FILTER_CUBIC: return 4;  /* s2.2:377 */
    case AAO_FILTER_BOX: return 1;    /* s2.2:333 */
    default: return -1;
  }
}

/* ATen UpSample.h area_pixel_compute_scale<scalar_t> / compute_scales_value<scalar_t> (third-party; call site
 * s2.2:314-315).  scale_opt<=0 means "no user scale". */
static float scale_f32(int64_t in_size, int64_t out_size, int align_corners, double scale_opt) {
  if (align_corners) {
    if (out_size > 1) return (float)(in_size - 1) / (float)(out_size - 1);
    return 0.0f;
  }
  if (scale_opt > 0.) return (float)(1.0 / scale_opt);
  return (float)in_size / (float)out_size;
}
static double scale_f64(int64_t in_size, int64_t out_size, int align_corners, double scale_opt) {
  if (align_corners) {
    if (out_size > 1) return (double)(in_size - 1) / (double)(out_size - 1);
    return 0.0;
  }
  if (scale_opt > 0.) return 1.0 / scale_opt;
  return (double)in_size / (double)out_size;
}

/* s2.2:207-210 */
static int ksize_from_scale_f32(int filter, float scale) {
  int interp_size = filter_interp_size(filter);
  float support = (scale >= 1.0) ? (float)((interp_size * 0.5) * scale) : (float)(interp_size * 0.5);
  return (int)ceilf(support) * 2 + 1;
}
static int ksize_from_scale_f64(int filter, double scale) {
  int interp_size = filter_interp_size(filter);
  double support = (scale >= 1.0) ? (interp_size * 0.5) * scale : interp_size * 0.5;
  /* the reference calls ceilf() even for double (s2.2:210): the argument narrows to float first */
  return (int)ceilf((float)support) * 2 + 1;
}

int aao_ksize(int filter, int64_t in_size, int64_t out_size, int align_corners, double scale_opt, int is_double) {
  if (filter_interp_size(filter) < 0) return -1;
  if (is_double) return ksize_from_scale_f64(filter, scale_f64(in_size, out_size, align_corners, scale_opt));
  return ksize_from_scale_f32(filter, scale_f32(in_size, out_size, align_corners, scale_opt));
}

/* s2.2:195-281 _compute_indices_weights_aa, scalar_t=float.  The C usual-arithmetic-conversion of every
 * sub-expression is spelled out because one ulp moves a window (SURVEY §7 "Weight parity"). */
int aao_weights_f32(int filter, int64_t in_size, int64_t out_size, int align_corners, double scale_opt,
                    int64_t *xmin_out, int64_t *xsize_out, float *w) {
  int interp_size = filter_interp_size(filter);
  if (interp_size < 0) return -1;
  float (*filter_fn)(float) =
      filter == AAO_FILTER_LINEAR ? filt_linear_f32 : (filter == AAO_FILTER_CUBIC ? filt_cubic_f32 : filt_box_f32);
  const float scale = scale_f32(in_size, out_size, align_corners, scale_opt);
  /* :208-209  (interp_size*0.5) is double; *scale in double; narrowed to scalar_t */
  const float support = (scale >= 1.0) ? (float)((interp_size * 0.5) * scale) : (float)(interp_size * 0.5);
  const int ksize = (int)ceilf(support) * 2 + 1; /* :210 */
  /* :242  1.0/scale in double, narrowed */
  const float invscale = (scale >= 1.0) ? (float)(1.0 / scale) : 1.0f;

  for (int64_t i = 0; i < out_size; i++) {
    const float center = (float)(scale * (i + 0.5)); /* :253 double product, narrowed */
    /* :254  (center - support) is a float subtraction; + 0.5 in double; truncating cast */
    int64_t xmin = (int64_t)((float)(center - support) + 0.5);
    if (xmin < 0) xmin = 0;
    /* :255-257 */
    int64_t xmax = (int64_t)((float)(center + support) + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    xmin_out[i] = xmin;
    xsize_out[i] = xmax;

    float total_w = 0.0f;
    int64_t j = 0;
    for (; j < xmax; j++) {
      /* :266  (j + xmin) is int64; int64 - float -> float; + 0.5 -> double; * invscale -> double; the
       * filter's parameter is scalar_t so the argument narrows to float */
      float arg = (float)(((float)((float)(j + xmin) - center) + 0.5) * invscale);
      float wj = filter_fn(arg);
      w[i * ksize + j] = wj;
      total_w += wj;
    }
    for (j = 0; j < xmax; j++) {
      if (total_w != 0.0) w[i * ksize + j] /= total_w; /* :270-274 */
    }
    for (j = (xmax > 0 ? xmax : 0); j < ksize; j++) w[i * ksize + j] = 0.0f; /* :276-278 */
  }
  return ksize;
}

int aao_weights_f64(int filter, int64_t in_size, int64_t out_size, int align_corners, double scale_opt,
                    int64_t *xmin_out, int64_t *xsize_out, double *w) {
  int interp_size = filter_interp_size(filter);
  if (interp_size < 0) return -1;
  double (*filter_fn)(double) =
      filter == AAO_FILTER_LINEAR ? filt_linear_f64 : (filter == AAO_FILTER_CUBIC ? filt_cubic_f64 : filt_box_f64);
  const double scale = scale_f64(in_size, out_size, align_corners, scale_opt);
  const double support = (scale >= 1.0) ? (interp_size * 0.5) * scale : interp_size * 0.5;
  const int ksize = (int)ceilf((float)support) * 2 + 1;
  const double invscale = (scale >= 1.0) ? 1.0 / scale : 1.0;

  for (int64_t i = 0; i < out_size; i++) {
    const double center = scale * (i + 0.5);
    int64_t xmin = (int64_t)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int64_t xmax = (int64_t)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    xmin_out[i] = xmin;
    xsize_out[i] = xmax;
    double total_w = 0.0;
    int64_t j = 0;
    for (; j < xmax; j++) {
      double wj = filter_fn((j + xmin - center + 0.5) * invscale);
      w[i * ksize + j] = wj;
      total_w += wj;
    }
    for (j = 0; j < xmax; j++) {
      if (total_w != 0.0) w[i * ksize + j] /= total_w;
    }
    for (j = (xmax > 0 ? xmax : 0); j < ksize; j++) w[i * ksize + j] = 0.0;
  }
  return ksize;
}

/* ------------------------------------------------------------------------------------------------
 * Forward passes.  One macro body instantiated for float and double.
 *
 * Inner tap loop = s2.2:29-58 (vertical, "zero strides") and :60-87 (horizontal): tap 0 is taken
 * unconditionally, taps 1..size-1 accumulate in order with separately rounded product and sum.
 * Driver = s2.2:628-683: W pass into a contiguous temp [N,C,H,oW] (:655-668), then H pass (:677-679).
 * ---------------------------------------------------------------------------------------------- */
#define DEFINE_FORWARD(NAME, T, WEIGHTS_FN)                                                                   \
  int NAME(int filter, const T *in, T *out, int64_t N, int64_t C, int64_t H, int64_t W, int64_t oH,           \
           int64_t oW, const int64_t is[4], const int64_t os[4], int align_corners, int nthreads) {           \
    if (N < 0 || C <= 0 || H <= 0 || W <= 0 || oH <= 0 || oW <= 0) return -2;                                 \
    if (N == 0) return 0;                                                                                     \
    int kw_max = aao_ksize(filter, W, oW, align_corners, 0., sizeof(T) == 8);                                 \
    int kh_max = aao_ksize(filter, H, oH, align_corners, 0., sizeof(T) == 8);                                 \
    if (kw_max < 0 || kh_max < 0) return -1;                                                                  \
    int64_t *xmin = (int64_t *)malloc(sizeof(int64_t) * (size_t)(2 * oW + 2 * oH));                           \
    T *ww = (T *)malloc(sizeof(T) * (size_t)(oW * kw_max + oH * kh_max));                                     \
    T *tmp = (T *)malloc(sizeof(T) * (size_t)(N * C * H * oW));                                               \
    if (!xmin || !ww || !tmp) { free(xmin); free(ww); free(tmp); return -3; }                                 \
    int64_t *xsize = xmin + oW, *ymin = xsize + oW, *ysize = ymin + oH;                                       \
    T *wh = ww + oW * kw_max;                                                                                 \
    const int kw = WEIGHTS_FN(filter, W, oW, align_corners, 0., xmin, xsize, ww);                             \
    const int kh = WEIGHTS_FN(filter, H, oH, align_corners, 0., ymin, ysize, wh);                             \
    const int64_t rows = N * C * H;                                                                           \
    (void)nthreads;                                                                                           \
    /* W pass: s2.2:60-87 per output element */                                                               \
    _Pragma("omp parallel for schedule(static) num_threads(nthreads > 1 ? nthreads : 1)")                     \
    for (int64_t r = 0; r < rows; r++) {                                                                      \
      const int64_t n = r / (C * H), c = (r / H) % C, y = r % H;                                              \
      const T *src = in + n * is[0] + c * is[1] + y * is[2];                                                  \
      T *dst = tmp + r * oW;                                                                                  \
      for (int64_t ox = 0; ox < oW; ox++) {                                                                   \
        const T *wp = ww + ox * kw;                                                                           \
        const T *sp = src + xmin[ox] * is[3];                                                                 \
        T acc = sp[0] * wp[0];                                                                                \
        for (int64_t j = 1; j < xsize[ox]; j++) acc += sp[j * is[3]] * wp[j];                                 \
        dst[ox] = acc;                                                                                        \
      }                                                                                                       \
    }                                                                                                         \
    /* H pass: s2.2:29-58 */                                                                                  \
    const int64_t orows = N * C * oH;                                                                         \
    _Pragma("omp parallel for schedule(static) num_threads(nthreads > 1 ? nthreads : 1)")                     \
    for (int64_t r = 0; r < orows; r++) {                                                                     \
      const int64_t n = r / (C * oH), c = (r / oH) % C, oy = r % oH;                                          \
      const T *src = tmp + ((n * C + c) * H + ymin[oy]) * oW;                                                 \
      const T *wp = wh + oy * kh;                                                                             \
      T *dst = out + n * os[0] + c * os[1] + oy * os[2];                                                      \
      for (int64_t ox = 0; ox < oW; ox++) {                                                                   \
        T acc = src[ox] * wp[0];                                                                              \
        for (int64_t j = 1; j < ysize[oy]; j++) acc += src[j * oW + ox] * wp[j];                              \
        dst[ox * os[3]] = acc;                                                                                \
      }                                                                                                       \
    }                                                                                                         \
    free(xmin); free(ww); free(tmp);                                                                          \
    return 0;                                                                                                 \
  }

DEFINE_FORWARD(aao_forward_f32, float, aao_weights_f32)
DEFINE_FORWARD(aao_forward_f64, double, aao_weights_f64)

/* ------------------------------------------------------------------------------------------------
 * TRUE adjoint.  forward: out = V(H(x));  adjoint: gi = H^T(V^T(go)), same tables.
 * (The reference header's backward is the non-AA one — see aao_legacy_nonaa_linear_backward_f32.)
 * ---------------------------------------------------------------------------------------------- */
#define DEFINE_BACKWARD(NAME, T, WEIGHTS_FN)                                                                  \
  int NAME(int filter, const T *go, T *gi, int64_t N, int64_t C, int64_t H, int64_t W, int64_t oH,            \
           int64_t oW, int align_corners) {                                                                   \
    if (N < 0 || C <= 0 || H <= 0 || W <= 0 || oH <= 0 || oW <= 0) return -2;                                 \
    if (N == 0) return 0;                                                                                     \
    int kw_max = aao_ksize(filter, W, oW, align_corners, 0., sizeof(T) == 8);                                 \
    int kh_max = aao_ksize(filter, H, oH, align_corners, 0., sizeof(T) == 8);                                 \
    if (kw_max < 0 || kh_max < 0) return -1;                                                                  \
    int64_t *xmin = (int64_t *)malloc(sizeof(int64_t) * (size_t)(2 * oW + 2 * oH));                           \
    T *ww = (T *)malloc(sizeof(T) * (size_t)(oW * kw_max + oH * kh_max));                                     \
    T *tmp = (T *)calloc((size_t)(N * C * H * oW), sizeof(T));                                                \
    if (!xmin || !ww || !tmp) { free(xmin); free(ww); free(tmp); return -3; }                                 \
    int64_t *xsize = xmin + oW, *ymin = xsize + oW, *ysize = ymin + oH;                                       \
    T *wh = ww + oW * kw_max;                                                                                 \
    const int kw = WEIGHTS_FN(filter, W, oW, align_corners, 0., xmin, xsize, ww);                             \
    const int kh = WEIGHTS_FN(filter, H, oH, align_corners, 0., ymin, ysize, wh);                             \
    memset(gi, 0, sizeof(T) * (size_t)(N * C * H * W));                                                       \
    for (int64_t p = 0; p < N * C; p++) {                                                                     \
      /* V^T : [oH,oW] -> tmp [H,oW] */                                                                       \
      for (int64_t oy = 0; oy < oH; oy++) {                                                                   \
        int64_t taps = ysize[oy] > 1 ? ysize[oy] : 1; /* tap 0 is unconditional in the forward */             \
        for (int64_t j = 0; j < taps; j++) {                                                                  \
          const T wj = wh[oy * kh + j];                                                                       \
          T *trow = tmp + (p * H + ymin[oy] + j) * oW;                                                        \
          const T *grow = go + (p * oH + oy) * oW;                                                            \
          for (int64_t ox = 0; ox < oW; ox++) trow[ox] += wj * grow[ox];                                      \
        }                                                                                                     \
      }                                                                                                       \
      /* H^T : tmp [H,oW] -> gi [H,W] */                                                                      \
      for (int64_t y = 0; y < H; y++) {                                                                       \
        const T *trow = tmp + (p * H + y) * oW;                                                               \
        T *girow = gi + (p * H + y) * W;                                                                      \
        for (int64_t ox = 0; ox < oW; ox++) {                                                                 \
          int64_t taps = xsize[ox] > 1 ? xsize[ox] : 1;                                                       \
          for (int64_t j = 0; j < taps; j++) girow[xmin[ox] + j] += ww[ox * kw + j] * trow[ox];               \
        }                                                                                                     \
      }                                                                                                       \
    }                                                                                                         \
    free(xmin); free(ww); free(tmp);                                                                          \
    return 0;                                                                                                 \
  }

DEFINE_BACKWARD(aao_backward_f32, float, aao_weights_f32)
DEFINE_BACKWARD(aao_backward_f64, double, aao_weights_f64)

/* ------------------------------------------------------------------------------------------------
 * Legacy backward exactly as the reference header writes it: NON-AA 2x2-tap scatter
 * (s2.2/aa_interpolation_backward_impl.h:80-108).  Index/lambda = ATen UpSample.h
 * compute_source_index_and_lambda / area_pixel_compute_source_index / guard_index_and_lambda (third-party).
 * ---------------------------------------------------------------------------------------------- */
static void src_index_lambda_f32(int64_t *i0, int64_t *i1, float *l0, float *l1, float ratio, int64_t oi,
                                 int64_t in_size, int64_t out_size, int align_corners) {
  if (out_size == in_size) {
    *i0 = oi; *i1 = oi; *l0 = 1.f; *l1 = 0.f;
    return;
  }
  float real;
  if (align_corners) {
    real = ratio * (float)oi;
  } else {
    real = ratio * ((float)oi + 0.5f) - 0.5f;
    if (real < 0.f) real = 0.f;
  }
  int64_t idx = (int64_t)floorf(real);
  if (idx > in_size - 1) idx = in_size - 1;
  float lam = real - (float)idx;
  if (lam < 0.f) lam = 0.f;
  if (lam > 1.f) lam = 1.f;
  *i0 = idx;
  *i1 = idx + ((idx < in_size - 1) ? 1 : 0);
  *l1 = lam;
  *l0 = 1.f - lam;
}

int aao_legacy_nonaa_linear_backward_f32(const float *go, float *gi, int64_t N, int64_t C, int64_t H, int64_t W,
                                         int64_t oH, int64_t oW, int align_corners) {
  if (N < 0 || C <= 0 || H <= 0 || W <= 0 || oH <= 0 || oW <= 0) return -2;
  const float hs = scale_f32(H, oH, align_corners, 0.), ws = scale_f32(W, oW, align_corners, 0.);
  memset(gi, 0, sizeof(float) * (size_t)(N * C * H * W));
  for (int64_t c = 0; c < N * C; c++) {
    for (int64_t oh = 0; oh < oH; oh++) {
      int64_t ih0, ih1, iw0, iw1;
      float h0, h1, w0, w1;
      src_index_lambda_f32(&ih0, &ih1, &h0, &h1, hs, oh, H, oH, align_corners);
      for (int64_t ow = 0; ow < oW; ow++) {
        src_index_lambda_f32(&iw0, &iw1, &w0, &w1, ws, ow, W, oW, align_corners);
        const float g = go[(c * oH + oh) * oW + ow];
        float *base = gi + c * H * W;
        base[ih0 * W + iw0] += h0 * w0 * g;
        base[ih0 * W + iw1] += h0 * w1 * g;
        base[ih1 * W + iw0] += h1 * w0 * g;
        base[ih1 * W + iw1] += h1 * w1 * g;
      }
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------------
 * uint8, Pillow semantics.  Algorithm: Pillow src/libImaging/Resample.c (cited by URL in the reference:
 * README.md:18,40; s2.2/aa_interpolation_impl.h:289-291,364-366,407-409) — precompute_coeffs,
 * normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc / Vertical_8bpc.  Pillow is a third-party
 * dependency absent from /root/reference (pinned here as Pillow 12.2.0, the version in this image); this
 * restatement is pinned bit-for-bit by Pillow outputs in tests/golden/pil_*.npz.
 * ---------------------------------------------------------------------------------------------- */
#define PIL_PRECISION_BITS (32 - 8 - 2)

static double pil_filter(int filter, double x) {
  switch (filter) {
    case AAO_FILTER_LINEAR: return filt_linear_f64(x);
    case AAO_FILTER_CUBIC: return filt_cubic_f64(x);
    default: /* Pillow's box_filter since 7.x: half-open the other way round from the reference's */
      return (x > -0.5 && x <= 0.5) ? 1.0 : 0.0;
  }
}
static double pil_support(int filter) {
  switch (filter) {
    case AAO_FILTER_LINEAR: return 1.0;
    case AAO_FILTER_CUBIC: return 2.0;
    default: return 0.5;
  }
}

int aao_pil_ksize(int filter, int64_t in_size, int64_t out_size) {
  double filterscale = (double)in_size / (double)out_size;
  if (filterscale < 1.0) filterscale = 1.0;
  double support = pil_support(filter) * filterscale;
  return (int)ceil(support) * 2 + 1;
}

int aao_pil_coeffs(int filter, int64_t in_size, int64_t out_size, int32_t *xmin_out, int32_t *xsize_out, int32_t *kk,
                   double *prekk_out) {
  double scale, filterscale;
  filterscale = scale = (double)in_size / (double)out_size;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = pil_support(filter) * filterscale;
  const int ksize = (int)ceil(support) * 2 + 1;
  double *k = (double *)malloc(sizeof(double) * (size_t)ksize);
  if (!k) return -3;
  for (int64_t xx = 0; xx < out_size; xx++) {
    const double center = 0.0 + (xx + 0.5) * scale;
    double ww = 0.0;
    const double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = (int)in_size;
    xmax -= xmin;
    int x;
    for (x = 0; x < xmax; x++) {
      double w = pil_filter(filter, (x + xmin - center + 0.5) * ss);
      k[x] = w;
      ww += w;
    }
    for (x = 0; x < xmax; x++) {
      if (ww != 0.0) k[x] /= ww;
    }
    for (x = xmax > 0 ? xmax : 0; x < ksize; x++) k[x] = 0;
    xmin_out[xx] = xmin;
    xsize_out[xx] = xmax;
    for (x = 0; x < ksize; x++) {
      if (prekk_out) prekk_out[xx * ksize + x] = k[x];
      /* normalize_coeffs_8bpc */
      if (k[x] < 0) kk[xx * ksize + x] = (int32_t)(-0.5 + k[x] * (1 << PIL_PRECISION_BITS));
      else kk[xx * ksize + x] = (int32_t)(0.5 + k[x] * (1 << PIL_PRECISION_BITS));
    }
  }
  free(k);
  return ksize;
}

static inline uint8_t pil_clip8(int32_t v) {
  v >>= PIL_PRECISION_BITS; /* arithmetic shift, as Pillow's clip8 lookup index */
  if (v < 0) return 0;
  if (v > 255) return 255;
  return (uint8_t)v;
}

int aao_pil_resize_u8(int filter, const uint8_t *in, uint8_t *out, int64_t N, int64_t C, int64_t H, int64_t W,
                      int64_t oH, int64_t oW, const int64_t is[4], const int64_t os[4], int nthreads) {
  if (N < 0 || C <= 0 || H <= 0 || W <= 0 || oH <= 0 || oW <= 0) return -2;
  if (N == 0) return 0;
  const int kw_max = aao_pil_ksize(filter, W, oW), kh_max = aao_pil_ksize(filter, H, oH);
  int32_t *bounds = (int32_t *)malloc(sizeof(int32_t) * (size_t)(2 * oW + 2 * oH + oW * kw_max + oH * kh_max));
  uint8_t *tmp = (uint8_t *)malloc((size_t)(N * C * H * oW));
  if (!bounds || !tmp) { free(bounds); free(tmp); return -3; }
  int32_t *xmin = bounds, *xsize = xmin + oW, *ymin = xsize + oW, *ysize = ymin + oH;
  int32_t *kw_tab = ysize + oH, *kh_tab = kw_tab + oW * kw_max;
  const int kw = aao_pil_coeffs(filter, W, oW, xmin, xsize, kw_tab, NULL);
  const int kh = aao_pil_coeffs(filter, H, oH, ymin, ysize, kh_tab, NULL);
  /* Pillow skips a pass whose size is unchanged (ImagingResampleInner need_horizontal/need_vertical); with
   * these filters the unchanged-size pass is an exact identity, so running it is equivalent. */
  (void)nthreads;
  const int64_t rows = N * C * H;
  _Pragma("omp parallel for schedule(static) num_threads(nthreads > 1 ? nthreads : 1)")
  for (int64_t r = 0; r < rows; r++) {
    const int64_t n = r / (C * H), c = (r / H) % C, y = r % H;
    const uint8_t *src = in + n * is[0] + c * is[1] + y * is[2];
    uint8_t *dst = tmp + r * oW;
    for (int64_t ox = 0; ox < oW; ox++) {
      int32_t ss = 1 << (PIL_PRECISION_BITS - 1);
      const int32_t *k = kw_tab + ox * kw;
      for (int x = 0; x < xsize[ox]; x++) ss += (int32_t)src[(x + xmin[ox]) * is[3]] * k[x];
      dst[ox] = pil_clip8(ss);
    }
  }
  const int64_t orows = N * C * oH;
  _Pragma("omp parallel for schedule(static) num_threads(nthreads > 1 ? nthreads : 1)")
  for (int64_t r = 0; r < orows; r++) {
    const int64_t n = r / (C * oH), c = (r / oH) % C, oy = r % oH;
    const uint8_t *src = tmp + ((n * C + c) * H + ymin[oy]) * oW;
    const int32_t *k = kh_tab + oy * kh;
    uint8_t *dst = out + n * os[0] + c * os[1] + oy * os[2];
    for (int64_t ox = 0; ox < oW; ox++) {
      int32_t ss = 1 << (PIL_PRECISION_BITS - 1);
      for (int y = 0; y < ysize[oy]; y++) ss += (int32_t)src[y * oW + ox] * k[y];
      dst[ox * os[3]] = pil_clip8(ss);
    }
  }
  free(bounds); free(tmp);
  return 0;
}

int aao_pil_resize_u8_nhwc(int filter, const uint8_t *in, uint8_t *out, int64_t N, int64_t H, int64_t W, int64_t C,
                           int64_t oH, int64_t oW, int nthreads) {
  const int64_t is[4] = {H * W * C, 1, W * C, C}, os[4] = {oH * oW * C, 1, oW * C, C};
  return aao_pil_resize_u8(filter, in, out, N, C, H, W, oH, oW, is, os, nthreads);
}

/* uint8 via the reference harness: test.py:52-58 (x.float()), :72 (bicubic clamp), :75 (.byte() truncation). */
int aao_harness_u8(int filter, const uint8_t *in, uint8_t *out, int64_t N, int64_t C, int64_t H, int64_t W,
                   int64_t oH, int64_t oW, const int64_t is[4], const int64_t os[4], int nthreads) {
  if (N < 0 || C <= 0 || H <= 0 || W <= 0 || oH <= 0 || oW <= 0) return -2;
  if (N == 0) return 0;
  float *fin = (float *)malloc(sizeof(float) * (size_t)(N * C * H * W));
  float *fout = (float *)malloc(sizeof(float) * (size_t)(N * C * oH * oW));
  if (!fin || !fout) { free(fin); free(fout); return -3; }
  for (int64_t n = 0; n < N; n++)
    for (int64_t c = 0; c < C; c++)
      for (int64_t y = 0; y < H; y++)
        for (int64_t x = 0; x < W; x++)
          fin[((n * C + c) * H + y) * W + x] = (float)in[n * is[0] + c * is[1] + y * is[2] + x * is[3]];
  const int64_t cis[4] = {C * H * W, H * W, W, 1}, cos_[4] = {C * oH * oW, oH * oW, oW, 1};
  int rc = aao_forward_f32(filter, fin, fout, N, C, H, W, oH, oW, cis, cos_, 0, nthreads);
  if (rc == 0) {
    for (int64_t n = 0; n < N; n++)
      for (int64_t c = 0; c < C; c++)
        for (int64_t y = 0; y < oH; y++)
          for (int64_t x = 0; x < oW; x++) {
            float v = fout[((n * C + c) * oH + y) * oW + x];
            if (filter == AAO_FILTER_CUBIC) { if (v < 0.f) v = 0.f; if (v > 255.f) v = 255.f; }
            /* torch .byte() on CPU: float -> int64 truncation -> low 8 bits */
            out[n * os[0] + c * os[1] + y * os[2] + x * os[3]] = (uint8_t)(int64_t)v;
          }
  }
  free(fin); free(fout);
  return rc;
}
