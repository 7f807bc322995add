/* asan_edge.c — the oracle's edge cases under AddressSanitizer + UBSan (TEST INFRASTRUCTURE, like everything under oracle/).
 *
 * The reference found its one memory bug this way: step_two_dot_one's unconditional `j < 2` unroll read tap 1 (weight AND
 * src_min[ids_stride]) whatever ids_size was, and README.md:507-520's ASAN recipe caught the heap-buffer-overflow for a window of
 * size 1 at the last row / column (fixed in step_two_dot_two/aa_interpolation_impl.h:45-51,75-80).  This driver runs the C
 * restatement over exactly those shapes — out = 1, in < ksize, windows of one tap ending at the last index, in = out, up-scaling,
 * one-pixel images — with every buffer malloc'ed at its exact size, so that any read or write one element past a window trips the
 * sanitizer.  Built by `make -C oracle asan`; run by tests/test_oracle_golden.py::test_oracle_edge_cases_under_asan.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "aa_oracle.h"

static int fails = 0;
#define CHECK(x) do { if (!(x)) { fprintf(stderr, "FAIL %s:%d %s\n", __FILE__, __LINE__, #x); fails++; } } while (0)

static void strides_nchw(int64_t C, int64_t H, int64_t W, int64_t s[4]) { s[0] = C * H * W; s[1] = H * W; s[2] = W; s[3] = 1; }

static void one_case(int filter, int64_t N, int64_t C, int64_t H, int64_t W, int64_t oH, int64_t oW, int align_corners) {
  const size_t nin = (size_t)(N * C * H * W), nout = (size_t)(N * C * oH * oW);
  int64_t si[4], so[4];
  strides_nchw(C, H, W, si);
  strides_nchw(C, oH, oW, so);
  /* tables: exact-size buffers */
  for (int axis = 0; axis < 2; axis++) {
    const int64_t in = axis ? W : H, out = axis ? oW : oH;
    const int kf = aao_ksize(filter, in, out, align_corners, 0.0, 0), kd = aao_ksize(filter, in, out, align_corners, 0.0, 1);
    CHECK(kf > 0 && kd > 0);
    int64_t *xmin = malloc(sizeof(int64_t) * (size_t)out), *xsize = malloc(sizeof(int64_t) * (size_t)out);
    float *wf = malloc(sizeof(float) * (size_t)(out * kf));
    double *wd = malloc(sizeof(double) * (size_t)(out * kd));
    CHECK(aao_weights_f32(filter, in, out, align_corners, 0.0, xmin, xsize, wf) >= 0);
    for (int64_t i = 0; i < out; i++) CHECK(xmin[i] >= 0 && xsize[i] >= 0 && xmin[i] + xsize[i] <= in && xsize[i] <= kf);
    CHECK(aao_weights_f64(filter, in, out, align_corners, 0.0, xmin, xsize, wd) >= 0);
    for (int64_t i = 0; i < out; i++) CHECK(xmin[i] >= 0 && xsize[i] >= 0 && xmin[i] + xsize[i] <= in && xsize[i] <= kd);
    const int kp = aao_pil_ksize(filter, in, out);
    CHECK(kp > 0);
    int32_t *pm = malloc(4 * (size_t)out), *ps = malloc(4 * (size_t)out), *kk = malloc(4 * (size_t)(out * kp));
    CHECK(aao_pil_coeffs(filter, in, out, pm, ps, kk, NULL) >= 0);
    for (int64_t i = 0; i < out; i++) CHECK(pm[i] >= 0 && ps[i] >= 0 && pm[i] + ps[i] <= in && ps[i] <= kp);
    free(xmin); free(xsize); free(wf); free(wd); free(pm); free(ps); free(kk);
  }
  float *xf = malloc(sizeof(float) * nin), *yf = malloc(sizeof(float) * nout), *gf = malloc(sizeof(float) * nin);
  double *xd = malloc(sizeof(double) * nin), *yd = malloc(sizeof(double) * nout), *gd = malloc(sizeof(double) * nin);
  uint8_t *xb = malloc(nin), *yb = malloc(nout), *yh = malloc(nout);
  for (size_t i = 0; i < nin; i++) { xb[i] = (uint8_t)(i * 37 + 11); xf[i] = (float)xb[i]; xd[i] = (double)xb[i]; }
  CHECK(aao_forward_f32(filter, xf, yf, N, C, H, W, oH, oW, si, so, align_corners, 1) == 0);
  CHECK(aao_forward_f64(filter, xd, yd, N, C, H, W, oH, oW, si, so, align_corners, 2) == 0);
  CHECK(aao_backward_f32(filter, yf, gf, N, C, H, W, oH, oW, align_corners) == 0);
  CHECK(aao_backward_f64(filter, yd, gd, N, C, H, W, oH, oW, align_corners) == 0);
  if (!align_corners) {
    CHECK(aao_pil_resize_u8(filter, xb, yb, N, C, H, W, oH, oW, si, so, 1) == 0);
    CHECK(aao_harness_u8(filter, xb, yh, N, C, H, W, oH, oW, si, so, 1) == 0);
    /* channels_last storage through the NHWC entry point: the same numbers */
    uint8_t *xl = malloc(nin), *yl = malloc(nout);
    for (int64_t n = 0; n < N; n++)
      for (int64_t c = 0; c < C; c++)
        for (int64_t h = 0; h < H; h++)
          for (int64_t w = 0; w < W; w++) xl[((n * H + h) * W + w) * C + c] = xb[((n * C + c) * H + h) * W + w];
    CHECK(aao_pil_resize_u8_nhwc(filter, xl, yl, N, H, W, C, oH, oW, 1) == 0);
    for (int64_t n = 0; n < N; n++)
      for (int64_t c = 0; c < C; c++)
        for (int64_t h = 0; h < oH; h++)
          for (int64_t w = 0; w < oW; w++) CHECK(yl[((n * oH + h) * oW + w) * C + c] == yb[((n * C + c) * oH + h) * oW + w]);
    free(xl); free(yl);
  }
  /* a constant image stays constant (weights sum to one): catches a window that reads a neighbour instead of faulting */
  for (size_t i = 0; i < nin; i++) xf[i] = 7.0f;
  CHECK(aao_forward_f32(filter, xf, yf, N, C, H, W, oH, oW, si, so, align_corners, 1) == 0);
  for (size_t i = 0; i < nout; i++) CHECK(yf[i] > 6.9999f && yf[i] < 7.0001f);
  if (H == oH && W == oW && filter != AAO_FILTER_CUBIC) { /* identity */
    for (size_t i = 0; i < nin; i++) xf[i] = (float)xb[i];
    CHECK(aao_forward_f32(filter, xf, yf, N, C, H, W, oH, oW, si, so, align_corners, 1) == 0);
    for (size_t i = 0; i < nout; i++) CHECK(yf[i] == xf[i]);
  }
  free(xf); free(yf); free(gf); free(xd); free(yd); free(gd); free(xb); free(yb); free(yh);
}

int main(void) {
  static const int64_t shapes[][4] = {
      /* H, W, oH, oW */
      {1, 1, 1, 1},   {1, 1, 3, 5},    {2, 3, 1, 1},   {7, 5, 1, 1},     /* out = 1: one window over everything */
      {2, 2, 1, 2},   {3, 2, 2, 1},    {2, 5, 2, 2},                     /* in < ksize */
      {5, 7, 5, 7},   {4, 4, 4, 4},                                      /* in = out: one tap of weight 1 at the last index */
      {5, 7, 11, 13}, {3, 4, 9, 10},   {2, 2, 7, 7},                     /* up-scaling: plain 2- / 4-tap windows, clipped at both ends */
      {17, 23, 5, 7}, {12, 17, 5, 7},  {61, 53, 17, 23}, {9, 100, 4, 3}, /* down-scaling with ragged last windows */
      {33, 2, 3, 2},  {2, 33, 2, 3},   {64, 1, 7, 1},   {1, 64, 1, 7},   /* one-pixel-wide images */
  };
  for (size_t s = 0; s < sizeof(shapes) / sizeof(shapes[0]); s++)
    for (int filter = 0; filter < 3; filter++)
      for (int ac = 0; ac < 2; ac++) one_case(filter, 2, 3, shapes[s][0], shapes[s][1], shapes[s][2], shapes[s][3], ac);
  if (fails) { fprintf(stderr, "%d checks failed\n", fails); return 1; }
  printf("ASAN_EDGE_OK %zu shapes x 3 filters x 2 align_corners\n", sizeof(shapes) / sizeof(shapes[0]));
  return 0;
}
