// aa_backward.hip — scatter form of the TRUE adjoint with float atomics (BASELINE config 5 names it).
//
// The reference's own backward header (step_two_dot_two/aa_interpolation_backward_impl.h:80-108) is the stock
// NON-antialiased 2x2-tap scatter and ignores `antialias` (:170-182); its API shape is kept
// (ti_upsample_bilinear2d_backward_cpu :185-219: zero-filled grad_input, then scatter-add) but the arithmetic is
// the adjoint of the AA forward, built from the same weight tables.
//
// The default backward is the gather form (aa_resample_bwd, transposed tables, no atomics, write-once); this file
// is the alternative that needs no transposed tables: V^T scatter into a zeroed [.,H,oW] intermediate, then H^T
// scatter into the zeroed grad_input.  Each wave-instruction adds to consecutive addresses (the shape the atomic
// units run at full rate for), one atomic per tap.

#include "aa_common.h"

namespace {

// pass 1: tmp[p][ymin[oy]+j][e] += w[oy][j] * go[p][oy][e]
template <typename T>
__global__ void __launch_bounds__(256)
vT_scatter(const T *__restrict__ go, T *__restrict__ tmp, const char *__restrict__ table, int64_t total, int H, int oH,
           int64_t rowlen, int ksize) {
  const TableView<T> tv = make_table_view<T>(table, oH, ksize);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
    const int64_t e = idx % rowlen;
    const int64_t t = idx / rowlen;
    const int oy = (int)(t % oH);
    const int64_t p = t / oH;
    const T g = go[idx];
    int n = tv.xsize[oy];
    n = n > 1 ? n : 1;
    const T *w = tv.w + (size_t)oy * ksize;
    T *dst = tmp + (p * H + tv.xmin[oy]) * rowlen + e;
    for (int j = 0; j < n; j++) atomicAdd(dst + (int64_t)j * rowlen, w[j] * g);
  }
}

// pass 2: gi[row][xmin[ox]+j][ci] += w[ox][j] * tmp[row][ox][ci]
template <typename T>
__global__ void __launch_bounds__(256)
hT_scatter(const T *__restrict__ tmp, T *__restrict__ gi, const char *__restrict__ table, int64_t total, int W, int oW,
           int inner, int ksize) {
  const TableView<T> tv = make_table_view<T>(table, oW, ksize);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
    const int ci = (int)(idx % inner);
    const int64_t t = idx / inner;
    const int ox = (int)(t % oW);
    const int64_t row = t / oW;
    const T v = tmp[idx];
    int n = tv.xsize[ox];
    n = n > 1 ? n : 1;
    const T *w = tv.w + (size_t)ox * ksize;
    T *dst = gi + (row * W + tv.xmin[ox]) * inner + ci;
    for (int j = 0; j < n; j++) atomicAdd(dst + (int64_t)j * inner, w[j] * v);
  }
}

inline int grid_for(int64_t total) {
  int64_t b = (total + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

template <typename T>
int run(const AAProblem &p) {
  const bool nhwc = p.layout == AA_NHWC;
  const int inner = nhwc ? (int)p.C : 1;
  const int64_t planes = nhwc ? p.N : p.N * p.C;
  const int64_t rowlen = nhwc ? p.oW * p.C : p.oW;
  T *tmp = (T *)p.ws;
  const size_t tmp_bytes = (size_t)p.N * p.C * p.H * p.oW * sizeof(T);
  const size_t gi_bytes = (size_t)p.N * p.C * p.H * p.W * sizeof(T);
  if (hipMemsetAsync(tmp, 0, tmp_bytes, p.stream) != hipSuccess) return AA_ERR_HIP;
  if (hipMemsetAsync(p.out, 0, gi_bytes, p.stream) != hipSuccess) return AA_ERR_HIP;
  const int64_t t1 = planes * p.oH * rowlen;
  hipLaunchKernelGGL((vT_scatter<T>), dim3(grid_for(t1)), dim3(256), 0, p.stream, (const T *)p.in, tmp,
                     (const char *)p.ah.table_dev, t1, (int)p.H, (int)p.oH, rowlen, p.ah.ksize);
  const int64_t hrows = nhwc ? p.N * p.H : p.N * p.C * p.H;
  const int64_t t2 = hrows * p.oW * inner;
  hipLaunchKernelGGL((hT_scatter<T>), dim3(grid_for(t2)), dim3(256), 0, p.stream, (const T *)tmp, (T *)p.out,
                     (const char *)p.aw.table_dev, t2, (int)p.W, (int)p.oW, inner, p.aw.ksize);
  AA_HIP_CHECK_LAUNCH();
  return AA_OK;
}

}  // namespace

int aa_launch_bwd_atomic(const AAProblem &p) {
  if (p.dtype == AA_F32) return run<float>(p);
  if (p.dtype == AA_F64) return run<double>(p);
  return AA_ERR_BAD_DTYPE;
}
