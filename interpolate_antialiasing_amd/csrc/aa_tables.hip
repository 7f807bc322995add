// aa_tables.hip — device-side precompute of per-output xmin / xsize / normalised filter weights.
//
// Replaces HelperInterpBase::_compute_indices_weights_aa (reference step_two_dot_two/aa_interpolation_impl.h:195-281;
// older copies step_three/aa_interpolation_impl.h:345-419, step_three/aa_separable_single_dim_loop2d_impl.h:303-377),
// which runs serially on the calling CPU thread on every call and every pass, and allocates five tensors.  Here it
// is one launch, one thread per output index, writing ONE packed buffer that is cached by the host and can be
// broadcast to other GPUs as a single small RCCL message.
//
// Built with -ffp-contract=off: a fused multiply-add in `center`, `xmin` or the filter argument moves a window by
// one pixel (SURVEY §7 "Weight parity").  Float divisions are evaluated in double and narrowed, which is the
// correctly rounded float quotient (53 >= 2*24+2 bits), so the result does not depend on the fp32 division mode.

#include "aa_common.h"

namespace {

// ---- filters (reference s2.2:292-300, :410-424, :367-372): argument type scalar_t, evaluated in double ----
template <typename S>
__device__ inline S filt_linear(S x) {
  if (x < 0.0) x = -x;
  if (x < 1.0) return (S)(1.0 - (double)x);
  return (S)0.0;
}
template <typename S>
__device__ inline S filt_cubic(S x) {
  const double a = -0.5;
  if (x < 0.0) x = -x;
  const double xd = (double)x;
  // first branch: the double literals promote every operation to double
  if (x < 1.0) return (S)(((a + 2.0) * xd - (a + 3.0)) * xd * xd + 1);
  // second branch: `(((x - 5) * x + 8) * x - 4)` has only scalar_t and int operands, so the reference evaluates it
  // in scalar_t (float for float tensors); only the final `* a` is a double product
  if (x < 2.0) return (S)((double)(((x - (S)5) * x + (S)8) * x - (S)4) * a);
  return (S)0.0;
}
template <typename S>
__device__ inline S filt_box(S x) {
  return (x > -0.5 && x <= 0.5) ? (S)1.0 : (S)0.0;
}
template <typename S>
__device__ inline S apply_filter(int filter, S x) {
  return filter == AA_FILTER_LINEAR ? filt_linear<S>(x) : (filter == AA_FILTER_CUBIC ? filt_cubic<S>(x) : filt_box<S>(x));
}

__device__ inline int interp_size_of(int filter) { return filter == AA_FILTER_LINEAR ? 2 : (filter == AA_FILTER_CUBIC ? 4 : 1); }

// Reference arithmetic, scalar_t = float (s2.2:207-209, :242, :253-278).  Every promotion spelled out.
__device__ void table_build_f32_one(int i, int filter, int in_size, int out_size, int ksize, float scale, char *table) {
  int32_t *xmin_p = (int32_t *)(table + aa_table_xmin_off());
  int32_t *xsize_p = (int32_t *)(table + aa_table_xsize_off(out_size));
  float *w = (float *)(table + aa_table_w_off(out_size)) + (size_t)i * ksize;
  int32_t *max_taps = &((aa_table_header *)table)->max_taps;

  const int interp_size = interp_size_of(filter);
  const float support = (scale >= 1.0) ? (float)((interp_size * 0.5) * (double)scale) : (float)(interp_size * 0.5);
  const float invscale = (scale >= 1.0) ? (float)(1.0 / (double)scale) : 1.0f;

  const float center = (float)((double)scale * ((double)i + 0.5));
  long long xmin = (long long)((double)(float)(center - support) + 0.5);
  if (xmin < 0) xmin = 0;
  long long xmax = (long long)((double)(float)(center + support) + 0.5);
  if (xmax > in_size) xmax = in_size;
  xmax -= xmin;
  xmin_p[i] = (int32_t)xmin;
  xsize_p[i] = (int32_t)xmax;

  float total_w = 0.0f;
  int j = 0;
  for (; j < xmax && j < ksize; j++) {
    const float t = (float)((float)(j + xmin) - center);
    const float arg = (float)(((double)t + 0.5) * (double)invscale);
    const float wj = apply_filter<float>(filter, arg);
    w[j] = wj;
    total_w = (float)(total_w + wj);
  }
  if (total_w != 0.0f) {
    for (int q = 0; q < j; q++) w[q] = (float)((double)w[q] / (double)total_w);
  }
  for (; j < ksize; j++) w[j] = 0.0f;
  atomicMax(max_taps, (int32_t)(xmax > 1 ? xmax : 1));
}
__global__ void table_build_f32(int filter, int in_size, int out_size, int ksize, float scale, char *table) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < out_size) table_build_f32_one(i, filter, in_size, out_size, ksize, scale, table);
}

// scalar_t = double
__device__ void table_build_f64_one(int i, int filter, int in_size, int out_size, int ksize, double scale, char *table) {
  int32_t *xmin_p = (int32_t *)(table + aa_table_xmin_off());
  int32_t *xsize_p = (int32_t *)(table + aa_table_xsize_off(out_size));
  double *w = (double *)(table + aa_table_w_off(out_size)) + (size_t)i * ksize;
  int32_t *max_taps = &((aa_table_header *)table)->max_taps;

  const int interp_size = interp_size_of(filter);
  const double support = (scale >= 1.0) ? (interp_size * 0.5) * scale : interp_size * 0.5;
  const double invscale = (scale >= 1.0) ? 1.0 / scale : 1.0;
  const double center = scale * ((double)i + 0.5);
  long long xmin = (long long)(center - support + 0.5);
  if (xmin < 0) xmin = 0;
  long long xmax = (long long)(center + support + 0.5);
  if (xmax > in_size) xmax = in_size;
  xmax -= xmin;
  xmin_p[i] = (int32_t)xmin;
  xsize_p[i] = (int32_t)xmax;
  double total_w = 0.0;
  int j = 0;
  for (; j < xmax && j < ksize; j++) {
    const double wj = apply_filter<double>(filter, ((double)(j + xmin) - center + 0.5) * invscale);
    w[j] = wj;
    total_w += wj;
  }
  if (total_w != 0.0) {
    for (int q = 0; q < j; q++) w[q] /= total_w;
  }
  for (; j < ksize; j++) w[j] = 0.0;
  atomicMax(max_taps, (int32_t)(xmax > 1 ? xmax : 1));
}
__global__ void table_build_f64(int filter, int in_size, int out_size, int ksize, double scale, char *table) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < out_size) table_build_f64_one(i, filter, in_size, out_size, ksize, scale, table);
}

// Pillow: precompute_coeffs + normalize_coeffs_8bpc (src/libImaging/Resample.c, cited by URL in the reference:
// README.md:18,40; s2.2/aa_interpolation_impl.h:289-291).  double coefficients -> 22-bit fixed point int32.
__device__ void table_build_pil_one(int i, int filter, int in_size, int out_size, int ksize, char *table) {
  int32_t *xmin_p = (int32_t *)(table + aa_table_xmin_off());
  int32_t *xsize_p = (int32_t *)(table + aa_table_xsize_off(out_size));
  int32_t *kk = (int32_t *)(table + aa_table_w_off(out_size)) + (size_t)i * ksize;
  int32_t *max_taps = &((aa_table_header *)table)->max_taps;

  double scale = (double)in_size / (double)out_size;
  double filterscale = scale < 1.0 ? 1.0 : scale;
  const double fsupport = filter == AA_FILTER_LINEAR ? 1.0 : (filter == AA_FILTER_CUBIC ? 2.0 : 0.5);
  const double support = fsupport * filterscale;
  const double center = 0.0 + ((double)i + 0.5) * scale;
  const double ss = 1.0 / filterscale;
  int xmin = (int)(center - support + 0.5);
  if (xmin < 0) xmin = 0;
  int xmax = (int)(center + support + 0.5);
  if (xmax > in_size) xmax = in_size;
  xmax -= xmin;
  xmin_p[i] = xmin;
  xsize_p[i] = xmax;
  // two sweeps (sum, then normalise + quantise) so no per-thread array is needed
  double ww = 0.0;
  for (int x = 0; x < xmax && x < ksize; x++) ww += apply_filter<double>(filter, ((double)(x + xmin) - center + 0.5) * ss);
  int x = 0;
  for (; x < xmax && x < ksize; x++) {
    double k = apply_filter<double>(filter, ((double)(x + xmin) - center + 0.5) * ss);
    if (ww != 0.0) k /= ww;
    kk[x] = (k < 0) ? (int32_t)(-0.5 + k * (double)(1 << 22)) : (int32_t)(0.5 + k * (double)(1 << 22));
  }
  for (; x < ksize; x++) kk[x] = 0;
  atomicMax(max_taps, (int32_t)(xmax > 1 ? xmax : 1));
}
__global__ void table_build_pil(int filter, int in_size, int out_size, int ksize, char *table) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < out_size) table_build_pil_one(i, filter, in_size, out_size, ksize, table);
}

// header.span64p1 = 1 + max_i (xmin[min(i+63, out-1)] - xmin[i]): the fused kernels stage, per strip of <= 64 consecutive
// outputs, the input range their windows cover; they size that range from this MEASURED spread (an explicit scale
// factor or align_corners moves the windows apart differently from in/out).  Runs after the table's own kernel.
__device__ void table_span_one(int i, char *table, int out_size) {
  const int32_t *xmin = (const int32_t *)(table + aa_table_xmin_off());
  const int j = i + 63 < out_size ? i + 63 : out_size - 1;
  int d = xmin[j] - xmin[i];
  if (d < 0) d = 0;
  atomicMax(&((aa_table_header *)table)->span64p1, d + 1);
  const int j4 = i + 3 < out_size ? i + 3 : out_size - 1;
  int d4 = xmin[j4] - xmin[i];
  if (d4 < 0) d4 = 0;
  atomicMax(&((aa_table_header *)table)->span4p1, d4 + 1);
}
__global__ void table_span_kernel(char *table, int out_size) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < out_size) table_span_one(i, table, out_size);
}

// gather section: record i = {xmin[i], xsize[i], w[i][0..5]} (weights beyond ksize are 0): one scalar load per output row
__device__ void table_gather_one(int i, char *table, int out_size, int ksize, int gather_off) {
  const int32_t *xmin = (const int32_t *)(table + aa_table_xmin_off());
  const int32_t *xsize = (const int32_t *)(table + aa_table_xsize_off(out_size));
  const int32_t *w = (const int32_t *)(table + aa_table_w_off(out_size)) + (size_t)i * ksize;  // float bits
  int32_t *rec = (int32_t *)(table + gather_off) + (size_t)i * 8;
  rec[0] = xmin[i];
  rec[1] = xsize[i];
  for (int k = 0; k < 6; k++) rec[2 + k] = k < ksize ? w[k] : 0;
}
__global__ void table_gather_kernel(char *table, int out_size, int ksize, int gather_off) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < out_size) table_gather_one(i, table, out_size, ksize, gather_off);
}

__global__ void table_write_header(aa_table_header h, char *table) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *(aa_table_header *)table = h;
}

// ---- adjoint (gather-form) table -----------------------------------------------------------------------
// For input index x: outputs whose window [xmin, xmin+max(xsize,1)) holds x form a contiguous range because
// xmin[] and xmin[]+xsize[] are non-decreasing.  tmin[x] = first such output, tsize[x] = their count,
// tw[x][k] = w[tmin+k][x - xmin[tmin+k]].
template <typename WT>
__global__ void table_transpose_kernel(const char *fwd, int32_t *tmin, int32_t *tsize, WT *tw_all, int32_t *max_taps,
                                       int in_size, int out_size, int ksize, int tr_ksize) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= in_size) return;
  const int32_t *xmin = (const int32_t *)(fwd + aa_table_xmin_off());
  const int32_t *xsize = (const int32_t *)(fwd + aa_table_xsize_off(out_size));
  const WT *w = (const WT *)(fwd + aa_table_w_off(out_size));
  WT *tw = tw_all + (size_t)x * tr_ksize;

  // lo = first o with xmin[o] + max(xsize[o],1) > x
  int lo = 0, hi = out_size;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    const int xs = xsize[mid] > 1 ? xsize[mid] : 1;
    if (xmin[mid] + xs > x) hi = mid; else lo = mid + 1;
  }
  const int first = lo;
  // end = first o with xmin[o] > x
  lo = first; hi = out_size;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (xmin[mid] > x) hi = mid; else lo = mid + 1;
  }
  int cnt = lo - first;
  if (cnt < 0) cnt = 0;
  if (cnt > tr_ksize) cnt = tr_ksize;  // cannot happen when tr_ksize comes from aa_table_transposed_ksize
  tmin[x] = cnt > 0 ? first : (first < out_size ? first : out_size - 1);  // (kept monotone: the span measurement and the fused kernels' segments rely on it)
  tsize[x] = cnt;
  int k = 0;
  for (; k < cnt; k++) {
    const int o = first + k;
    tw[k] = w[(size_t)o * ksize + (x - xmin[o])];
  }
  for (; k < tr_ksize; k++) tw[k] = (WT)0;
  atomicMax(max_taps, cnt > 1 ? cnt : 1);
}

// Scatter section of AA_TABLE_PIL tables: one 32-byte record per INPUT index x, read by the fused kernels with a
// single s_load_dwordx8: {first output fed, number of outputs fed, weight in output first+0 .. first+5}.
template <typename WT>
__device__ void table_scatter_one(int x, const char *fwd, int32_t *rec_all, int32_t *scatter_max, int in_size, int out_size, int ksize,
                                  const int32_t *xmin_fast = nullptr, const int32_t *xsize_fast = nullptr) {
  // (xmin_fast / xsize_fast: copies of the two arrays in LDS — the binary searches below are chains of ~30 dependent loads)
  const int32_t *xmin = xmin_fast ? xmin_fast : (const int32_t *)(fwd + aa_table_xmin_off());
  const int32_t *xsize = xsize_fast ? xsize_fast : (const int32_t *)(fwd + aa_table_xsize_off(out_size));
  const WT *w = (const WT *)(fwd + aa_table_w_off(out_size));  // (int32 and float weights travel bit for bit; doubles as doubles)
  constexpr int REC_INTS = sizeof(WT) == 8 ? 16 : 8;
  int32_t *rec = rec_all + (size_t)x * REC_INTS;
  auto first_ending_after = [&](int row) {  // first o whose last input row, xmin[o] + max(xsize[o],1) - 1, is >= row
    int lo = 0, hi = out_size;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      const int xs = xsize[mid] > 1 ? xsize[mid] : 1;
      if (xmin[mid] + xs > row) hi = mid; else lo = mid + 1;
    }
    return lo;
  };
  const int first = first_ending_after(x);
  const int completes = first_ending_after(x + 1) - first;  // outputs [first, first+completes) end exactly at row x
  int lo = first, hi = out_size;
  while (lo < hi) {  // first o with xmin[o] > x
    const int mid = (lo + hi) >> 1;
    if (xmin[mid] > x) hi = mid; else lo = mid + 1;
  }
  const int cnt = (x < in_size && lo - first > 0) ? lo - first : 0;
  rec[0] = first;
  rec[1] = cnt | (completes << 16);
  WT *rw = (WT *)(rec + 2);  // (8-byte aligned: records are 32 or 64 bytes)
  for (int k = 0; k < 6; k++) {
    const int o = first + k;
    rw[k] = (k < cnt) ? w[(size_t)o * ksize + (x - xmin[o])] : (WT)0;
  }
  atomicMax(scatter_max, cnt > 1 ? cnt : 1);
}
template <typename WT>
__global__ void table_scatter_kernel(const char *fwd, int32_t *rec_all, int32_t *scatter_max, int in_size, int out_size,
                                     int ksize) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x <= in_size) table_scatter_one<WT>(x, fwd, rec_all, scatter_max, in_size, out_size, ksize);  // (record in_size is a sentinel: feeds nothing,
                                                                                                    //  completes nothing; readers may prefetch it)
}

// One launch for a whole table (round 3): header, rows, scatter records, measured spreads, gather records — the phases of the five launches
// above, run by ONE workgroup with a barrier between them.  A table is a few thousand entries of a few dozen operations: the five launches
// cost five launch latencies (a cold call = a shape never seen: two tables = ~0.1 ms of them; a data pipeline of random crops meets a
// new shape every call).  Same device functions, same results.
struct TableJob { aa_table_header h; int in_size, out_size, ksize; double scale; char *table; };
template <int KIND>
__device__ void table_build_body(const aa_table_header &h, int filter, int in_size, int out_size, int ksize, double scale, char *table);
template <int KIND>
__global__ void __launch_bounds__(1024) table_build_all(aa_table_header h, int filter, int in_size, int out_size, int ksize, double scale, char *table) {
  table_build_body<KIND>(h, filter, in_size, out_size, ksize, scale, table);
}
// ... and the two tables of a call (H and W axis) as the two workgroups of one launch
template <int KIND>
__global__ void __launch_bounds__(1024) table_build_pair(TableJob a, TableJob b, int filter) {
  const TableJob &j = blockIdx.x == 0 ? a : b;
  table_build_body<KIND>(j.h, filter, j.in_size, j.out_size, j.ksize, j.scale, j.table);
}
template <int KIND>
__device__ void table_build_body(const aa_table_header &h, int filter, int in_size, int out_size, int ksize, double scale, char *table) {
  if (threadIdx.x == 0) *(aa_table_header *)table = h;
  __syncthreads();
  for (int i = threadIdx.x; i < out_size; i += blockDim.x) {
    if constexpr (KIND == AA_TABLE_F32) table_build_f32_one(i, filter, in_size, out_size, ksize, (float)scale, table);
    else if constexpr (KIND == AA_TABLE_F64) table_build_f64_one(i, filter, in_size, out_size, ksize, scale, table);
    else table_build_pil_one(i, filter, in_size, out_size, ksize, table);
  }
  __threadfence();
  __syncthreads();
  // window starts and sizes into LDS for the scatter phase's binary searches (dependent loads: ~1 us each from memory, ~0.1 from LDS)
  constexpr int kFast = 4096;
  __shared__ int32_t s_xmin[kFast], s_xsize[kFast];
  const bool fast = out_size <= kFast;
  if (fast) {
    const int32_t *gx = (const int32_t *)(table + aa_table_xmin_off());
    const int32_t *gs = (const int32_t *)(table + aa_table_xsize_off(out_size));
    for (int i = threadIdx.x; i < out_size; i += blockDim.x) { s_xmin[i] = gx[i]; s_xsize[i] = gs[i]; }
    __syncthreads();
  }
  if (h.scatter_off) {
    int32_t *rec_all = (int32_t *)(table + h.scatter_off);
    int32_t *smax = &((aa_table_header *)table)->scatter_max;
    for (int x = threadIdx.x; x <= in_size; x += blockDim.x) {
      if constexpr (KIND == AA_TABLE_F64) table_scatter_one<double>(x, table, rec_all, smax, in_size, out_size, ksize, fast ? s_xmin : nullptr, fast ? s_xsize : nullptr);
      else table_scatter_one<int32_t>(x, table, rec_all, smax, in_size, out_size, ksize, fast ? s_xmin : nullptr, fast ? s_xsize : nullptr);
    }
  }
  for (int i = threadIdx.x; i < out_size; i += blockDim.x) {
    table_span_one(i, table, out_size);
    if (h.gather_off) table_gather_one(i, table, out_size, ksize, h.gather_off);
  }
}

}  // namespace

static aa_table_header make_header(int filter, int kind, int64_t in_size, int64_t out_size, int align_corners, int ksize, int scatter_ksize);

bool aa_table_pair_fits(int64_t in_a, int64_t out_a, int64_t in_b, int64_t out_b) {
  return out_a <= 16384 && in_a <= 32768 && out_b <= 16384 && in_b <= 32768;
}

int aa_launch_table_build_pair(int filter, int kind, int align_corners, int64_t in_a, int64_t out_a, double scale_a, int ksize_a, int sk_a, void *tab_a,
                               int64_t in_b, int64_t out_b, double scale_b, int ksize_b, int sk_b, void *tab_b, hipStream_t stream) {
  TableJob a = {make_header(filter, kind, in_a, out_a, align_corners, ksize_a, sk_a), (int)in_a, (int)out_a, ksize_a, scale_a, (char *)tab_a};
  TableJob b = {make_header(filter, kind, in_b, out_b, align_corners, ksize_b, sk_b), (int)in_b, (int)out_b, ksize_b, scale_b, (char *)tab_b};
  if (kind == AA_TABLE_F32) hipLaunchKernelGGL(table_build_pair<AA_TABLE_F32>, dim3(2), dim3(1024), 0, stream, a, b, filter);
  else if (kind == AA_TABLE_F64) hipLaunchKernelGGL(table_build_pair<AA_TABLE_F64>, dim3(2), dim3(1024), 0, stream, a, b, filter);
  else hipLaunchKernelGGL(table_build_pair<AA_TABLE_PIL>, dim3(2), dim3(1024), 0, stream, a, b, filter);
  AA_HIP_CHECK_LAUNCH();
  return AA_OK;
}

static aa_table_header make_header(int filter, int kind, int64_t in_size, int64_t out_size, int align_corners, int ksize, int scatter_ksize) {
  aa_table_header h = {};
  h.magic = AA_TABLE_MAGIC;
  h.filter = filter;
  h.kind = kind;
  h.in_size = (int32_t)in_size;
  h.out_size = (int32_t)out_size;
  h.ksize = ksize;
  h.align_corners = align_corners;
  h.max_taps = 0;
  h.transposed = 0;
  h.scatter_off = 0;
  h.scatter_ksize = 0;
  h.scatter_max = 0;
  h.span64p1 = 0;
  h.span4p1 = 0;
  h.gather_off = (kind == AA_TABLE_F32 || kind == AA_TABLE_PIL) ? (int32_t)aa_table_weights_end(kind, out_size, ksize) : 0;
  if (scatter_ksize > 0) {
    h.scatter_off = (int32_t)aa_table_total_bytes(kind, out_size, ksize);
    h.scatter_ksize = scatter_ksize;
  }
  return h;
}

int aa_launch_table_build(int filter, int kind, int64_t in_size, int64_t out_size, int align_corners, double scale,
                          int ksize, int scatter_ksize, void *table_dev, hipStream_t stream) {
  const aa_table_header h = make_header(filter, kind, in_size, out_size, align_corners, ksize, scatter_ksize);
  char *t = (char *)table_dev;
  if (out_size <= 16384 && in_size <= 32768) {  // one launch, one workgroup (see table_build_all); larger tables: the five launches below
    if (kind == AA_TABLE_F32)
      hipLaunchKernelGGL(table_build_all<AA_TABLE_F32>, dim3(1), dim3(1024), 0, stream, h, filter, (int)in_size, (int)out_size, ksize, scale, t);
    else if (kind == AA_TABLE_F64)
      hipLaunchKernelGGL(table_build_all<AA_TABLE_F64>, dim3(1), dim3(1024), 0, stream, h, filter, (int)in_size, (int)out_size, ksize, scale, t);
    else
      hipLaunchKernelGGL(table_build_all<AA_TABLE_PIL>, dim3(1), dim3(1024), 0, stream, h, filter, (int)in_size, (int)out_size, ksize, scale, t);
    AA_HIP_CHECK_LAUNCH();
    return AA_OK;
  }
  hipLaunchKernelGGL(table_write_header, dim3(1), dim3(64), 0, stream, h, t);
  const int threads = 128;
  const int blocks = (int)((out_size + threads - 1) / threads);
  if (kind == AA_TABLE_F32) {
    hipLaunchKernelGGL(table_build_f32, dim3(blocks), dim3(threads), 0, stream, filter, (int)in_size, (int)out_size, ksize,
                       (float)scale, t);
    if (h.scatter_off) {  // float weights travel through the 32-bit record fields bit for bit
      const int b2 = (int)((in_size + 1 + threads - 1) / threads);
      hipLaunchKernelGGL(table_scatter_kernel<int32_t>, dim3(b2), dim3(threads), 0, stream, (const char *)t,
                         (int32_t *)(t + h.scatter_off), &((aa_table_header *)t)->scatter_max, (int)in_size, (int)out_size,
                         ksize);
    }
  } else if (kind == AA_TABLE_F64) {
    hipLaunchKernelGGL(table_build_f64, dim3(blocks), dim3(threads), 0, stream, filter, (int)in_size, (int)out_size, ksize,
                       scale, t);
    if (h.scatter_off) {  // 64-byte records: double weights
      const int b2 = (int)((in_size + 1 + threads - 1) / threads);
      hipLaunchKernelGGL(table_scatter_kernel<double>, dim3(b2), dim3(threads), 0, stream, (const char *)t,
                         (int32_t *)(t + h.scatter_off), &((aa_table_header *)t)->scatter_max, (int)in_size, (int)out_size,
                         ksize);
    }
  } else {
    hipLaunchKernelGGL(table_build_pil, dim3(blocks), dim3(threads), 0, stream, filter, (int)in_size, (int)out_size, ksize,
                       t);
    if (h.scatter_off) {
      // scatter (adjoint-form) section for the fused kernels' in-register vertical pass: for every INPUT index the
      // outputs it feeds and their fixed-point weights
      const int b2 = (int)((in_size + 1 + threads - 1) / threads);
      hipLaunchKernelGGL(table_scatter_kernel<int32_t>, dim3(b2), dim3(threads), 0, stream, (const char *)t,
                         (int32_t *)(t + h.scatter_off), &((aa_table_header *)t)->scatter_max, (int)in_size, (int)out_size,
                         ksize);
    }
  }
  hipLaunchKernelGGL(table_span_kernel, dim3(blocks), dim3(threads), 0, stream, t, (int)out_size);
  if (h.gather_off) hipLaunchKernelGGL(table_gather_kernel, dim3(blocks), dim3(threads), 0, stream, t, (int)out_size, ksize, h.gather_off);
  AA_HIP_CHECK_LAUNCH();
  return AA_OK;
}

int aa_launch_table_transpose(const aa_table_header &fh, const void *table_dev, void *tr_dev, int tr_ksize,
                              hipStream_t stream) {
  aa_table_header h = fh;
  h.in_size = fh.out_size;
  h.out_size = fh.in_size;
  h.ksize = tr_ksize;
  h.max_taps = 0;
  h.transposed = 1;
  h.scatter_off = 0;  // the adjoint table has no scatter section (its buffer is sized by aa_table_bytes)
  h.scatter_ksize = 0;
  h.scatter_max = 0;
  h.span64p1 = 0;
  h.span4p1 = 0;
  h.gather_off = fh.kind == AA_TABLE_F32 ? (int32_t)aa_table_weights_end(fh.kind, fh.in_size, tr_ksize) : 0;
  char *t = (char *)tr_dev;
  hipLaunchKernelGGL(table_write_header, dim3(1), dim3(64), 0, stream, h, t);
  const int threads = 128;
  const int blocks = (fh.in_size + threads - 1) / threads;
  int32_t *tmin = (int32_t *)(t + aa_table_xmin_off());
  int32_t *tsize = (int32_t *)(t + aa_table_xsize_off(fh.in_size));
  int32_t *mt = &((aa_table_header *)t)->max_taps;
  if (fh.kind == AA_TABLE_F32) {
    hipLaunchKernelGGL(table_transpose_kernel<float>, dim3(blocks), dim3(threads), 0, stream, (const char *)table_dev, tmin,
                       tsize, (float *)(t + aa_table_w_off(fh.in_size)), mt, fh.in_size, fh.out_size, fh.ksize, tr_ksize);
  } else if (fh.kind == AA_TABLE_F64) {
    hipLaunchKernelGGL(table_transpose_kernel<double>, dim3(blocks), dim3(threads), 0, stream, (const char *)table_dev, tmin,
                       tsize, (double *)(t + aa_table_w_off(fh.in_size)), mt, fh.in_size, fh.out_size, fh.ksize, tr_ksize);
  } else {
    return AA_ERR_BAD_DTYPE;
  }
  hipLaunchKernelGGL(table_span_kernel, dim3(blocks), dim3(threads), 0, stream, t, fh.in_size);
  if (h.gather_off) hipLaunchKernelGGL(table_gather_kernel, dim3(blocks), dim3(threads), 0, stream, t, fh.in_size, tr_ksize, h.gather_off);
  AA_HIP_CHECK_LAUNCH();
  return AA_OK;
}
