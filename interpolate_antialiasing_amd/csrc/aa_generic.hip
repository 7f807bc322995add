// aa_generic.hip — generic separable path: one horizontal-pass launch + one vertical-pass launch with the
// intermediate in HBM.  Always applicable (any dtype / layout / ksize / scale); the fused single-launch
// kernels in aa_fused_*.hip take over for the shapes they cover.
//
// Replaces the reference's inner kernels and driver:
//   horizontal body  interpolate_aa_single_dim            s2.2/aa_interpolation_impl.h:60-87  (+ :102-120)
//   vertical body    interpolate_aa_single_dim_zero_strides                          :29-58   (+ :89-100)
//   driver           ti_separable_upsample_generic_Nd_kernel_impl (W pass -> temp -> H pass)  :628-683
// Arithmetic contract (bit-comparable with the reference's CPU build): tap 0 unconditionally, taps 1..xsize-1
// in order, product and sum rounded separately (this file is built with -ffp-contract=off).
//
// Both layouts collapse to the same two kernels:
//   NCHW: H-pass rows = N*C*H, inner = 1;       V-pass planes = N*C, rowlen = oW
//   NHWC: H-pass rows = N*H,   inner = C;       V-pass planes = N,   rowlen = oW*C
// so consecutive threads always touch consecutive output elements (coalesced stores; input rows are read
// through L1/L2 with ~scale-fold reuse between neighbouring threads).

#include "aa_common.h"

namespace {

// ---- per-pipeline arithmetic -----------------------------------------------------------------------------
// 16-bit floats: storage types only, all arithmetic is fp32 (PipeF32)
struct f16_t { _Float16 v; };
struct bf16_t { uint16_t v; };
__device__ inline float to_f32(float x) { return x; }
__device__ inline float to_f32(uint8_t x) { return (float)x; }
__device__ inline float to_f32(f16_t x) { return (float)x.v; }
__device__ inline float to_f32(bf16_t x) { return __uint_as_float((unsigned)x.v << 16); }

struct PipeF32 {
  using W = float; using Acc = float;
  static __device__ inline Acc init() { return -0.0f; }  // (-0) + p == p exactly, for every p: the first tap is an assignment
  template <typename T> static __device__ inline Acc first(T x, W w) { return to_f32(x) * w; }
  template <typename T> static __device__ inline Acc next(Acc a, T x, W w) { return a + to_f32(x) * w; }
};
struct PipeF64 {
  using W = double; using Acc = double;
  static __device__ inline Acc init() { return -0.0; }
  template <typename T> static __device__ inline Acc first(T x, W w) { return (double)x * w; }
  template <typename T> static __device__ inline Acc next(Acc a, T x, W w) { return a + (double)x * w; }
};
struct PipePIL {  // Pillow 8bpc: ss0 = 1 << (PRECISION_BITS-1); ss0 += pixel * k[x]
  using W = int32_t; using Acc = int32_t;
  static __device__ inline Acc init() { return 1 << 21; }
  template <typename T> static __device__ inline Acc first(T x, W w) { return (1 << 21) + (int32_t)x * w; }
  template <typename T> static __device__ inline Acc next(Acc a, T x, W w) { return a + (int32_t)x * w; }
};

template <typename TOut, typename Acc>
struct Store;
template <> struct Store<float, float> { static __device__ inline float cvt(float a, int) { return a; } };
template <> struct Store<f16_t, float> {  // round to nearest even
  static __device__ inline f16_t cvt(float a, int) { f16_t r; r.v = (_Float16)a; return r; }
};
template <> struct Store<bf16_t, float> {  // round to nearest even, NaN stays NaN
  static __device__ inline bf16_t cvt(float a, int) {
    const unsigned u = __float_as_uint(a);
    bf16_t r;
    if ((u & 0x7fffffffu) > 0x7f800000u) r.v = (uint16_t)((u >> 16) | 0x0040u);
    else r.v = (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
    return r;
  }
};
template <> struct Store<double, double> { static __device__ inline double cvt(double a, int) { return a; } };
template <> struct Store<uint8_t, int32_t> {  // clip8(ss >> PRECISION_BITS)
  static __device__ inline uint8_t cvt(int32_t a, int) {
    a >>= 22;
    return (uint8_t)(a < 0 ? 0 : (a > 255 ? 255 : a));
  }
};
template <> struct Store<uint8_t, float> {  // harness: (bicubic clamp, test.py:72) then truncating .byte() (test.py:75)
  static __device__ inline uint8_t cvt(float a, int) {
    a = a < 0.f ? 0.f : (a > 255.f ? 255.f : a);
    return (uint8_t)(int)a;
  }
};

// out[row][ox][ci] = sum_j in[row][xmin[ox]+j][ci] * w[ox][j]
// One thread owns one output column (ox, ci) and walks the rows R at a time: the column's window start, length and weights are
// read once per R rows, and the R rows' taps are R independent loads per weight (the dependent chain is only the sum, in tap
// order).  No per-element 64-bit divisions: the column index is a 32-bit quantity, the row index comes from the grid.
template <typename Pipe, typename TIn, typename TOut, int R>
__global__ void __launch_bounds__(256)
hpass_generic(const TIn *__restrict__ in, TOut *__restrict__ out, const char *__restrict__ table, int64_t rows,
              int W, int oW, int inner, int ksize) {
  using WT = typename Pipe::W;
  using Acc = typename Pipe::Acc;
  const TableView<WT> tv = make_table_view<WT>(table, oW, ksize);
  const int ncol = oW * inner;
  int col = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  const bool active = col < ncol;
  col = active ? col : 0;
  const int ox = col / inner;
  const int ci = col - ox * inner;
  const int xmin = tv.xmin[ox];
  int n = tv.xsize[ox];
  n = n > 1 ? n : 1;
  const WT *__restrict__ w = tv.w + (size_t)ox * ksize;
  const int64_t row_elems = (int64_t)W * inner;
  const TIn *__restrict__ col_src = in + (int64_t)xmin * inner + ci;
  for (int64_t row0 = (int64_t)blockIdx.y * R; row0 < rows; row0 += (int64_t)gridDim.y * R) {
    const TIn *src[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
      const int64_t row = row0 + r < rows ? row0 + r : rows - 1;  // (rows past the end re-read the last one; never stored)
      src[r] = col_src + row * row_elems;
    }
    Acc acc[R];
    const WT w0 = w[0];
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = Pipe::first(src[r][0], w0);
    for (int j = 1; j < n; j++) {
      const WT wj = w[j];
      const int64_t off = (int64_t)j * inner;
#pragma unroll
      for (int r = 0; r < R; r++) acc[r] = Pipe::next(acc[r], src[r][off], wj);
    }
    if (active) {
#pragma unroll
      for (int r = 0; r < R; r++)
        if (row0 + r < rows) out[(row0 + r) * ncol + col] = Store<TOut, Acc>::cvt(acc[r], 0);
    }
  }
}

// Wide windows (strong down-scaling: 17 .. 40 taps): the same sum, but a thread owns one output PIXEL (all INNER channels) and
// reads its window — xsize * INNER consecutive input elements — four elements per load instead of one per tap and channel.
// Only whole groups of four taps are read that way (they lie inside the window, so inside the row); the last xsize % 4 taps
// are read one by one.  Tap order and rounding are those of hpass_generic.
template <typename T>
struct alignas(sizeof(T)) Packed4 { T v[4]; };  // (element-aligned only: a window starts anywhere in the row)
template <typename T>
__device__ inline Packed4<T> load_packed4(const T *p) {
  Packed4<T> r;
  __builtin_memcpy(&r, p, sizeof(r));  // one 4/8/16/32-byte global load; the hardware accepts element alignment
  return r;
}

template <typename Pipe, typename TIn, typename TOut, int INNER, int R>
__global__ void __launch_bounds__(256)
hpass_wide(const TIn *__restrict__ in, TOut *__restrict__ out, const char *__restrict__ table, int64_t rows, int W, int oW, int ksize) {
  using WT = typename Pipe::W;
  using Acc = typename Pipe::Acc;
  const TableView<WT> tv = make_table_view<WT>(table, oW, ksize);
  int ox = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  const bool active = ox < oW;
  ox = active ? ox : 0;
  const int xmin = tv.xmin[ox];
  int n = tv.xsize[ox];
  n = n > 1 ? n : 1;
  const WT *__restrict__ w = tv.w + (size_t)ox * ksize;
  const int64_t row_elems = (int64_t)W * INNER;
  const TIn *__restrict__ col_src = in + (int64_t)xmin * INNER;
  const int nfull = n & ~3;  // taps covered by whole groups of four
  for (int64_t row0 = (int64_t)blockIdx.y * R; row0 < rows; row0 += (int64_t)gridDim.y * R) {
    const TIn *src[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
      const int64_t row = row0 + r < rows ? row0 + r : rows - 1;
      src[r] = col_src + row * row_elems;
    }
    Acc acc[R][INNER];
#pragma unroll
    for (int r = 0; r < R; r++)
#pragma unroll
      for (int c = 0; c < INNER; c++) acc[r][c] = Pipe::init();
    int j = 0;
    {
      for (; j < nfull; j += 4) {
        WT wj[4];
#pragma unroll
        for (int k = 0; k < 4; k++) wj[k] = w[j + k];
#pragma unroll
        for (int r = 0; r < R; r++) {
          TIn x[4 * INNER];  // 4 taps x INNER channels, consecutive in memory
#pragma unroll
          for (int q = 0; q < INNER; q++) {
            const Packed4<TIn> v = load_packed4(src[r] + (int64_t)j * INNER + 4 * q);
#pragma unroll
            for (int k = 0; k < 4; k++) x[4 * q + k] = v.v[k];
          }
#pragma unroll
          for (int k = 0; k < 4; k++)
#pragma unroll
            for (int c = 0; c < INNER; c++) acc[r][c] = Pipe::next(acc[r][c], x[k * INNER + c], wj[k]);
        }
      }
    }
    for (; j < n; j++) {
      const WT wj = w[j];
#pragma unroll
      for (int r = 0; r < R; r++)
#pragma unroll
        for (int c = 0; c < INNER; c++) {
          acc[r][c] = Pipe::next(acc[r][c], src[r][(int64_t)j * INNER + c], wj);
        }
    }
    if (active) {
#pragma unroll
      for (int r = 0; r < R; r++)
        if (row0 + r < rows) {
#pragma unroll
          for (int c = 0; c < INNER; c++) out[((row0 + r) * oW + ox) * INNER + c] = Store<TOut, Acc>::cvt(acc[r][c], 0);
        }
    }
  }
}

template <typename T, int V>
struct alignas(sizeof(T) * V) VecOf { T v[V]; };

// out[p][oy][e] = sum_j mid[p][ymin[oy]+j][e] * w[oy][j]
// A workgroup is 4 waves; each wave owns one output row (p, oy) at a time and 64 * VEC consecutive elements of it, so the row's
// window start, length and weights are wave-uniform (scalar loads, a uniform tap loop) and every tap is one fully coalesced load
// of 64 * VEC elements.
template <typename Pipe, typename TIn, typename TOut, int VEC>
__global__ void __launch_bounds__(256)
vpass_generic(const TIn *__restrict__ mid, TOut *__restrict__ out, const char *__restrict__ table, int64_t planes,
              int H, int oH, int64_t rowlen, int ksize) {
  using WT = typename Pipe::W;
  using Acc = typename Pipe::Acc;
  const TableView<WT> tv = make_table_view<WT>(table, oH, ksize);
  const int64_t e0 = ((int64_t)blockIdx.x * 64 + threadIdx.x) * VEC;
  const bool active = e0 < rowlen;  // (rowlen is a multiple of VEC: a lane's VEC elements exist together)
  const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.y);  // blockDim.x == 64: a wave has one threadIdx.y
  const int64_t total_rows = planes * oH;
  for (int64_t t = (int64_t)blockIdx.y * 4 + wv; t < total_rows; t += (int64_t)gridDim.y * 4) {
    const int oy = (int)(t % oH);
    const int64_t p = t / oH;
    const int ymin = tv.xmin[oy];
    int n = tv.xsize[oy];
    n = n > 1 ? n : 1;
    const WT *__restrict__ w = tv.w + (size_t)oy * ksize;
    const TIn *__restrict__ src = mid + (p * H + ymin) * rowlen + (active ? e0 : 0);
    Acc acc[VEC];
    {
      const VecOf<TIn, VEC> x = *(const VecOf<TIn, VEC> *)src;
      const WT w0 = w[0];
#pragma unroll
      for (int v = 0; v < VEC; v++) acc[v] = Pipe::first(x.v[v], w0);
    }
#pragma unroll 4
    for (int j = 1; j < n; j++) {
      const VecOf<TIn, VEC> x = *(const VecOf<TIn, VEC> *)(src + (int64_t)j * rowlen);
      const WT wj = w[j];
#pragma unroll
      for (int v = 0; v < VEC; v++) acc[v] = Pipe::next(acc[v], x.v[v], wj);
    }
    if (active) {
      VecOf<TOut, VEC> y;
#pragma unroll
      for (int v = 0; v < VEC; v++) y.v[v] = Store<TOut, Acc>::cvt(acc[v], 0);
      *(VecOf<TOut, VEC> *)(out + t * rowlen + e0) = y;
    }
  }
}

// Short rows (fewer than 256 elements): a wave per output row would leave most lanes idle, so here a thread owns one element
// (oy, e) of the output plane and walks the planes R at a time, like hpass_generic walks rows: the window and weights are read
// once per R planes, the R planes' taps are independent loads, and consecutive threads store consecutive elements.
template <typename Pipe, typename TIn, typename TOut, int R>
__global__ void __launch_bounds__(256)
vpass_short(const TIn *__restrict__ mid, TOut *__restrict__ out, const char *__restrict__ table, int64_t planes, int H, int oH,
            int rowlen, int ksize) {
  using WT = typename Pipe::W;
  using Acc = typename Pipe::Acc;
  const TableView<WT> tv = make_table_view<WT>(table, oH, ksize);
  const int plane_elems = oH * rowlen;  // (< 2^31: checked on the host)
  int idx = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  const bool active = idx < plane_elems;
  idx = active ? idx : 0;
  const int oy = idx / rowlen;
  const int e = idx - oy * rowlen;
  const int ymin = tv.xmin[oy];
  int n = tv.xsize[oy];
  n = n > 1 ? n : 1;
  const WT *__restrict__ w = tv.w + (size_t)oy * ksize;
  const int64_t in_plane = (int64_t)H * rowlen;
  const TIn *__restrict__ col_src = mid + (int64_t)ymin * rowlen + e;
  for (int64_t p0 = (int64_t)blockIdx.y * R; p0 < planes; p0 += (int64_t)gridDim.y * R) {
    const TIn *src[R];
#pragma unroll
    for (int r = 0; r < R; r++) src[r] = col_src + (p0 + r < planes ? p0 + r : planes - 1) * in_plane;
    Acc acc[R];
    const WT w0 = w[0];
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = Pipe::first(src[r][0], w0);
    for (int j = 1; j < n; j++) {
      const WT wj = w[j];
      const int64_t off = (int64_t)j * rowlen;
#pragma unroll
      for (int r = 0; r < R; r++) acc[r] = Pipe::next(acc[r], src[r][off], wj);
    }
    if (active) {
#pragma unroll
      for (int r = 0; r < R; r++)
        if (p0 + r < planes) out[(p0 + r) * plane_elems + idx] = Store<TOut, Acc>::cvt(acc[r], 0);
    }
  }
}

// rows of a pass spread over grid.y, at most 65535 workgroups there and about 32 workgroups per CU in all
inline unsigned grid_y_for(int64_t units, unsigned grid_x) {
  int64_t cap = (256 * 32 + grid_x - 1) / grid_x;
  if (cap > 65535) cap = 65535;
  if (cap < 1) cap = 1;
  int64_t y = units < cap ? units : cap;
  return (unsigned)(y < 1 ? 1 : y);
}

template <typename Pipe, typename TIn, typename TOut>
void launch_hpass(const TIn *in, TOut *out, const char *table, int64_t rows, int W, int oW, int inner, int ksize, hipStream_t stream) {
  if (rows <= 0 || oW <= 0) return;
  constexpr int R = 4;
  if (ksize >= 9 && (inner == 1 || inner == 3 || inner == 4)) {  // wide windows: a thread per pixel, four elements per load
    const int bs = oW >= 256 ? 256 : (oW + 63) / 64 * 64;
    const unsigned gx = (unsigned)((oW + bs - 1) / bs);
    if (inner == 1) {
      const unsigned gy = grid_y_for((rows + 3) / 4, gx);
      hipLaunchKernelGGL((hpass_wide<Pipe, TIn, TOut, 1, 4>), dim3(gx, gy), dim3(bs), 0, stream, in, out, table, rows, W, oW, ksize);
    } else {
      const unsigned gy = grid_y_for((rows + 1) / 2, gx);
      if (inner == 3) hipLaunchKernelGGL((hpass_wide<Pipe, TIn, TOut, 3, 2>), dim3(gx, gy), dim3(bs), 0, stream, in, out, table, rows, W, oW, ksize);
      else hipLaunchKernelGGL((hpass_wide<Pipe, TIn, TOut, 4, 2>), dim3(gx, gy), dim3(bs), 0, stream, in, out, table, rows, W, oW, ksize);
    }
    return;
  }
  const int ncol = oW * inner;
  const int bs = ncol >= 256 ? 256 : (ncol + 63) / 64 * 64;
  const unsigned gx = (unsigned)((ncol + bs - 1) / bs);
  const unsigned gy = grid_y_for((rows + R - 1) / R, gx);
  hipLaunchKernelGGL((hpass_generic<Pipe, TIn, TOut, R>), dim3(gx, gy), dim3(bs), 0, stream, in, out, table, rows, W, oW, inner, ksize);
}

template <typename Pipe, typename TIn, typename TOut>
void launch_vpass(const TIn *mid, TOut *out, const char *table, int64_t planes, int H, int oH, int64_t rowlen, int ksize,
                  hipStream_t stream) {
  if (planes <= 0 || oH <= 0 || rowlen <= 0) return;
  if (rowlen < 256 && (int64_t)oH * rowlen <= 0x7FFFFFFF) {  // short rows: a thread per element, planes walked 4 (or 1) at a time
    const int plane_elems = (int)(oH * rowlen);
    const unsigned gx = (unsigned)((plane_elems + 255) / 256);
    if (planes >= 4) {
      const unsigned gy = grid_y_for((planes + 3) / 4, gx);
      hipLaunchKernelGGL((vpass_short<Pipe, TIn, TOut, 4>), dim3(gx, gy), dim3(256), 0, stream, mid, out, table, planes, H, oH, (int)rowlen, ksize);
    } else {
      const unsigned gy = grid_y_for(planes, gx);
      hipLaunchKernelGGL((vpass_short<Pipe, TIn, TOut, 1>), dim3(gx, gy), dim3(256), 0, stream, mid, out, table, planes, H, oH, (int)rowlen, ksize);
    }
    return;
  }
  // 16 bytes of input per lane (4 floats; 16 bytes of a uint8 intermediate) when every row start is aligned for it and the rows
  // are long enough to keep the lanes busy; then 4 or 2 elements per lane; otherwise one
  auto fits = [&](int vw) {
    return rowlen % vw == 0 && rowlen >= 64 * vw && ((uintptr_t)mid % (vw * sizeof(TIn))) == 0 && ((uintptr_t)out % (vw * sizeof(TOut))) == 0;
  };
  const int vw = (sizeof(TIn) == 1 && fits(16)) ? 16 : (fits(4) ? 4 : (fits(2) ? 2 : 1));
  const unsigned gx = (unsigned)((rowlen + 64 * vw - 1) / (64 * vw));
  const unsigned gy = grid_y_for((planes * oH + 3) / 4, gx);
  if (vw == 16) {
    if constexpr (sizeof(TIn) == 1)
      hipLaunchKernelGGL((vpass_generic<Pipe, TIn, TOut, 16>), dim3(gx, gy), dim3(64, 4), 0, stream, mid, out, table, planes, H, oH, rowlen, ksize);
  } else if (vw == 4) {
    hipLaunchKernelGGL((vpass_generic<Pipe, TIn, TOut, 4>), dim3(gx, gy), dim3(64, 4), 0, stream, mid, out, table, planes, H, oH, rowlen, ksize);
  } else if (vw == 2) {  // (906-wide rows: 2 mod 4; measured on the channels_last backward: 1.35 -> 1.0 ms)
    hipLaunchKernelGGL((vpass_generic<Pipe, TIn, TOut, 2>), dim3(gx, gy), dim3(64, 4), 0, stream, mid, out, table, planes, H, oH, rowlen, ksize);
  } else {
    hipLaunchKernelGGL((vpass_generic<Pipe, TIn, TOut, 1>), dim3(gx, gy), dim3(64, 4), 0, stream, mid, out, table, planes, H, oH, rowlen, ksize);
  }
}

inline int grid_for(int64_t total) {
  int64_t b = (total + 255) / 256;
  const int64_t cap = 256 * 16;  // 256 CUs x 16 blocks, grid-stride beyond
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

template <typename Pipe, typename TIn, typename TMid, typename TOut>
int run_two_pass(const AAProblem &p) {
  const int64_t N = p.N, C = p.C, H = p.H, W = p.W, oH = p.oH, oW = p.oW;
  const bool nhwc = p.layout == AA_NHWC;
  const int inner = nhwc ? (int)C : 1;
  if (oW * inner > 0x7FFFFFFF || W * inner > 0x7FFFFFFF) return AA_ERR_BAD_SHAPE;
  const int64_t hrows = nhwc ? N * H : N * C * H;
  TMid *mid = (TMid *)p.ws;
  launch_hpass<Pipe, TIn, TMid>((const TIn *)p.in, mid, (const char *)p.aw.table_dev, hrows, (int)W, (int)oW, inner, p.aw.ksize, p.stream);
  const int64_t planes = nhwc ? N : N * C;
  const int64_t rowlen = nhwc ? oW * C : oW;
  launch_vpass<Pipe, TMid, TOut>((const TMid *)mid, (TOut *)p.out, (const char *)p.ah.table_dev, planes, (int)H, (int)oH, rowlen,
                                 p.ah.ksize, p.stream);
  AA_HIP_CHECK_LAUNCH();
  return AA_OK;
}

}  // namespace

size_t aa_generic_workspace_bytes(int dtype, int kind_w, int64_t N, int64_t C, int64_t H, int64_t oW) {
  size_t elem;
  if (dtype == AA_F64) elem = 8;
  else if (dtype == AA_F32 || dtype == AA_F16 || dtype == AA_BF16) elem = 4;  // 16-bit floats: fp32 intermediate
  else elem = (kind_w == AA_TABLE_PIL) ? 1 : 4;  // u8: Pillow keeps a uint8 intermediate, the harness an fp32 one
  return aa_align16((size_t)N * C * H * oW * elem);
}

int aa_launch_generic_fwd(const AAProblem &p, const char **variant) {
  if (p.dtype == AA_F32 && p.aw.kind == AA_TABLE_F32 && p.ah.kind == AA_TABLE_F32) {
    *variant = "generic_2pass_f32";
    return run_two_pass<PipeF32, float, float, float>(p);
  }
  if (p.dtype == AA_F64 && p.aw.kind == AA_TABLE_F64 && p.ah.kind == AA_TABLE_F64) {
    *variant = "generic_2pass_f64";
    return run_two_pass<PipeF64, double, double, double>(p);
  }
  if (p.dtype == AA_U8 && p.aw.kind == AA_TABLE_PIL && p.ah.kind == AA_TABLE_PIL) {
    *variant = "generic_2pass_u8_pil";
    return run_two_pass<PipePIL, uint8_t, uint8_t, uint8_t>(p);
  }
  if (p.dtype == AA_U8 && p.aw.kind == AA_TABLE_F32 && p.ah.kind == AA_TABLE_F32) {
    *variant = "generic_2pass_u8_harness";
    return run_two_pass<PipeF32, uint8_t, float, uint8_t>(p);
  }
  if (p.dtype == AA_F16 && p.aw.kind == AA_TABLE_F32 && p.ah.kind == AA_TABLE_F32) {
    *variant = "generic_2pass_f16";
    return run_two_pass<PipeF32, f16_t, float, f16_t>(p);
  }
  if (p.dtype == AA_BF16 && p.aw.kind == AA_TABLE_F32 && p.ah.kind == AA_TABLE_F32) {
    *variant = "generic_2pass_bf16";
    return run_two_pass<PipeF32, bf16_t, float, bf16_t>(p);
  }
  return AA_ERR_BAD_DTYPE;
}

// ---- decode-adjacent conversion, generic form: uint8 in -> fp32 horizontal pass (input layout) -> this vertical pass, which
// writes float32 in the REQUESTED layout (threads walk the output in its own order: coalesced stores) and applies the optional
// per-channel (v - mean) / std.  Same arithmetic as the harness path (PipeF32, taps in order).
namespace {
struct ConvertArgs { int in_nhwc, out_nhwc, normalize; float mean[4], std[4]; };
__global__ void __launch_bounds__(256)
vpass_convert(const float *__restrict__ mid, float *__restrict__ out, const char *__restrict__ table, int64_t total, int C, int H,
              int oH, int oW, int ksize, const ConvertArgs cv) {
  const TableView<float> tv = make_table_view<float>(table, oH, ksize);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
    int c, ox, oy;
    int64_t n;
    if (cv.out_nhwc) {
      c = (int)(idx % C); int64_t t = idx / C;
      ox = (int)(t % oW); t /= oW;
      oy = (int)(t % oH); n = t / oH;
    } else {
      ox = (int)(idx % oW); int64_t t = idx / oW;
      oy = (int)(t % oH); t /= oH;
      c = (int)(t % C); n = t / C;
    }
    const int ymin = tv.xmin[oy];
    int cnt = tv.xsize[oy];
    cnt = cnt > 1 ? cnt : 1;
    const float *w = tv.w + (size_t)oy * ksize;
    const float *src;
    int64_t rstride;
    if (cv.in_nhwc) { src = mid + (((n * H + ymin) * oW + ox) * C + c); rstride = (int64_t)oW * C; }
    else { src = mid + (((n * C + c) * H + ymin) * oW + ox); rstride = oW; }
    float acc = src[0] * w[0];
    for (int j = 1; j < cnt; j++) acc = acc + src[j * rstride] * w[j];
    if (cv.normalize) acc = (acc - cv.mean[c & 3]) / cv.std[c & 3];
    out[idx] = acc;
  }
}
}  // namespace

int aa_launch_generic_convert(const AAProblem &p, const char **variant) {
  if (p.dtype != AA_U8 || p.aw.kind != AA_TABLE_F32 || p.ah.kind != AA_TABLE_F32) return AA_ERR_BAD_DTYPE;
  const int64_t N = p.N, C = p.C, H = p.H, W = p.W, oH = p.oH, oW = p.oW;
  if (p.normalize && C > 4) return AA_ERR_BAD_SHAPE;
  const bool nhwc = p.layout == AA_NHWC;
  const int inner = nhwc ? (int)C : 1;
  const int64_t hrows = nhwc ? N * H : N * C * H;
  if (oW * inner > 0x7FFFFFFF || W * inner > 0x7FFFFFFF) return AA_ERR_BAD_SHAPE;
  float *mid = (float *)p.ws;
  launch_hpass<PipeF32, uint8_t, float>((const uint8_t *)p.in, mid, (const char *)p.aw.table_dev, hrows, (int)W, (int)oW, inner, p.aw.ksize,
                                        p.stream);
  ConvertArgs cv;
  cv.in_nhwc = nhwc; cv.out_nhwc = p.out_layout == AA_NHWC; cv.normalize = p.normalize;
  for (int i = 0; i < 4; i++) { cv.mean[i] = p.mean[i]; cv.std[i] = p.std[i]; }
  const int64_t vtotal = N * C * oH * oW;
  hipLaunchKernelGGL(vpass_convert, dim3(grid_for(vtotal)), dim3(256), 0, p.stream, (const float *)mid, (float *)p.out,
                     (const char *)p.ah.table_dev, vtotal, (int)C, (int)H, (int)oH, (int)oW, p.ah.ksize, cv);
  AA_HIP_CHECK_LAUNCH();
  *variant = "generic_2pass_u8_to_f32";
  return AA_OK;
}

// one pass along one axis of [outer][in_size][inner]
template <typename Pipe, typename T>
static int run_axis(const void *in, void *out, int64_t outer, int64_t in_size, int64_t inner, const aa_axis &ax, hipStream_t stream) {
  const int64_t total = outer * ax.out_size * inner;
  if (total == 0) return AA_OK;
  launch_vpass<Pipe, T, T>((const T *)in, (T *)out, (const char *)ax.table_dev, outer, (int)in_size, (int)ax.out_size, inner, ax.ksize, stream);
  AA_HIP_CHECK_LAUNCH();
  return AA_OK;
}

int aa_launch_axis_fwd(const void *in, void *out, int dtype, int64_t outer, int64_t in_size, int64_t inner, const aa_axis &ax,
                       hipStream_t stream) {
  if (dtype == AA_F32 && ax.kind == AA_TABLE_F32) return run_axis<PipeF32, float>(in, out, outer, in_size, inner, ax, stream);
  if (dtype == AA_F64 && ax.kind == AA_TABLE_F64) return run_axis<PipeF64, double>(in, out, outer, in_size, inner, ax, stream);
  if (dtype == AA_U8 && ax.kind == AA_TABLE_PIL) return run_axis<PipePIL, uint8_t>(in, out, outer, in_size, inner, ax, stream);
  if (dtype == AA_F16 && ax.kind == AA_TABLE_F32) return run_axis<PipeF32, f16_t>(in, out, outer, in_size, inner, ax, stream);
  if (dtype == AA_BF16 && ax.kind == AA_TABLE_F32) return run_axis<PipeF32, bf16_t>(in, out, outer, in_size, inner, ax, stream);
  return AA_ERR_BAD_DTYPE;
}


// ---- HBM copy-ceiling probe (bench.py: "measure the attainable ceiling on the box") -----------------------
namespace {
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
// form 0: one 16-byte element per thread; form 1: grid-stride over a grid of 32 workgroups per CU; form 2: four elements per
// thread, each wave-instruction a contiguous 1 KiB, all four loads in flight before the first store; form 3: form 2 with
// the streaming (nt) policy on loads and stores; form 4: WRITE ONLY (a fill of dst, nothing read); form 5: READ ONLY (src is
// summed, one dword per workgroup lands in dst).  bench.py reports the best copy form and the write-only / read-only rates.
template <int FORM>
__global__ void __launch_bounds__(256) probe_copy_kernel(const u32x4_t *__restrict__ src, u32x4_t *__restrict__ dst, size_t n16) {
  if constexpr (FORM == 0) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) dst[i] = src[i];
  } else if constexpr (FORM == 1) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
  } else if constexpr (FORM == 4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const u32x4_t v = {(unsigned)i, 1u, 2u, 3u};
    if (i < n16) dst[i] = v;
  } else if constexpr (FORM == 5) {
    const size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
    unsigned acc = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const size_t i = base + 256 * k;
      if (i < n16) { const u32x4_t v = src[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    }
    if (acc == 0x9e3779b9u) ((unsigned *)dst)[blockIdx.x] = acc;  // (practically never: keeps the loads alive)
  } else {
    const size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
    u32x4_t v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const size_t i = base + 256 * k;
      if (i < n16) v[k] = FORM == 3 ? __builtin_nontemporal_load(&src[i]) : src[i];
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const size_t i = base + 256 * k;
      if (i < n16) {
        if (FORM == 3) __builtin_nontemporal_store(v[k], &dst[i]);
        else dst[i] = v[k];
      }
    }
  }
}
}  // namespace

int aa_launch_probe_copy(const void *src, void *dst, size_t bytes, int form, hipStream_t stream) {
  const size_t n16 = bytes / 16;
  if (n16 == 0) return AA_OK;
  const u32x4_t *s = (const u32x4_t *)src;
  u32x4_t *d = (u32x4_t *)dst;
  if (form == 1) {
    size_t blocks = (n16 + 255) / 256;
    const size_t cap = (size_t)aa_device_cu_count() * 32;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(probe_copy_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, stream, s, d, n16);
  } else if (form == 0 || form == 4) {
    const size_t blocks = (n16 + 255) / 256;
    if (blocks > 0x7FFFFFFF) return AA_ERR_BAD_SHAPE;
    if (form == 0) hipLaunchKernelGGL(probe_copy_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, stream, s, d, n16);
    else hipLaunchKernelGGL(probe_copy_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, stream, s, d, n16);
  } else {
    const size_t blocks = (n16 + 1023) / 1024;
    if (blocks > 0x7FFFFFFF) return AA_ERR_BAD_SHAPE;
    if (form == 2) hipLaunchKernelGGL(probe_copy_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, stream, s, d, n16);
    else if (form == 5) hipLaunchKernelGGL(probe_copy_kernel<5>, dim3((unsigned)blocks), dim3(256), 0, stream, s, d, n16);
    else hipLaunchKernelGGL(probe_copy_kernel<3>, dim3((unsigned)blocks), dim3(256), 0, stream, s, d, n16);
  }
  AA_HIP_CHECK_LAUNCH();
  return AA_OK;
}
