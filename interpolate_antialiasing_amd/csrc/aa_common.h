// aa_common.h — shared host/device definitions for libaa_interp.so (gfx950 only; no CUDA paths).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/aa_interp.h"

#define AA_TABLE_MAGIC 0x42544141
#define AA_MAX_KSIZE 4096  // generic kernels loop over taps, so this only bounds sanity

// ---- packed table views --------------------------------------------------------------------------------
__host__ __device__ inline size_t aa_align16(size_t x) { return (x + 15) & ~(size_t)15; }
__host__ __device__ inline size_t aa_weight_elem_bytes(int kind) { return kind == AA_TABLE_F64 ? 8 : 4; }
__host__ __device__ inline size_t aa_table_xmin_off() { return sizeof(aa_table_header); }
__host__ __device__ inline size_t aa_table_xsize_off(int64_t out) { return sizeof(aa_table_header) + 4 * (size_t)out; }
__host__ __device__ inline size_t aa_table_w_off(int64_t out) {
  return aa_align16(sizeof(aa_table_header) + 8 * (size_t)out);
}
// end of the weight rows = start of the gather section
__host__ __device__ inline size_t aa_table_weights_end(int kind, int64_t out, int ksize) {
  return aa_align16(aa_table_w_off(out) + (size_t)out * (size_t)ksize * aa_weight_elem_bytes(kind));
}
// gather section (AA_TABLE_F32 and AA_TABLE_PIL tables; F32 also transposed): one 32-byte record per OUTPUT index
// {xmin, xsize, w[0..5]} — a table row in one scalar load for kernels whose vertical pass gathers (aa_fused_float_up.hip, the UPK mode of aa_fused_u8_v3_impl.h)
__host__ __device__ inline size_t aa_table_gather_bytes(int kind, int64_t out) { return (kind == AA_TABLE_F32 || kind == AA_TABLE_PIL) ? 32 * (size_t)out : 0; }
__host__ __device__ inline size_t aa_table_total_bytes(int kind, int64_t out, int ksize) {
  return aa_table_weights_end(kind, out, ksize) + aa_table_gather_bytes(kind, out);
}

template <typename WT>
struct TableView {
  const int32_t *xmin;
  const int32_t *xsize;
  const WT *w;
  int ksize;
};

template <typename WT>
__host__ __device__ inline TableView<WT> make_table_view(const void *table, int out_size, int ksize) {
  const char *p = (const char *)table;
  TableView<WT> v;
  v.xmin = (const int32_t *)(p + aa_table_xmin_off());
  v.xsize = (const int32_t *)(p + aa_table_xsize_off(out_size));
  v.w = (const WT *)(p + aa_table_w_off(out_size));
  v.ksize = ksize;
  return v;
}

// ---- experiment knobs ------------------------------------------------------------------------------------------------
// Environment variables are read by developer builds only (make TUNING=-DAA_V2_TUNING, tools/ab_build.sh): the shipped library
// never calls getenv, so its behaviour cannot depend on the ambient environment and a B = 1 call pays no lookups.
#include <stdlib.h>
#ifdef AA_V2_TUNING
inline const char *aa_knob(const char *name) { return getenv(name); }
#else
inline const char *aa_knob(const char *) { return nullptr; }
#endif
// store form of the up-scaling / backward kernel, set through aa_set_store_form() (tests force the streaming forms at small sizes):
// -1 automatic (by output size), 0 never streaming, 1 always streaming
extern int g_aa_store_form;
// plane groups of the fused uint8 kernel (aa_set_plane_groups): 1 = planar three-channel images run all three planes in one wave
extern int g_aa_plane_groups;

// ---- launch-error plumbing -------------------------------------------------------------------------------
#define AA_HIP_CHECK_LAUNCH()                 \
  do {                                        \
    if (hipGetLastError() != hipSuccess) return AA_ERR_HIP; \
  } while (0)

// entry points implemented in the .hip files and called from aa_api.cpp
bool aa_table_pair_fits(int64_t in_a, int64_t out_a, int64_t in_b, int64_t out_b);
int aa_launch_table_build_pair(int filter, int kind, int align_corners, int64_t in_a, int64_t out_a, double scale_a, int ksize_a, int sk_a, void *tab_a,
                               int64_t in_b, int64_t out_b, double scale_b, int ksize_b, int sk_b, void *tab_b, hipStream_t stream);
int aa_launch_table_build(int filter, int kind, int64_t in_size, int64_t out_size, int align_corners, double scale,
                          int ksize, int scatter_ksize, void *table_dev, hipStream_t stream);
// bytes of the scatter section appended to AA_TABLE_PIL tables (0 when scatter_ksize == 0)
__host__ __device__ inline size_t aa_table_scatter_pitch(int kind) { return kind == AA_TABLE_F64 ? 64 : 32; }
__host__ __device__ inline size_t aa_table_scatter_bytes(int kind, int64_t in_size, int scatter_ksize) {
  // one record per input index + a sentinel: 8 ints (32-bit weights), or {int32 first, int32 cc, double w[6], pad} = 64 bytes
  return scatter_ksize > 0 ? aa_table_scatter_pitch(kind) * ((size_t)in_size + 1) : 0;
}
int aa_launch_table_transpose(const aa_table_header &h, const void *table_dev, void *tr_dev, int tr_ksize,
                              hipStream_t stream);

struct AAProblem {
  const void *in;
  void *out;
  void *ws;
  size_t ws_bytes;
  int dtype, layout;
  int64_t N, C, H, W, oH, oW;
  aa_axis ah, aw;
  hipStream_t stream;
  // decode-adjacent conversion (aa_resample_fwd_u8_to_f32): uint8 in, float32 out, optional layout change and per-channel
  // (v - mean) / std.  out_f32 == 0: the output has the input's dtype and layout.
  int out_f32 = 0;
  int out_layout = AA_NCHW;
  int normalize = 0;
  float mean[4] = {0.f, 0.f, 0.f, 0.f};
  float std[4] = {1.f, 1.f, 1.f, 1.f};
  // strided input view (aa_resample_fwd_strided): rows are dense, but consecutive rows / images (channels_last) or planes (NCHW; n * C + c,
  // uniformly spaced) lie these many BYTES apart.  0 = the dense tensor.  Only the kernels that say so take pitched problems.
  int64_t in_row_pitch = 0, in_img_pitch = 0;
  int fast = 0;  // AA_FLAG_FAST: the caller accepts results within 1e-4 relative of the reference's (FMA accumulation)
};

// generic two-launch separable path (always available); returns variant name through *variant
int aa_launch_generic_fwd(const AAProblem &p, const char **variant);
int aa_launch_axis_fwd(const void *in, void *out, int dtype, int64_t outer, int64_t in_size, int64_t inner, const aa_axis &ax,
                       hipStream_t stream);
size_t aa_generic_workspace_bytes(int dtype, int kind_w, int64_t N, int64_t C, int64_t H, int64_t oW);
int aa_launch_generic_convert(const AAProblem &p, const char **variant);  // u8 -> f32 (+ layout, normalisation), two launches
// fused single-launch paths; return 1 when they took the problem, 0 when not applicable, <0 on error
int aa_try_fused_u8_nhwc(const AAProblem &p, const char **variant);
int aa_try_fused_float_nchw(const AAProblem &p, const char **variant);
int aa_try_fused_float_nchw_up(const AAProblem &p, const char **variant);  // H <= oH: adjoint (gather form), up-scaling
int aa_try_fused_float_nchw_fast(const AAProblem &p, const char **variant);  // tolerance mode (aa_fused_float.hip built with -DAA_F32_FAST_BUILD)
bool aa_fused_float_nchw_fast_applicable(int dtype, int layout, int64_t N, int64_t C, int64_t H, int64_t W, const aa_axis *ah, const aa_axis *aw);
int aa_try_fused_u8_nhwc_v3(const AAProblem &p, const char **variant);  // LDS-DMA staged, wave-autonomous, V pass in registers
// The *_applicable predicates hold EVERY reason a fused path can decline that does not depend on the pointers (shape, LDS
// size, grid size, dispatch widths): aa_workspace_bytes() answers 0 exactly when aa_resample_fwd() will not need one.
bool aa_fused_u8_v3_applicable(int dtype, int layout, int64_t N, int64_t C, int64_t H, int64_t W, const aa_axis *ah, const aa_axis *aw,
                               int out_f32 = 0, int out_layout = AA_NCHW);
bool aa_fused_u8_nhwc_applicable(int dtype, int layout, int64_t N, int64_t C, int64_t H, int64_t W, const aa_axis *ah, const aa_axis *aw);
bool aa_fused_float_nchw_up_applicable(int dtype, int layout, int64_t N, int64_t C, int64_t H, int64_t W, const aa_axis *ah, const aa_axis *aw);
bool aa_fused_float_nchw_applicable(int dtype, int layout, int64_t N, int64_t C, int64_t H, int64_t W, const aa_axis *ah, const aa_axis *aw);
// Input pixels the windows (tw taps each) of one strip of <= 64 consecutive outputs cover, from the spread the table
// kernel MEASURED (header.span64p1; explicit scale factors and align_corners make it differ from 63 * in / out).
// -1 = unknown (a caller that did not fill aa_axis.span64p1): the fused kernels decline.
inline int aa_strip_span_px(const aa_axis &aw, int tw) { return aw.span64p1 > 0 ? aw.span64p1 + tw : -1; }
// the same for a strip of 32 outputs: 31 steps are at most 11 runs of 3 steps, each bounded by the measured spread of 4
// neighbouring window starts (span4p1), and never more than the spread of 64
inline int aa_strip_span_px32(const aa_axis &aw, int tw) {
  if (aw.span64p1 <= 0 || aw.span4p1 <= 0) return -1;
  const int by4 = 11 * (aw.span4p1 - 1) + 1;
  return (by4 < aw.span64p1 ? by4 : aw.span64p1) + tw;
}
// ... and for a strip of 16 outputs (split windows: four lanes per pixel): 15 steps are 5 runs of 3
inline int aa_strip_span_px16(const aa_axis &aw, int tw) {
  if (aw.span64p1 <= 0 || aw.span4p1 <= 0) return -1;
  const int by4 = 5 * (aw.span4p1 - 1) + 1;
  return (by4 < aw.span64p1 ? by4 : aw.span64p1) + tw;
}
// (row bands never exceed 64, so a grid of `units * 64` workgroups bounds every launch)
inline bool aa_grid_fits(int64_t units) { return units > 0 && units <= (int64_t)0x7FFFFFFF / 64 - 8; }
// CU count of the current device (cached); 256 on MI355X
int aa_device_cu_count();
// Resident workgroups per CU of `kern` at `threads` threads and `lds` bytes of dynamic LDS, cached per (kernel, device,
// threads, lds) behind a mutex — the occupancy query costs microseconds, and a B = 1 call is ~10 us end to end.  -1 when
// the query fails (callers fall back to an estimate; it only feeds launch-shape heuristics, never results).
#include <mutex>
template <typename K>
int aa_resident_blocks(K kern, int threads, size_t lds) {
  // (K is the kernel's function-pointer TYPE, shared by every instantiation of a kernel template: the kernel itself is part of the key.
  //  Until round 3 it was not, and a kernel could be sized with the register budget of whichever instantiation had asked first with the
  //  same threads and LDS bytes — found when the plane-group kernels, whose LDS size is fixed, ran 40 % slower after a sibling had run)
  struct Entry { const void *kern; int dev, threads; size_t lds; int nb; };
  constexpr int kEntries = 256;
  static std::mutex mu;
  static Entry cache[kEntries];
  static int n = 0, next = 0;
  int dev = 0;
  (void)hipGetDevice(&dev);
  const void *id = (const void *)kern;
  std::lock_guard<std::mutex> lock(mu);
  for (int i = 0; i < n; i++)
    if (cache[i].kern == id && cache[i].dev == dev && cache[i].threads == threads && cache[i].lds == lds) return cache[i].nb;
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, threads, lds) != hipSuccess || nb <= 0) {
    (void)hipGetLastError();
    nb = -1;
  }
  cache[next] = Entry{id, dev, threads, lds, nb};  // (a full cache forgets its oldest entry)
  next = next + 1 == kEntries ? 0 : next + 1;
  if (n < kEntries) n++;
  return nb;
}

int aa_launch_probe_copy(const void *src, void *dst, size_t bytes, int form, hipStream_t stream);
// scatter-add adjoint
int aa_launch_bwd_atomic(const AAProblem &p);
