// aa_api.hip — the C-ABI (include/aa_interp.h): argument checks, host-side ksize arithmetic, dispatch.
// No allocation, no synchronisation (except aa_table_query, documented), no torch types.

#include <math.h>
#include <string.h>

#include "aa_common.h"

int g_aa_store_form = -1;  // (aa_common.h; read by aa_fused_float_up.hip)
int g_aa_plane_groups = 1;  // (aa_common.h; read by aa_fused_u8_v3.hip)

namespace {

thread_local const char *g_last_variant = "none";
int g_fused_enabled = 1;

int interp_size_of(int filter) {
  switch (filter) {
    case AA_FILTER_LINEAR: return 2;  // s2.2/aa_interpolation_impl.h:287
    case AA_FILTER_CUBIC: return 4;   // :377
    case AA_FILTER_BOX: return 1;     // :333
    default: return -1;
  }
}

// ATen area_pixel_compute_scale<scalar_t> (UpSample.h; call site s2.2:314-315). scale<=0: not given.
double scale_for(int kind, int64_t in_size, int64_t out_size, int align_corners, double scale_opt) {
  if (kind == AA_TABLE_F32) {
    if (align_corners) return out_size > 1 ? (double)((float)(in_size - 1) / (float)(out_size - 1)) : 0.0;
    if (scale_opt > 0.) return (double)(float)(1.0 / scale_opt);
    return (double)((float)in_size / (float)out_size);
  }
  if (align_corners) return out_size > 1 ? (double)(in_size - 1) / (double)(out_size - 1) : 0.0;
  if (scale_opt > 0.) return 1.0 / scale_opt;
  return (double)in_size / (double)out_size;
}

int ksize_for(int filter, int kind, double scale) {
  const int interp_size = interp_size_of(filter);
  if (kind == AA_TABLE_PIL) {
    // Pillow precompute_coeffs: support = filter.support * max(scale,1); ksize = (int)ceil(support)*2+1
    const double fs = filter == AA_FILTER_LINEAR ? 1.0 : (filter == AA_FILTER_CUBIC ? 2.0 : 0.5);
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    return (int)ceil(fs * filterscale) * 2 + 1;
  }
  if (kind == AA_TABLE_F32) {
    const float s = (float)scale;
    const float support = (s >= 1.0) ? (float)((interp_size * 0.5) * s) : (float)(interp_size * 0.5);  // s2.2:208-209
    return (int)ceilf(support) * 2 + 1;                                                                  // s2.2:210
  }
  const double support = (scale >= 1.0) ? (interp_size * 0.5) * scale : interp_size * 0.5;
  return (int)ceilf((float)support) * 2 + 1;  // the reference calls ceilf() for double too
}

bool valid_kind(int kind) { return kind == AA_TABLE_PIL || kind == AA_TABLE_F32 || kind == AA_TABLE_F64; }

int check_axis(const aa_axis *a, int64_t in_size) {
  if (!a || !a->table_dev) return AA_ERR_NULL;
  if (a->in_size != in_size || a->out_size <= 0 || a->ksize <= 0) return AA_ERR_BAD_SHAPE;
  if (a->ksize > AA_MAX_KSIZE) return AA_ERR_KSIZE;
  if (!valid_kind(a->kind)) return AA_ERR_BAD_DTYPE;
  return AA_OK;
}

int check_dtype_kind(int dtype, int kh, int kw) {
  if (kh != kw) return AA_ERR_BAD_DTYPE;
  if (dtype == AA_F32 && kh == AA_TABLE_F32) return AA_OK;
  if (dtype == AA_F64 && kh == AA_TABLE_F64) return AA_OK;
  if (dtype == AA_U8 && (kh == AA_TABLE_PIL || kh == AA_TABLE_F32)) return AA_OK;
  if ((dtype == AA_F16 || dtype == AA_BF16) && kh == AA_TABLE_F32) return AA_OK;
  return AA_ERR_BAD_DTYPE;
}

}  // namespace

int aa_device_cu_count() {
  static int cached[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cached[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cached[dev] = n;
  }
  return cached[dev];
}

extern "C" {

size_t aa_table_build_bytes(int filter, int kind, int64_t in_size, int64_t out_size, int align_corners, double scale);

int aa_abi_version(void) { return AA_INTERP_ABI_VERSION; }

const char *aa_strerror(int status) {
  switch (status) {
    case AA_OK: return "ok";
    case AA_ERR_BAD_FILTER: return "unknown filter";
    case AA_ERR_BAD_DTYPE: return "dtype / weight-table kind combination not implemented";
    case AA_ERR_BAD_LAYOUT: return "layout must be AA_NCHW or AA_NHWC";
    case AA_ERR_BAD_SHAPE: return "Input and output sizes should be greater than 0 and match the weight tables";
    case AA_ERR_NULL: return "null pointer argument";
    case AA_ERR_WORKSPACE: return "workspace smaller than aa_workspace_bytes()";
    case AA_ERR_KSIZE: return "filter support (ksize) beyond the supported maximum";
    case AA_ERR_HIP: return "HIP kernel launch failed";
    case AA_ERR_STRIDES: return "input view not supported without a copy (rows must be dense, planes uniformly spaced, and a fused kernel must apply)";
    case AA_ERR_NO_DEVICE: return "no HIP device";
    default: return "unknown aa_status";
  }
}

int aa_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

int aa_table_ksize(int filter, int kind, int64_t in_size, int64_t out_size, int align_corners, double scale) {
  if (interp_size_of(filter) < 0) return AA_ERR_BAD_FILTER;
  if (!valid_kind(kind)) return AA_ERR_BAD_DTYPE;
  if (in_size <= 0 || out_size <= 0 || in_size > INT32_MAX / 4 || out_size > INT32_MAX / 4) return AA_ERR_BAD_SHAPE;
  if (kind == AA_TABLE_PIL && (align_corners || scale > 0.)) return AA_ERR_BAD_DTYPE;  // Pillow has neither
  const int k = ksize_for(filter, kind, scale_for(kind, in_size, out_size, align_corners, scale));
  if (k > AA_MAX_KSIZE) return AA_ERR_KSIZE;
  return k;
}

size_t aa_table_bytes(int kind, int64_t out_size, int ksize) {
  if (!valid_kind(kind) || out_size <= 0 || ksize <= 0) return 0;
  return aa_table_total_bytes(kind, out_size, ksize);
}

static int scatter_ksize_for(int filter, int kind, int64_t in_size, int64_t out_size) {
  (void)filter; (void)in_size; (void)out_size;
  // 32-bit weights only (a record holds 6).  Always present: whether an input index really feeds <= 6 outputs is
  // measured by the device kernel (header.scatter_max) and the fused kernels check that before using the section.
  return (kind == AA_TABLE_PIL || kind == AA_TABLE_F32 || kind == AA_TABLE_F64) ? 6 : 0;
}

size_t aa_table_build_bytes(int filter, int kind, int64_t in_size, int64_t out_size, int align_corners, double scale) {
  const int k = aa_table_ksize(filter, kind, in_size, out_size, align_corners, scale);
  if (k < 0) return 0;
  return aa_table_total_bytes(kind, out_size, k) + aa_table_scatter_bytes(kind, in_size, scatter_ksize_for(filter, kind, in_size, out_size));
}

int aa_table_build(int filter, int kind, int64_t in_size, int64_t out_size, int align_corners, double scale,
                   void *table_dev, size_t table_bytes, aa_stream_t stream) {
  const int k = aa_table_ksize(filter, kind, in_size, out_size, align_corners, scale);
  if (k < 0) return k;
  if (!table_dev) return AA_ERR_NULL;
  const int sk = scatter_ksize_for(filter, kind, in_size, out_size);
  if (table_bytes < aa_table_total_bytes(kind, out_size, k) + aa_table_scatter_bytes(kind, in_size, sk)) return AA_ERR_WORKSPACE;
  return aa_launch_table_build(filter, kind, in_size, out_size, align_corners,
                               scale_for(kind, in_size, out_size, align_corners, scale), k, sk, table_dev,
                               (hipStream_t)stream);
}

int aa_table_build2(int filter, int kind, int align_corners, int64_t in_a, int64_t out_a, double scale_a, void *table_a_dev, size_t bytes_a,
                    int64_t in_b, int64_t out_b, double scale_b, void *table_b_dev, size_t bytes_b, aa_stream_t stream) {
  const int ka = aa_table_ksize(filter, kind, in_a, out_a, align_corners, scale_a);
  if (ka < 0) return ka;
  const int kb = aa_table_ksize(filter, kind, in_b, out_b, align_corners, scale_b);
  if (kb < 0) return kb;
  if (!table_a_dev || !table_b_dev) return AA_ERR_NULL;
  const int ska = scatter_ksize_for(filter, kind, in_a, out_a), skb = scatter_ksize_for(filter, kind, in_b, out_b);
  if (bytes_a < aa_table_total_bytes(kind, out_a, ka) + aa_table_scatter_bytes(kind, in_a, ska)) return AA_ERR_WORKSPACE;
  if (bytes_b < aa_table_total_bytes(kind, out_b, kb) + aa_table_scatter_bytes(kind, in_b, skb)) return AA_ERR_WORKSPACE;
  if (!aa_table_pair_fits(in_a, out_a, in_b, out_b)) {  // very large tables: one after the other
    const int rc = aa_launch_table_build(filter, kind, in_a, out_a, align_corners, scale_for(kind, in_a, out_a, align_corners, scale_a), ka, ska, table_a_dev,
                                         (hipStream_t)stream);
    if (rc != AA_OK) return rc;
    return aa_launch_table_build(filter, kind, in_b, out_b, align_corners, scale_for(kind, in_b, out_b, align_corners, scale_b), kb, skb, table_b_dev,
                                 (hipStream_t)stream);
  }
  return aa_launch_table_build_pair(filter, kind, align_corners, in_a, out_a, scale_for(kind, in_a, out_a, align_corners, scale_a), ka, ska, table_a_dev, in_b,
                                    out_b, scale_for(kind, in_b, out_b, align_corners, scale_b), kb, skb, table_b_dev, (hipStream_t)stream);
}

int aa_table_transposed_ksize(int filter, int kind, int64_t in_size, int64_t out_size, int align_corners, double scale) {
  const int k = aa_table_ksize(filter, kind, in_size, out_size, align_corners, scale);
  if (k < 0) return k;
  if (kind == AA_TABLE_PIL) return AA_ERR_BAD_DTYPE;
  // an input index x lies in the windows of the outputs whose centre is within +-support of it: about
  // 2*support/scale of them (= interp_size when down-scaling, interp_size/scale when up-scaling); +3 covers the
  // integer rounding of both window ends.
  const double s = scale_for(kind, in_size, out_size, align_corners, scale);
  const int interp_size = interp_size_of(filter);
  const double support = (s >= 1.0) ? interp_size * 0.5 * s : interp_size * 0.5;
  double cover = (s > 0.) ? (2.0 * support + 1.0) / s : (double)out_size;
  int tk = (int)ceil(cover) + 3;
  if (tk > out_size) tk = (int)out_size;
  if (tk < 1) tk = 1;
  if (tk > AA_MAX_KSIZE) return AA_ERR_KSIZE;
  return tk;
}

// Header read-backs land in PINNED host memory (one small buffer per process, behind a mutex) and are copied out from there: an
// asynchronous device-to-host copy into pageable memory — the caller's struct — goes through the runtime's staging path and costs tens of
// microseconds more per copy (a cold call is two of them).
namespace {
std::mutex g_pin_mu;
aa_table_header *g_pin = nullptr;  // two headers
aa_table_header *pinned_headers() {  // (call with g_pin_mu held; nullptr: fall back to the caller's memory)
  if (!g_pin) {
    void *p = nullptr;
    if (hipHostMalloc(&p, 2 * sizeof(aa_table_header), hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
    g_pin = (aa_table_header *)p;
  }
  return g_pin;
}
}  // namespace

int aa_table_query(const void *table_dev, aa_table_header *host_header, aa_stream_t stream) {
  if (!table_dev || !host_header) return AA_ERR_NULL;
  std::lock_guard<std::mutex> lock(g_pin_mu);
  aa_table_header *pin = pinned_headers();
  aa_table_header *dst = pin ? pin : host_header;
  if (hipMemcpyAsync(dst, table_dev, sizeof(aa_table_header), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess)
    return AA_ERR_HIP;
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return AA_ERR_HIP;
  if (pin) *host_header = *pin;
  if (host_header->magic != AA_TABLE_MAGIC) return AA_ERR_BAD_SHAPE;
  return AA_OK;
}

int aa_table_query2(const void *table_a_dev, const void *table_b_dev, aa_table_header *host_a, aa_table_header *host_b, aa_stream_t stream) {
  if (!table_a_dev || !table_b_dev || !host_a || !host_b) return AA_ERR_NULL;
  std::lock_guard<std::mutex> lock(g_pin_mu);
  aa_table_header *pin = pinned_headers();
  aa_table_header *da = pin ? pin : host_a, *db = pin ? pin + 1 : host_b;
  if (hipMemcpyAsync(da, table_a_dev, sizeof(aa_table_header), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return AA_ERR_HIP;
  if (hipMemcpyAsync(db, table_b_dev, sizeof(aa_table_header), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return AA_ERR_HIP;
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return AA_ERR_HIP;
  if (pin) { *host_a = pin[0]; *host_b = pin[1]; }
  if (host_a->magic != AA_TABLE_MAGIC || host_b->magic != AA_TABLE_MAGIC) return AA_ERR_BAD_SHAPE;
  return AA_OK;
}

int aa_table_transpose(const void *table_dev, void *tr_table_dev, size_t tr_table_bytes, int tr_ksize,
                       aa_stream_t stream) {
  if (!table_dev || !tr_table_dev) return AA_ERR_NULL;
  aa_table_header h;
  int rc = aa_table_query(table_dev, &h, stream);  // table-build time only
  if (rc != AA_OK) return rc;
  if (h.kind == AA_TABLE_PIL) return AA_ERR_BAD_DTYPE;
  if (tr_ksize <= 0 || tr_ksize > AA_MAX_KSIZE) return AA_ERR_KSIZE;
  if (tr_table_bytes < aa_table_total_bytes(h.kind, h.in_size, tr_ksize)) return AA_ERR_WORKSPACE;
  return aa_launch_table_transpose(h, table_dev, tr_table_dev, tr_ksize, (hipStream_t)stream);
}

size_t aa_workspace_bytes(int dtype, int layout, int64_t N, int64_t C, int64_t H, int64_t W, int64_t oH, int64_t oW,
                          const aa_axis *ax_h, const aa_axis *ax_w) {
  (void)oH;
  if (!ax_h || !ax_w || N <= 0) return 0;
  if (g_fused_enabled) {
    if (g_fused_enabled == 1 && aa_fused_u8_v3_applicable(dtype, layout, N, C, H, W, ax_h, ax_w)) return 0;
    if (aa_fused_u8_nhwc_applicable(dtype, layout, N, C, H, W, ax_h, ax_w)) return 0;
    if (aa_fused_float_nchw_applicable(dtype, layout, N, C, H, W, ax_h, ax_w)) return 0;
    if (aa_fused_float_nchw_up_applicable(dtype, layout, N, C, H, W, ax_h, ax_w)) return 0;
  }
  return aa_generic_workspace_bytes(dtype, ax_w->kind, N, C, H, oW);
}

int aa_resample_fwd(const void *in_dev, void *out_dev, void *workspace_dev, size_t workspace_bytes, int dtype,
                    int layout, int64_t N, int64_t C, int64_t H, int64_t W, const aa_axis *ax_h, const aa_axis *ax_w,
                    aa_stream_t stream) {
  return aa_resample_fwd_ex(in_dev, out_dev, workspace_dev, workspace_bytes, dtype, layout, N, C, H, W, ax_h, ax_w, 0u, stream);
}

static int resample_fwd_impl(const void *in_dev, void *out_dev, void *workspace_dev, size_t workspace_bytes, int dtype,
                             int layout, int64_t N, int64_t C, int64_t H, int64_t W, const aa_axis *ax_h, const aa_axis *ax_w,
                             unsigned flags, int64_t row_pitch, int64_t img_pitch, aa_stream_t stream);

int aa_resample_fwd_ex(const void *in_dev, void *out_dev, void *workspace_dev, size_t workspace_bytes, int dtype,
                       int layout, int64_t N, int64_t C, int64_t H, int64_t W, const aa_axis *ax_h, const aa_axis *ax_w,
                       unsigned flags, aa_stream_t stream) {
  return resample_fwd_impl(in_dev, out_dev, workspace_dev, workspace_bytes, dtype, layout, N, C, H, W, ax_h, ax_w, flags, 0, 0, stream);
}

int aa_resample_fwd_strided(const void *in_dev, void *out_dev, int dtype, int layout, int64_t N, int64_t C, int64_t H, int64_t W,
                            const int64_t *in_strides, const aa_axis *ax_h, const aa_axis *ax_w, unsigned flags, aa_stream_t stream) {
  if (!in_strides) return AA_ERR_NULL;
  if (dtype < AA_U8 || dtype > AA_BF16) return AA_ERR_BAD_DTYPE;
  if (N < 0 || C <= 0 || H <= 0 || W <= 0) return AA_ERR_BAD_SHAPE;
  const int64_t es = dtype == AA_U8 ? 1 : (dtype == AA_F64 ? 8 : (dtype == AA_F32 ? 4 : 2));
  const int64_t sN = in_strides[0], sC = in_strides[1], sH = in_strides[2], sW = in_strides[3];
  int64_t row_pitch, img_pitch;
  if (layout == AA_NCHW) {  // rows of W consecutive elements; planes n * C + c uniformly spaced
    if (sW != 1 || sH < W || sC < 0 || sN < 0 || (C > 1 && N > 1 && sN != C * sC)) return AA_ERR_STRIDES;
    row_pitch = sH * es;
    img_pitch = (C > 1 ? sC : sN) * es;
  } else if (layout == AA_NHWC) {  // rows of W pixels of C consecutive channels; images anywhere
    if (sC != 1 || sW != C || sH < W * C || sN < 0) return AA_ERR_STRIDES;
    row_pitch = sH * es;
    img_pitch = sN * es;
  } else {
    return AA_ERR_BAD_LAYOUT;
  }
  const bool dense = row_pitch == W * (layout == AA_NHWC ? C : 1) * es && (N * (layout == AA_NCHW ? C : 1) <= 1 || img_pitch == H * row_pitch);
  if (dense) return resample_fwd_impl(in_dev, out_dev, nullptr, 0, dtype, layout, N, C, H, W, ax_h, ax_w, flags, 0, 0, stream);
  if (img_pitch == 0 && N * (layout == AA_NCHW ? C : 1) > 1) return AA_ERR_STRIDES;  // (a broadcast batch: make it dense)
  return resample_fwd_impl(in_dev, out_dev, nullptr, 0, dtype, layout, N, C, H, W, ax_h, ax_w, flags, row_pitch, img_pitch ? img_pitch : H * row_pitch, stream);
}

static int resample_fwd_impl(const void *in_dev, void *out_dev, void *workspace_dev, size_t workspace_bytes, int dtype,
                             int layout, int64_t N, int64_t C, int64_t H, int64_t W, const aa_axis *ax_h, const aa_axis *ax_w,
                             unsigned flags, int64_t row_pitch, int64_t img_pitch, aa_stream_t stream) {
  if (flags & ~(unsigned)AA_FLAG_FAST) return AA_ERR_BAD_SHAPE;
  if (dtype < AA_U8 || dtype > AA_BF16) return AA_ERR_BAD_DTYPE;
  if (layout != AA_NCHW && layout != AA_NHWC) return AA_ERR_BAD_LAYOUT;
  if (N < 0 || C <= 0 || H <= 0 || W <= 0) return AA_ERR_BAD_SHAPE;
  int rc = check_axis(ax_h, H);
  if (rc != AA_OK) return rc;
  rc = check_axis(ax_w, W);
  if (rc != AA_OK) return rc;
  rc = check_dtype_kind(dtype, ax_h->kind, ax_w->kind);
  if (rc != AA_OK) return rc;
  if (N == 0) {  // empty batch is allowed (s2.2:747-750)
    g_last_variant = "empty";
    return AA_OK;
  }
  if (!in_dev || !out_dev) return AA_ERR_NULL;
  {  // a tensor of 2 / 4 / 8-byte elements starts on an element boundary; anything else is not a tensor (and the kernels' dispatch must
     // not depend on the pointers: aa_workspace_bytes answers from the shape alone)
    const uintptr_t es = dtype == AA_U8 ? 1 : (dtype == AA_F64 ? 8 : (dtype == AA_F32 ? 4 : 2));
    if ((((uintptr_t)in_dev | (uintptr_t)out_dev) & (es - 1)) != 0) return AA_ERR_BAD_SHAPE;
  }

  AAProblem p;
  p.in = in_dev; p.out = out_dev; p.ws = workspace_dev; p.ws_bytes = workspace_bytes;
  p.dtype = dtype; p.layout = layout;
  p.N = N; p.C = C; p.H = H; p.W = W; p.oH = ax_h->out_size; p.oW = ax_w->out_size;
  p.ah = *ax_h; p.aw = *ax_w;
  p.stream = (hipStream_t)stream;
  p.in_row_pitch = row_pitch; p.in_img_pitch = img_pitch;
  // (Pillow's integer arithmetic and double arithmetic have no tolerance mode; uint8 images with AA_TABLE_F32 tables = the harness's float arithmetic do)
  p.fast = (flags & AA_FLAG_FAST) && dtype != AA_F64 && ax_w->kind == AA_TABLE_F32 ? 1 : 0;

  const char *variant = "none";
  rc = 0;
  if (g_fused_enabled) {
    if (g_fused_enabled == 1 && p.fast && dtype != AA_U8) rc = aa_try_fused_float_nchw_fast(p, &variant);  // declines -> the exact kernels (always within tolerance)
    if (rc == 0 && g_fused_enabled == 1) rc = aa_try_fused_u8_nhwc_v3(p, &variant);
    if (rc == 0 && !row_pitch) rc = aa_try_fused_u8_nhwc(p, &variant);  // (only the two main kernels take pitched views)
    if (rc == 0) rc = aa_try_fused_float_nchw(p, &variant);
    if (rc == 0 && !row_pitch) rc = aa_try_fused_float_nchw_up(p, &variant);
  }
  if (rc < 0) return rc;
  if (rc == 1) {
    g_last_variant = variant;
    return AA_OK;
  }
  if (row_pitch) return AA_ERR_STRIDES;  // no kernel for this view: the caller makes a dense copy (what the two-launch path needs anyway)
  const size_t need = aa_generic_workspace_bytes(dtype, ax_w->kind, N, C, H, p.oW);
  if (!workspace_dev || workspace_bytes < need) return AA_ERR_WORKSPACE;
  rc = aa_launch_generic_fwd(p, &variant);
  if (rc == AA_OK) g_last_variant = variant;
  return rc;
}

static int fill_convert(AAProblem &p, const aa_convert *cv, int64_t C) {
  if (!cv) return AA_ERR_NULL;
  if (cv->out_layout != AA_NCHW && cv->out_layout != AA_NHWC) return AA_ERR_BAD_LAYOUT;
  if (cv->normalize && C > 4) return AA_ERR_BAD_SHAPE;
  p.out_f32 = 1;
  p.out_layout = cv->out_layout;
  p.normalize = cv->normalize ? 1 : 0;
  for (int i = 0; i < 4; i++) { p.mean[i] = cv->mean[i]; p.std[i] = cv->std[i]; }
  if (cv->flags & ~(uint32_t)AA_FLAG_FAST) return AA_ERR_BAD_SHAPE;
  p.fast = (cv->flags & AA_FLAG_FAST) ? 1 : 0;
  return AA_OK;
}

size_t aa_workspace_bytes_u8_to_f32(int layout, int64_t N, int64_t C, int64_t H, int64_t W, const aa_axis *ax_h, const aa_axis *ax_w,
                                    const aa_convert *cv) {
  if (!ax_h || !ax_w || !cv || N <= 0) return 0;
  if (g_fused_enabled == 1 && aa_fused_u8_v3_applicable(AA_U8, layout, N, C, H, W, ax_h, ax_w, 1, cv->out_layout)) return 0;
  return aa_generic_workspace_bytes(AA_U8, AA_TABLE_F32, N, C, H, ax_w->out_size);
}

int aa_resample_fwd_u8_to_f32(const void *in_dev, void *out_dev, void *workspace_dev, size_t workspace_bytes, int layout,
                              int64_t N, int64_t C, int64_t H, int64_t W, const aa_axis *ax_h, const aa_axis *ax_w,
                              const aa_convert *cv, aa_stream_t stream) {
  if (layout != AA_NCHW && layout != AA_NHWC) return AA_ERR_BAD_LAYOUT;
  if (N < 0 || C <= 0 || H <= 0 || W <= 0) return AA_ERR_BAD_SHAPE;
  int rc = check_axis(ax_h, H);
  if (rc != AA_OK) return rc;
  rc = check_axis(ax_w, W);
  if (rc != AA_OK) return rc;
  if (ax_h->kind != AA_TABLE_F32 || ax_w->kind != AA_TABLE_F32) return AA_ERR_BAD_DTYPE;  // float output = float arithmetic
  AAProblem p;
  rc = fill_convert(p, cv, C);
  if (rc != AA_OK) return rc;
  if (N == 0) {
    g_last_variant = "empty";
    return AA_OK;
  }
  if (!in_dev || !out_dev) return AA_ERR_NULL;
  p.in = in_dev; p.out = out_dev; p.ws = workspace_dev; p.ws_bytes = workspace_bytes;
  p.dtype = AA_U8; p.layout = layout;
  p.N = N; p.C = C; p.H = H; p.W = W; p.oH = ax_h->out_size; p.oW = ax_w->out_size;
  p.ah = *ax_h; p.aw = *ax_w;
  p.stream = (hipStream_t)stream;
  const char *variant = "none";
  rc = g_fused_enabled == 1 ? aa_try_fused_u8_nhwc_v3(p, &variant) : 0;
  if (rc < 0) return rc;
  if (rc == 1) {
    g_last_variant = variant;
    return AA_OK;
  }
  const size_t need = aa_generic_workspace_bytes(AA_U8, AA_TABLE_F32, N, C, H, p.oW);
  if (!workspace_dev || workspace_bytes < need) return AA_ERR_WORKSPACE;
  rc = aa_launch_generic_convert(p, &variant);
  if (rc == AA_OK) g_last_variant = variant;
  return rc;
}

size_t aa_workspace_bytes_bwd(int dtype, int layout, int64_t N, int64_t C, int64_t H, int64_t W, int64_t oH, int64_t oW) {
  (void)layout; (void)H; (void)oW;
  if (N <= 0) return 0;
  // gather form: intermediate [N,C,oH,W]; scatter form: intermediate [N,C,H,oW]; size for the larger
  const size_t elem = dtype == AA_F64 ? 8 : 4;
  const size_t a = (size_t)N * C * oH * W, b = (size_t)N * C * H * oW;
  return aa_align16((a > b ? a : b) * elem);
}

int aa_resample_bwd(const void *grad_out_dev, void *grad_in_dev, void *workspace_dev, size_t workspace_bytes, int dtype,
                    int layout, int64_t N, int64_t C, int64_t H, int64_t W, const aa_axis *tr_h, const aa_axis *tr_w,
                    aa_stream_t stream) {
  if (dtype != AA_F32 && dtype != AA_F64) return AA_ERR_BAD_DTYPE;
  if (!tr_h || !tr_w) return AA_ERR_NULL;
  // The adjoint in gather form IS a forward resample of grad_out [N,C,oH,oW] with the transposed tables
  // (tr_w: oW -> W, tr_h: oH -> H); the two 1-D adjoints act on different axes and commute.
  if (tr_h->out_size != H || tr_w->out_size != W) return AA_ERR_BAD_SHAPE;
  return aa_resample_fwd(grad_out_dev, grad_in_dev, workspace_dev, workspace_bytes, dtype, layout, N, C, tr_h->in_size,
                         tr_w->in_size, tr_h, tr_w, stream);
}

int aa_resample_bwd_atomic(const void *grad_out_dev, void *grad_in_dev, void *workspace_dev, size_t workspace_bytes,
                           int dtype, int layout, int64_t N, int64_t C, int64_t H, int64_t W, const aa_axis *ax_h,
                           const aa_axis *ax_w, aa_stream_t stream) {
  if (dtype != AA_F32 && dtype != AA_F64) return AA_ERR_BAD_DTYPE;
  if (layout != AA_NCHW && layout != AA_NHWC) return AA_ERR_BAD_LAYOUT;
  if (N < 0 || C <= 0 || H <= 0 || W <= 0) return AA_ERR_BAD_SHAPE;
  int rc = check_axis(ax_h, H);
  if (rc != AA_OK) return rc;
  rc = check_axis(ax_w, W);
  if (rc != AA_OK) return rc;
  rc = check_dtype_kind(dtype, ax_h->kind, ax_w->kind);
  if (rc != AA_OK) return rc;
  if (N == 0) {
    g_last_variant = "empty";
    return AA_OK;
  }
  if (!grad_out_dev || !grad_in_dev) return AA_ERR_NULL;
  const size_t need = aa_workspace_bytes_bwd(dtype, layout, N, C, H, W, ax_h->out_size, ax_w->out_size);
  if (!workspace_dev || workspace_bytes < need) return AA_ERR_WORKSPACE;
  AAProblem p;
  p.in = grad_out_dev; p.out = grad_in_dev; p.ws = workspace_dev; p.ws_bytes = workspace_bytes;
  p.dtype = dtype; p.layout = layout;
  p.N = N; p.C = C; p.H = H; p.W = W; p.oH = ax_h->out_size; p.oW = ax_w->out_size;
  p.ah = *ax_h; p.aw = *ax_w;
  p.stream = (hipStream_t)stream;
  rc = aa_launch_bwd_atomic(p);
  if (rc == AA_OK) g_last_variant = "bwd_scatter_atomics";
  return rc;
}

int aa_resample_axis_fwd(const void *in_dev, void *out_dev, int dtype, int64_t outer, int64_t in_size, int64_t inner,
                         const aa_axis *ax, aa_stream_t stream) {
  if (dtype < AA_U8 || dtype > AA_BF16) return AA_ERR_BAD_DTYPE;
  if (outer < 0 || in_size <= 0 || inner <= 0) return AA_ERR_BAD_SHAPE;
  int rc = check_axis(ax, in_size);
  if (rc != AA_OK) return rc;
  if (dtype == AA_U8 && ax->kind != AA_TABLE_PIL) return AA_ERR_BAD_DTYPE;
  rc = check_dtype_kind(dtype, ax->kind, ax->kind);
  if (rc != AA_OK) return rc;
  if (outer == 0) return AA_OK;
  if (!in_dev || !out_dev) return AA_ERR_NULL;
  g_last_variant = "generic_axis";
  return aa_launch_axis_fwd(in_dev, out_dev, dtype, outer, in_size, inner, *ax, (hipStream_t)stream);
}

int aa_probe_copy(const void *src_dev, void *dst_dev, size_t bytes, int form, aa_stream_t stream) {
  if (!src_dev || !dst_dev) return AA_ERR_NULL;
  if ((((uintptr_t)src_dev | (uintptr_t)dst_dev) & 15) != 0 || form < 0 || form > 5) return AA_ERR_BAD_SHAPE;
  return aa_launch_probe_copy(src_dev, dst_dev, bytes, form, (hipStream_t)stream);
}

int aa_set_fused(int enabled) {
  const int prev = g_fused_enabled;
  // 0 generic two-pass, 1 auto (newest fused design first), 2 first-generation fused kernels only
  g_fused_enabled = (enabled < 0 || enabled > 2) ? 1 : enabled;
  return prev;
}

int aa_set_store_form(int form) {
  const int prev = g_aa_store_form;
  g_aa_store_form = (form < -1 || form > 1) ? -1 : form;
  return prev;
}

int aa_set_plane_groups(int enabled) {
  const int prev = g_aa_plane_groups;
  g_aa_plane_groups = enabled ? 1 : 0;
  return prev;
}

const char *aa_last_variant(void) { return g_last_variant; }

}  // extern "C"
