/*
 * aa_interp.h — C-ABI of libaa_interp.so: MI355X (gfx950) antialiased separable resample.
 *
 * This is the drop-in boundary for the reference's hot path.  The reference binds the path through a
 * pybind11 module (step_two_dot_two/extension_interpolate.cpp:46-51: linear_forward, nearest_forward,
 * cubic_forward, linear_backward; step_three/extension_interpolate.cpp:17-19: forward); a maintainer
 * replaces the bodies of those four wrappers (s2.2/extension_interpolate.cpp:7-42) with calls to the entry
 * points below (INTEGRATION.md shows the stub).  Plain pointers and sizes only: no torch/ATen types.
 *
 * Conventions
 *   - every pointer named *_dev is a DEVICE pointer (HBM); the caller allocates everything (outputs,
 *     tables, workspace); nothing here allocates, frees or synchronises unless its comment says so;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); all work is enqueued on it;
 *   - tensors are 4-D (N,C,H,W) in one of two dense layouts: AA_NCHW (contiguous) or AA_NHWC
 *     (torch.channels_last storage); the output uses the same layout as the input
 *     (s2.2/aa_interpolation_impl.h:752 `suggest_memory_format`);
 *   - return value: 0 on success, a negative aa_status otherwise (aa_strerror() gives the text).
 */
#ifndef AA_INTERP_H
#define AA_INTERP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AA_INTERP_ABI_VERSION 3 /* 3: aa_resample_fwd_ex (flags), aa_resample_fwd_strided, aa_set_store_form, aa_convert.flags; every other
                                   v2 entry point unchanged */

typedef void *aa_stream_t; /* hipStream_t */

enum aa_status {
  AA_OK = 0,
  AA_ERR_BAD_FILTER = -1,    /* unknown filter id */
  AA_ERR_BAD_DTYPE = -2,     /* dtype / table-kind combination not implemented ("... not implemented for 'X'") */
  AA_ERR_BAD_LAYOUT = -3,
  AA_ERR_BAD_SHAPE = -4,     /* non-positive sizes (upsample_2d_common_check), table/shape mismatch */
  AA_ERR_NULL = -5,
  AA_ERR_WORKSPACE = -6,     /* workspace smaller than aa_workspace_bytes() */
  AA_ERR_KSIZE = -7,         /* ksize beyond what the kernels support */
  AA_ERR_HIP = -8,           /* a HIP runtime call failed (launch error) */
  AA_ERR_NO_DEVICE = -9,
  AA_ERR_STRIDES = -10       /* aa_resample_fwd_strided: this view needs a dense copy (make one and call aa_resample_fwd) */
};

/* Filters: s2.2/aa_interpolation_impl.h:292-300 (triangle), :410-424 (Keys cubic a=-0.5), :367-372 (box;
 * the reference binds it as "nearest_forward": "it's not nearest but box", extension_interpolate.cpp:48). */
enum aa_filter { AA_FILTER_LINEAR = 0, AA_FILTER_CUBIC = 1, AA_FILTER_BOX = 2 };

/* Element type of the image tensors. */
enum aa_dtype { AA_U8 = 0, AA_F32 = 1, AA_F64 = 2, AA_F16 = 3, AA_BF16 = 4 };
/* AA_F16 / AA_BF16 (the reference dispatches float and double only): AA_TABLE_F32 tables, fp32
 * arithmetic in the reference's order, fp32 intermediate between the passes, ONE round-to-nearest-even at the store —
 * i.e. exactly  half(reference_fp32(float(x))). */

enum aa_layout { AA_NCHW = 0, AA_NHWC = 1 };

/* Arithmetic a weight table is built in (replaces HelperInterpBase::_compute_indices_weights_aa,
 * s2.2/aa_interpolation_impl.h:195-281):
 *   AA_TABLE_F32   the reference's scalar_t=float promotions, float weights   (f32 images; u8 "harness" mode)
 *   AA_TABLE_F64   scalar_t=double                                            (f64 images)
 *   AA_TABLE_PIL   Pillow's precompute_coeffs + normalize_coeffs_8bpc: double coefficients, int32 weights in
 *                  22-bit fixed point (u8 images, bit-exact with PIL.Image.resize; SURVEY §8 a-U)         */
enum aa_table_kind { AA_TABLE_PIL = 0, AA_TABLE_F32 = 1, AA_TABLE_F64 = 2 };

/* ---- packed weight table (one flat device buffer = one RCCL broadcast payload) -------------------------
 *   [ aa_table_header : 64 B ][ int32 xmin[out] ][ int32 xsize[out] ][ pad to 16 B ][ weight w[out*ksize] ]
 * weight = float (F32), double (F64) or int32 (PIL).  Rows are zero-padded from xsize to ksize (s2.2:276-278). */
typedef struct aa_table_header {
  int32_t magic;         /* 'AATB' 0x42544141 */
  int32_t filter;        /* aa_filter */
  int32_t kind;          /* aa_table_kind */
  int32_t in_size;
  int32_t out_size;
  int32_t ksize;         /* row pitch of w, = (int)ceilf(support)*2+1 (s2.2:210) */
  int32_t align_corners;
  int32_t max_taps;      /* max_i xsize[i], filled by the device kernel */
  int32_t transposed;    /* 1 for an adjoint (backward) table built by aa_table_transpose */
  int32_t scatter_off;   /* byte offset of the scatter section (0 = none), see below */
  int32_t scatter_ksize; /* row pitch of the scatter weights */
  int32_t scatter_max;   /* max outputs fed by one input index, filled by the device kernel */
  int32_t span64p1;      /* 1 + max_i (xmin[min(i+63,out-1)] - xmin[i]): how far the window starts of 64 consecutive outputs
                            spread, measured by the device kernel from the table itself (explicit scale factors and
                            align_corners make it differ from 63*in/out); 0 = not measured */
  int32_t span4p1;       /* the same over 4 consecutive outputs: 1 + max_i (xmin[min(i+3,out-1)] - xmin[i]) (kernels in which
                            a lane computes 4 neighbouring outputs from one shared window) */
  int32_t gather_off;    /* byte offset of the gather section (AA_TABLE_F32 and AA_TABLE_PIL tables; 0 = none): one 32-byte record per OUTPUT index,
                            { int32 xmin, int32 xsize, float w[6] } = a table row in one scalar load (rows wider than 6 taps keep
                            their full weights in w[] above) */
  int32_t reserved[1];
} aa_table_header;
/* Scatter section (every table kind), used by the fused kernels whose vertical pass runs in registers: one record per INPUT
 * index x (in_size + 1 records; the last is an all-zero sentinel a reader may prefetch).  AA_TABLE_PIL / AA_TABLE_F32: 32-byte
 * records { int32 first, int32 count | completes << 16, int32 (PIL) or float (F32) w[6] }; AA_TABLE_F64: 64-byte records
 * { int32 first, int32 count | completes << 16, double w[6], 8 bytes of padding }.  first = the first output whose window ends at or after
 * x; first .. first+count-1 = the outputs whose window holds x, w[k] = weight[first+k][x - xmin[first+k]] (zero
 * padded); completes = how many outputs, starting at `first`, have x as the LAST index of their window (they can be
 * emitted once x has been absorbed).  Present only when count <= 6 everywhere. */

/* Host-side description of one axis handed to the resample calls. */
typedef struct aa_axis {
  const void *table_dev; /* packed table in HBM */
  int32_t in_size;
  int32_t out_size;
  int32_t ksize;
  int32_t max_taps;      /* from aa_table_query(); 0 = unknown (kernels then use ksize) */
  int32_t kind;          /* aa_table_kind */
  int32_t filter;        /* aa_filter */
  int32_t scatter_off;   /* from the table header (aa_table_query); 0 = no scatter section */
  int32_t scatter_ksize;
  int32_t scatter_max;
  int32_t span64p1;      /* from the table header; 0 = unknown: the fused single-launch kernels then decline (they size
                            their staged row segments from it) and the generic two-launch path runs */
  int32_t span4p1;       /* from the table header; 0 = unknown */
  int32_t gather_off;    /* from the table header; 0 = no gather section */
  int32_t reserved[2];
} aa_axis;

/* Version / diagnostics. */
int aa_abi_version(void);
const char *aa_strerror(int status);
/* Number of visible HIP devices (0 on a CPU-only host; never throws). */
int aa_device_count(void);

/* ksize for (filter, kind, in, out): host arithmetic with the reference's promotions (s2.2:207-210), or
 * Pillow's for AA_TABLE_PIL.  scale<=0: scale derived from sizes (area_pixel_compute_scale, call site
 * s2.2:314-315).  Returns ksize (>0) or a negative aa_status. */
int aa_table_ksize(int filter, int kind, int64_t in_size, int64_t out_size, int align_corners, double scale);
/* Bytes of the packed table for (kind, out_size, ksize) without a scatter section (transposed tables); AA_TABLE_F32
 * and AA_TABLE_PIL tables include their gather section (32 bytes per output index). */
size_t aa_table_bytes(int kind, int64_t out_size, int ksize);
/* Bytes aa_table_build() needs for this table (AA_TABLE_PIL tables carry a scatter section as well). */
size_t aa_table_build_bytes(int filter, int kind, int64_t in_size, int64_t out_size, int align_corners, double scale);

/* Build a packed table ON DEVICE (one thread per output index).  Asynchronous on `stream`. */
int aa_table_build(int filter, int kind, int64_t in_size, int64_t out_size, int align_corners, double scale,
                   void *table_dev, size_t table_bytes, aa_stream_t stream);

/* Upper bound for the ksize of the adjoint of a table (host arithmetic only). */
int aa_table_transposed_ksize(int filter, int kind, int64_t in_size, int64_t out_size, int align_corners, double scale);
/* Build the adjoint (gather-form) table of `table_dev` on device: for every INPUT index x, the contiguous
 * range of outputs whose window holds x and their weights.  The result is a packed table whose in_size/out_size
 * are swapped, usable with aa_resample_fwd to compute the TRUE adjoint (what test.py:387-398 asks for; the
 * reference's own backward header is non-AA).  F32/F64 kinds only.  Asynchronous. */
int aa_table_transpose(const void *table_dev, void *tr_table_dev, size_t tr_table_bytes, int tr_ksize,
                       aa_stream_t stream);

/* Copy a table header back to the host.  SYNCHRONISES `stream` (one-off, at table-build time, never in the
 * per-call path). */
int aa_table_query(const void *table_dev, aa_table_header *host_header, aa_stream_t stream);
/* aa_table_build for the two tables of a call (H and W axis: same filter, kind and align_corners) as ONE launch. */
int aa_table_build2(int filter, int kind, int align_corners, int64_t in_a, int64_t out_a, double scale_a, void *table_a_dev, size_t bytes_a,
                    int64_t in_b, int64_t out_b, double scale_b, void *table_b_dev, size_t bytes_b, aa_stream_t stream);
/* aa_table_query for the two tables of a call (H and W axis) with ONE synchronisation: a shape never seen before costs two table builds, and a
 * data pipeline of random crops meets a new shape on every call. */
int aa_table_query2(const void *table_a_dev, const void *table_b_dev, aa_table_header *host_a, aa_table_header *host_b, aa_stream_t stream);

/* Workspace (bytes) the forward needs for this problem; 0 when a fused single-launch path applies.  The answer depends on the
 * shape and the tables only, never on the pointers: which kernel runs is decided from the same facts, and a uint8 view that starts
 * on an odd byte is served by the same kernels (byte stores instead of dword stores).  Tensors of 2 / 4 / 8-byte elements must
 * start on an element boundary (AA_ERR_BAD_SHAPE otherwise). */
size_t aa_workspace_bytes(int dtype, int layout, int64_t N, int64_t C, int64_t H, int64_t W, int64_t oH, int64_t oW,
                          const aa_axis *ax_h, const aa_axis *ax_w);

/* Forward: replaces ti_upsample_{bilinear,bicubic,nearest}2d_cpu + the separable driver + both passes
 * (s2.2/aa_interpolation_impl.h:731-807, :628-683, :536-625, :131-187, :29-120).
 *   in_dev  [N,C,H,W]  dtype/layout as given          out_dev [N,C,oH,oW] same dtype/layout
 *   ax_w: table for W -> oW (first pass, like the reference and PIL); ax_h: H -> oH (second pass).
 * Table kind selects the arithmetic: F32/F64 tables with f32/f64 images (bit-comparable with the reference's
 * CPU path: separately rounded product and sum, taps in order); u8 images with AA_TABLE_PIL tables give
 * Pillow's integer result; u8 images with AA_TABLE_F32 tables give the reference harness semantics
 * (test.py:52-58,72,75: float(), fp32 op, bicubic clamp, truncating byte()).
 * Empty batch (N==0) is allowed (s2.2:747-750).  Asynchronous on `stream`. */
int aa_resample_fwd(const void *in_dev, void *out_dev, void *workspace_dev, size_t workspace_bytes, int dtype,
                    int layout, int64_t N, int64_t C, int64_t H, int64_t W, const aa_axis *ax_h, const aa_axis *ax_w,
                    aa_stream_t stream);

/* The same with flags.  AA_FLAG_FAST — the opt-in TOLERANCE mode for f32 / f16 / bf16 images: the caller accepts results within
 * 1e-4 relative of the reference's (BASELINE.json's float bar) instead of bit-identical ones, and the fused kernels then accumulate
 * with fused multiply-adds over zero-padded windows (no separately rounded product and sum, no per-position select).  Differences
 * are rounding only (~1e-7 relative); a NaN / Inf input value additionally poisons every output whose 16-byte-aligned window
 * holds it (0 * inf), not only those whose taps do.  u8 images with AA_TABLE_F32 tables (the harness's float arithmetic) have the
 * mode too: FMAs in both passes, so the truncated byte may differ from the exact mode's by one count.  Ignored for Pillow's integer
 * arithmetic, for f64 images and wherever no tolerance kernel applies (the exact kernels run: bit-identical results are always
 * within tolerance).  aa_workspace_bytes() answers for both. */
#define AA_FLAG_FAST 1u
int aa_resample_fwd_ex(const void *in_dev, void *out_dev, void *workspace_dev, size_t workspace_bytes, int dtype,
                       int layout, int64_t N, int64_t C, int64_t H, int64_t W, const aa_axis *ax_h, const aa_axis *ax_w,
                       unsigned flags, aa_stream_t stream);

/* Forward over a STRIDED VIEW of a larger tensor, without a copy: the reference walks arbitrary strides through TensorIterator
 * (s2.2/aa_interpolation_impl.h:555-559), and the views a data pipeline produces — a crop x[:, :, y0:y1, x0:x1] (RandomResizedCrop),
 * a batch slice — are read here where they lie.  in_strides = the view's strides in ELEMENTS for (N, C, H, W).  Served: rows of
 * consecutive elements (AA_NCHW: stride_W = 1; AA_NHWC: stride_C = 1, stride_W = C), any row pitch, planes n * C + c uniformly spaced
 * (AA_NCHW: stride_N = C * stride_C; AA_NHWC: any stride_N), and a shape one of the fused single-launch kernels takes (no workspace).
 * Anything else returns AA_ERR_STRIDES: make a dense copy and call aa_resample_fwd.  The output is dense, in `layout`. */
int aa_resample_fwd_strided(const void *in_dev, void *out_dev, int dtype, int layout, int64_t N, int64_t C, int64_t H, int64_t W,
                            const int64_t *in_strides, const aa_axis *ax_h, const aa_axis *ax_w, unsigned flags, aa_stream_t stream);

/* Decode-adjacent forward: uint8 image in, float32 tensor out, ONE launch.  Replaces what the reference's harness
 * does around the op on the CPU — np.asarray(pil) -> transpose(2,0,1) -> .float() -> op (test.py:337-339,55; README.md:416
 * prices those conversions at 0.33 of 2.27 ms) — and, optionally, the per-channel normalisation that follows in a data
 * loader.  out = op(float(in)) in the reference's fp32 arithmetic (AA_TABLE_F32 tables; bit-identical to aa_resample_fwd on
 * the converted tensor), written in cv->out_layout, which may differ from the input's; with cv->normalize,
 * out = (out - mean[c]) / std[c] in fp32 (C <= 4).  in_dev [N,C,H,W] uint8 in `layout`; out_dev [N,C,oH,oW] float32. */
typedef struct aa_convert {
  int32_t out_layout; /* aa_layout of the float32 output */
  int32_t normalize;  /* 0: raw op result */
  float mean[4];
  float std[4];
  uint32_t flags;     /* 0, or AA_FLAG_FAST: the tolerance mode (fused multiply-adds; results within 1e-4 relative of the exact mode's) */
} aa_convert;
size_t aa_workspace_bytes_u8_to_f32(int layout, int64_t N, int64_t C, int64_t H, int64_t W, const aa_axis *ax_h, const aa_axis *ax_w,
                                    const aa_convert *cv);
int aa_resample_fwd_u8_to_f32(const void *in_dev, void *out_dev, void *workspace_dev, size_t workspace_bytes, int layout,
                              int64_t N, int64_t C, int64_t H, int64_t W, const aa_axis *ax_h, const aa_axis *ax_w,
                              const aa_convert *cv, aa_stream_t stream);

/* Backward (true adjoint): grad_in[N,C,H,W] = H^T V^T grad_out[N,C,oH,oW], F32/F64 only.  Replaces
 * ti_upsample_bilinear2d_backward_cpu (s2.2/aa_interpolation_backward_impl.h:185-219) in API shape.
 * tr_h / tr_w are TRANSPOSED tables from aa_table_transpose (gather form: deterministic, no atomics, no zero fill). */
int aa_resample_bwd(const void *grad_out_dev, void *grad_in_dev, void *workspace_dev, size_t workspace_bytes, int dtype,
                    int layout, int64_t N, int64_t C, int64_t H, int64_t W, const aa_axis *tr_h, const aa_axis *tr_w,
                    aa_stream_t stream);
/* Scatter form of the same adjoint with fp32/fp64 atomics straight from the FORWARD tables (BASELINE config 5
 * names it); grad_in is zero-filled first by the call.  Results agree with aa_resample_bwd to rounding only. */
int aa_resample_bwd_atomic(const void *grad_out_dev, void *grad_in_dev, void *workspace_dev, size_t workspace_bytes,
                           int dtype, int layout, int64_t N, int64_t C, int64_t H, int64_t W, const aa_axis *ax_h,
                           const aa_axis *ax_w, aa_stream_t stream);
size_t aa_workspace_bytes_bwd(int dtype, int layout, int64_t N, int64_t C, int64_t H, int64_t W, int64_t oH, int64_t oW);

/* One separable pass along one axis of a dense array viewed as [outer][in_size][inner] -> [outer][out_size][inner]
 * (the body the reference instantiates per dimension, `_ti_separable_upsample_generic_Nd_kernel_impl_single_dim`
 * s2.2/aa_interpolation_impl.h:536-625, "NCHW, NCL or NCKHW" :545).  The N-d front-ends (1-D NCL, 3-D NCDHW) are this
 * call once per resampled axis, last axis first like the reference (:658); no workspace.  dtype/table-kind pairs as
 * for aa_resample_fwd except u8 with AA_TABLE_F32 (the harness mode needs a float intermediate). */
int aa_resample_axis_fwd(const void *in_dev, void *out_dev, int dtype, int64_t outer, int64_t in_size, int64_t inner,
                         const aa_axis *ax, aa_stream_t stream);

/* Device-to-device copy of `bytes` bytes with 16-byte vector loads/stores, enqueued on `stream`: the probe bench.py times
 * on the box to report the attainable HBM copy ceiling next to the 8 TB/s spec peak.  form 0: one element per
 * thread; 1: grid-stride; 2: four elements per thread, loads in flight before the stores; 3: form 2, streaming (nt) policy;
 * 4: write only (fills dst); 5: read only (sums src). */
int aa_probe_copy(const void *src_dev, void *dst_dev, size_t bytes, int form, aa_stream_t stream);

/* Kernel selection, process-wide; returns the previous setting.  1 (default): fused single-launch kernels, newest
 * design first; 2: first-generation fused kernels only (A/B measurements); 0: none.
 * With 0 every call takes the generic two-launch path — used by tests to cross-check the fused kernels against an
 * independent implementation at sizes the CPU oracle cannot reach, and by bench.py for A/B numbers. */
int aa_set_fused(int enabled);

/* Store form of the up-scaling / backward kernel, process-wide; returns the previous setting.  -1 (default): chosen from the output
 * size (outputs beyond 64 MiB are stored with the streaming policy, in one of three forms picked from the row pitch); 1: always the
 * streaming forms; 0: never.  A test hook: it lets the parity tests reach the streaming forms at sizes the CPU oracle can check.
 * The library reads NO environment variables (developer builds with -DAA_V2_TUNING do, for experiments). */
int aa_set_store_form(int form);

/* Plane groups of the fused uint8 kernel, process-wide; returns the previous setting.  1 (default): planar (NCHW) uint8 images of
 * three channels run the three planes of an image in one wave (Pillow arithmetic, shrinking heights); 0: one wave per plane, as every
 * other planar shape does.  Same results either way — a test hook and the A/B switch of the measurements in DESIGN.md. */
int aa_set_plane_groups(int enabled);

/* Name of the kernel variant the last aa_resample_fwd on this thread dispatched to (for tests/bench). */
const char *aa_last_variant(void);

#ifdef __cplusplus
}
#endif
#endif /* AA_INTERP_H */
