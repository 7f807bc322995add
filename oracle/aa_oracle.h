/*
 * aa_oracle.h — CPU ORACLE for the antialiased separable resample hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is shipped, linked or called by the product
 * (interpolate_antialiasing_amd/): only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library, and there only as the checker.
 *
 * It is a plain-C restatement (no ATen, no TensorIterator) of the reference algorithm; every
 * function cites the reference file:line it follows ("s2.2" = step_two_dot_two, paths relative to
 * the reference checkout).  Parity is PINNED (tests/test_oracle_golden.py):
 *   - bit-for-bit against the reference's own C++ compiled here (oracle/_ref, see oracle/Makefile)
 *     on weight tables and fp32/fp64 forward outputs (tests/golden/ref_*.npz);
 *   - against the reference's committed known-answer PNG data/proto_aa_interp_lin_step_one_output.png;
 *   - the uint8 (Pillow-semantics) path bit-for-bit against Pillow 12.2.0 outputs (tests/golden/pil_*.npz).
 */
#ifndef AA_ORACLE_H
#define AA_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* filter ids shared with include/aa_interp.h */
enum { AAO_FILTER_LINEAR = 0, AAO_FILTER_CUBIC = 1, AAO_FILTER_BOX = 2 };

/* s2.2/aa_interpolation_impl.h:287,333,377 (interp_size) and :208-210 (ksize) — returns ksize for
 * scalar_t=float (is_double=0) or double (is_double=1) arithmetic; scale_opt<=0 means "not given". */
int aao_ksize(int filter, int64_t in_size, int64_t out_size, int align_corners, double scale_opt, int is_double);

/* s2.2/aa_interpolation_impl.h:195-281 with scalar_t=float.  xmin/xsize: [out_size]; w: [out_size*ksize]. */
int aao_weights_f32(int filter, int64_t in_size, int64_t out_size, int align_corners, double scale_opt,
                    int64_t *xmin, int64_t *xsize, float *w);
/* same with scalar_t=double */
int aao_weights_f64(int filter, int64_t in_size, int64_t out_size, int align_corners, double scale_opt,
                    int64_t *xmin, int64_t *xsize, double *w);

/* Forward 2-D separable AA resample, W pass then H pass (s2.2:628-683, inner loops :29-87).
 * Strides are in ELEMENTS for (N,C,H,W); the temp is contiguous [N,C,H,oW] like the reference (:660).
 * nthreads<=1: serial. Returns 0 or a negative error. */
int aao_forward_f32(int filter, const float *in, float *out, int64_t N, int64_t C, int64_t H, int64_t W,
                    int64_t oH, int64_t oW, const int64_t in_strides[4], const int64_t out_strides[4],
                    int align_corners, int nthreads);
int aao_forward_f64(int filter, const double *in, double *out, int64_t N, int64_t C, int64_t H, int64_t W,
                    int64_t oH, int64_t oW, const int64_t in_strides[4], const int64_t out_strides[4],
                    int align_corners, int nthreads);

/* TRUE adjoint of aao_forward_* (what test.py:387-398 asks for): grad_in = H^T V^T grad_out,
 * built from the same tables.  Contiguous NCHW only. */
int aao_backward_f32(int filter, const float *grad_out, float *grad_in, int64_t N, int64_t C, int64_t H, int64_t W,
                     int64_t oH, int64_t oW, int align_corners);
int aao_backward_f64(int filter, const double *grad_out, double *grad_in, int64_t N, int64_t C, int64_t H, int64_t W,
                     int64_t oH, int64_t oW, int align_corners);

/* The backward AS WRITTEN in the reference: stock non-AA 2x2-tap bilinear scatter
 * (s2.2/aa_interpolation_backward_impl.h:80-108 with ATen compute_source_index_and_lambda).
 * Kept only to pin "the header's backward is not the AA adjoint" (label: legacy, do not match). */
int aao_legacy_nonaa_linear_backward_f32(const float *grad_out, float *grad_in, int64_t N, int64_t C, int64_t H,
                                         int64_t W, int64_t oH, int64_t oW, int align_corners);

/* ---- uint8, Pillow semantics (SURVEY §8 a-U; algorithm cited by URL at reference README.md:18,40 and
 * s2.2/aa_interpolation_impl.h:289-291,364-366,407-409: Pillow src/libImaging/Resample.c).
 * Coefficients in double, 22-bit fixed point, accumulate from 1<<21, clip8, uint8 intermediate. */
int aao_pil_ksize(int filter, int64_t in_size, int64_t out_size);
int aao_pil_coeffs(int filter, int64_t in_size, int64_t out_size, int32_t *xmin, int32_t *xsize, int32_t *kk /*[out*ksize]*/,
                   double *prekk /*[out*ksize] or NULL*/);
/* in: [N,H,W,C] uint8 dense (channels_last storage of an NCHW tensor); out: [N,oH,oW,C]. */
int aao_pil_resize_u8_nhwc(int filter, const uint8_t *in, uint8_t *out, int64_t N, int64_t H, int64_t W, int64_t C,
                           int64_t oH, int64_t oW, int nthreads);
/* generic-stride variant (element strides for N,C,H,W) */
int aao_pil_resize_u8(int filter, const uint8_t *in, uint8_t *out, int64_t N, int64_t C, int64_t H, int64_t W,
                      int64_t oH, int64_t oW, const int64_t in_strides[4], const int64_t out_strides[4], int nthreads);

/* uint8 through the reference harness semantics (test.py:52-58,75): float() -> fp32 op -> truncating byte()
 * (bicubic: clamp to [0,255] first, test.py:72). */
int aao_harness_u8(int filter, const uint8_t *in, uint8_t *out, int64_t N, int64_t C, int64_t H, int64_t W,
                   int64_t oH, int64_t oW, const int64_t in_strides[4], const int64_t out_strides[4], int nthreads);

int aao_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
