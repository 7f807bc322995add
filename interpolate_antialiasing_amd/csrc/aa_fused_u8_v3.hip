// aa_fused_u8_v3.hip — host-side dispatcher of the fused uint8 kernel; the kernel itself, its design notes and its
// launch chain are in aa_fused_u8_v3_impl.h, instantiated per channel count in aa_fused_u8_v3_c{1,3,4}.hip.

#include "aa_fused_u8_v3_impl.h"

// Everything the kernel needs that can be known without the pointers (aa_workspace_bytes asks before they exist).
static bool v3_shape_ok(int dtype, int layout, int64_t N, int64_t Cin, int64_t H, int64_t W, const aa_axis &ah, const aa_axis &aw, bool *flt_out,
                        bool *planar_out, int *tw_out, int out_f32 = 0, int out_layout = AA_NCHW, bool *up_out = nullptr,
                        int *cap_out = nullptr) {
  if (dtype != AA_U8) return false;
  if (out_f32 && (ah.kind != AA_TABLE_F32 || aw.kind != AA_TABLE_F32)) return false;  // float output = float arithmetic
  const bool flt = ah.kind == AA_TABLE_F32 && aw.kind == AA_TABLE_F32;  // the reference harness's uint8 semantics
  if (!flt && (ah.kind != AA_TABLE_PIL || aw.kind != AA_TABLE_PIL)) return false;
  // channels_last with 3 or 4 interleaved channels, or planar bytes: NCHW is N*C single-channel images
  const bool planar = layout == AA_NCHW || Cin == 1;
  const int C = planar ? 1 : (int)Cin;
  if (C != 1 && C != 3 && C != 4) return false;
  // a planar wave holds one channel: it can write its own float plane, not an interleaved pixel
  if (out_f32 && planar && Cin != 1 && out_layout != AA_NCHW) return false;
  const int64_t oH = ah.out_size, oW = aw.out_size;
  if (out_f32 && (uint64_t)oH * oW * (planar ? 1 : Cin) * 4 > 0xFFFFFFF0ull) return false;
  const bool up = H < oH;  // growing heights: the vertical pass gathers (template parameter UPK of the kernel)
  if (up) {  // needs the H table's gather records (6 weights each)
    const int taps_h = ah.max_taps > 0 ? ah.max_taps : ah.ksize;
    if (ah.gather_off <= 0 || taps_h > 6) return false;
    // planar bytes store 64-byte pieces per strip and output row: with many strips the generic two-launch path is faster
    // (measured, [128,3,438,906] -> 1200x1200: fused 1.14 ms, generic 0.64 ms; -> 120 columns: fused 0.123, generic 0.148)
    if (planar && oW > 256) return false;
  } else {  // the in-register scatter pass needs the H table's scatter section and at most 4 open output rows
    if (ah.scatter_off <= 0 || ah.scatter_max <= 0 || ah.scatter_max > 6) return false;
    if (ah.scatter_max > 4 && (aw.max_taps > 0 ? aw.max_taps : aw.ksize) <= 16) return false;  // (6 open rows: the wide-window instantiations only)
  }
  const int taps_w = aw.max_taps > 0 ? aw.max_taps : aw.ksize;
  int tw = round_tw(taps_w);
  if (flt && tw != 0 && tw < 6) tw = 6;  // the float variant is instantiated for windows of 6, 8, 12 and 16 taps
  // windows of 17 .. 34 taps: Pillow arithmetic, shrinking heights.  (With growing heights — test.py's (120, 1200) — the gather form with
  // such windows was built and measured SLOWER than the two-launch path: bicubic channels_last 0.226 vs 0.205 ms per 128 images.)
  if (tw > 16 && up) return false;
  // windows of 35 .. 136 taps (down-scaling by 17 .. 68 bilinear, 9 .. 34 bicubic): SPLIT windows — four lanes share an output pixel, each
  // holds a quarter of its window (tw = taps per lane: 16 / 24 / 34) and the partial sums meet in two DPP additions; Pillow arithmetic
  // (integer sums are associative: bit-exact), shrinking heights, uint8 out; strips of 16 columns
  const bool split = tw == 0 && taps_w <= 136 && !flt && !up && !out_f32;
  if (split) tw = taps_w <= 64 ? 16 : (taps_w <= 96 ? 24 : 34);
  if (tw == 0 || W < (split ? 4 * tw : tw)) return false;
  if ((uint64_t)H * W * C > 0x7FFFFFF0ull || (uint64_t)oH * oW * C > 0xFFFFFFF0ull) return false;
  int span_px = split ? aa_strip_span_px16(aw, 4 * tw) : aa_strip_span_px(aw, tw);
  if (span_px < 0) return false;
  int nseg = (span_px * C + 3 + 15 + 15) / 16;
  int cap = split ? 16 : 64;  // output columns per strip
  if (up && nseg > 64) {
    // the gather form is instantiated with one staging DMA per row (64 pieces): strong down-scaling in W (test.py's 906 -> 120
    // with growing heights) gets strips of 32 columns — half the lanes idle, but such a shape is bound by its input stream
    span_px = aa_strip_span_px32(aw, tw);
    if (span_px < 0) return false;
    nseg = (span_px * C + 3 + 15 + 15) / 16;
    cap = 32;
    if (nseg > 64) return false;
    // ... unless the first-generation kernel (Pillow arithmetic, channels_last, uint8 out) takes the shape: its block-wide tiles
    // handle these wide windows better (measured, [128,3,438,906] -> 1200 x 120: 0.060 ms against 0.095 ms here)
    if (!flt && !planar && !out_f32 && aa_fused_u8_nhwc_applicable(dtype, layout, N, Cin, H, W, &ah, &aw)) return false;
  }
  if (nseg > 128 || (size_t)aa_v3_group() * nseg * 16 > 64 * 1024) return false;
  const int64_t nstrips = (oW + cap - 1) / cap + 1;  // (balanced strips can be one more)
  if (!aa_grid_fits((planar ? N * Cin : N) * nstrips)) return false;
  *flt_out = flt; *planar_out = planar; *tw_out = tw;
  if (up_out) *up_out = up;
  if (cap_out) *cap_out = cap;
  return true;
}

bool aa_fused_u8_v3_applicable(int dtype, int layout, int64_t N, int64_t C, int64_t H, int64_t W, const aa_axis *ah, const aa_axis *aw,
                               int out_f32, int out_layout) {
  bool flt, planar;
  int tw;
  return ah && aw && v3_shape_ok(dtype, layout, N, C, H, W, *ah, *aw, &flt, &planar, &tw, out_f32, out_layout);
}

int aa_try_fused_u8_nhwc_v3(const AAProblem &q, const char **variant) {
  bool flt, planar, up = false;
  int tw, cap = 64;
  if (!v3_shape_ok(q.dtype, q.layout, q.N, q.C, q.H, q.W, q.ah, q.aw, &flt, &planar, &tw, q.out_f32, q.out_layout, &up, &cap)) return 0;
  const int C = planar ? 1 : (int)q.C;
  const int64_t NI = planar ? q.N * q.C : q.N;  // images the kernel sees
  const int G = aa_v3_group();

  FusedU8V3Params p;
  p.H = (int)q.H; p.W = (int)q.W; p.oH = (int)q.oH; p.oW = (int)q.oW;
  p.ksize_w = q.aw.ksize; p.ksize_h = q.ah.ksize;
  // a pitched view (cropped / batch-sliced tensor): rows q.in_row_pitch bytes apart, images (planes) q.in_img_pitch bytes apart
  p.row_pitch = q.in_row_pitch ? (unsigned)q.in_row_pitch : (unsigned)(q.W * C);
  p.img_in_bytes = q.in_img_pitch ? (unsigned long long)q.in_img_pitch : (unsigned long long)q.H * q.W * C;
  if (q.in_row_pitch && ((uint64_t)q.H * (uint64_t)q.in_row_pitch > 0x7FFFFFF0ull || up)) return 0;  // (32-bit offsets inside an image; growing heights: dense only)
  p.img_out_bytes = (unsigned long long)q.oH * q.oW * C * (q.out_f32 ? 4 : 1);
  p.outm = q.out_f32 ? (planar || q.out_layout == AA_NCHW ? 1 : 2) : 0;
  p.normalize = q.out_f32 ? q.normalize : 0;
  p.cin = (int)q.C;
  p.fast = 0;
  p.byte_store = (!q.out_f32 && ((q.oW * C) % 4 != 0 || (C == 3 && q.oW % 4 != 0) || ((uintptr_t)q.out & 3) != 0)) ? 1 : 0;
  if (q.out_f32 && ((uintptr_t)q.out & 3) != 0) return AA_ERR_BAD_SHAPE;  // a float tensor that is not float aligned
  for (int c = 0; c < 4; c++) { p.mean[c] = q.mean[c]; p.std[c] = q.std[c]; }
  p.in_mis = (int)((uintptr_t)q.in & 15);
  p.total_in_bytes = q.in_row_pitch ? p.img_in_bytes * (unsigned long long)(NI - 1) + (unsigned long long)(q.H - 1) * p.row_pitch + (unsigned long long)q.W * C + p.in_mis
                                    : p.img_in_bytes * (unsigned long long)NI + (unsigned long long)p.in_mis;
  p.total_out_bytes = p.img_out_bytes * (unsigned long long)NI;
  p.n_images = NI;
  p.sc_off = q.ah.scatter_off;
  p.gather_off = q.ah.gather_off;
  p.nstrips = (int)((q.oW + cap - 1) / cap);
  p.strip_w = (int)(((q.oW + p.nstrips - 1) / p.nstrips + 3) & ~3);  // balanced strips (196 -> 4 x 52, not 3 x 64 + 4)
  p.nstrips = (int)((q.oW + p.strip_w - 1) / p.strip_w);
  // all strips of a band in one workgroup when they fit (<= 8 waves); wider images: groups of 4 strips
  p.strips_per_block = p.nstrips <= 8 ? p.nstrips : 4;
  p.spb_forced = 0;
  if (const char *e = aa_knob("AA_V3_SPB")) {  // experiment knob
    const int v = atoi(e);
    if (v >= 1 && v <= 8) { p.strips_per_block = v; p.spb_forced = 1; }
  }

  // segment: bytes covered by 64 consecutive windows of one input row (+ alignment slack), see aa_fused_u8_v2.hip
  const bool split = cap == 16;  // (v3_shape_ok: four lanes per output pixel, tw taps each)
  const int span_px = split ? aa_strip_span_px16(q.aw, 4 * tw) : (cap == 32 ? aa_strip_span_px32(q.aw, tw) : aa_strip_span_px(q.aw, tw));
  p.nseg = (span_px * C + 3 + 15 + 15) / 16;
  if (p.nseg > 128) return 0;
  p.seg_bytes = p.nseg * 16;
  const size_t lds = (size_t)G * p.seg_bytes;
  if (lds > 64 * 1024) return 0;

  p.ybands = 1;
  p.plane_in_bytes = p.plane_out_bytes = 0;
  p.pl_planes = 0;

  // Plane groups (template parameter PL of the kernel): planar bytes, either arithmetic, shrinking heights — one wave filters the same strip
  // and band of THREE CONSECUTIVE PLANES of the tensor (the channels of an RGB image; three grayscale images; planes of neighbouring
  // images when C is 2, 4, 5, ...: planes are independent and uniformly spaced), sharing each row's staging DMA and fixed work.  The
  // single-plane form keeps: growing heights; windows beyond 12 taps (8 in float arithmetic: the wider instantiations need 133-147 VGPRs
  // = 3 waves per SIMD); segments beyond 16 pieces (down-scaling by 4 and more: the single planes' staging DMAs are full enough as they
  // are, measured +4 % at 1024 -> 224).
  if (planar && NI >= 2 && !up && tw <= (flt ? 8 : 12) && tw >= 4 && p.nseg <= 16 && G == 8 && q.ah.scatter_max <= 4 && (!q.out_f32 || p.outm == 1) &&
      3 * p.img_in_bytes <= 0x7FFFFFF0ull && 3 * p.img_out_bytes <= 0x7FFFFFF0ull && g_aa_plane_groups != 0) {
    FusedU8V3Params pg = p;
    pg.plane_in_bytes = p.img_in_bytes;    // (the single-plane form's "images" are the planes)
    pg.plane_out_bytes = p.img_out_bytes;
    pg.img_in_bytes = 3 * p.img_in_bytes;
    pg.img_out_bytes = 3 * p.img_out_bytes;
    pg.n_images = (NI + 2) / 3;  // groups of three consecutive planes (the last one may hold one or two)
    pg.pl_planes = NI;
    const bool fastg = flt && q.fast;
    pg.fast = fastg ? 1 : 0;
    const size_t lds_g = (size_t)G * 1024;  // (a 1-KiB stage slot per row: the kernel's fixed layout)
    const int rcg = !flt ? aa_v3_launch_c3g(tw, q.ah.scatter_max, pg, q, lds_g)
                         : (fastg ? aa_v3_launch_c3gff(tw, q.ah.scatter_max, pg, q, lds_g) : aa_v3_launch_c3gf(tw, q.ah.scatter_max, pg, q, lds_g));
    if (rcg != 0) {
      if (rcg == 1)
        *variant = !flt ? "fused_u8_planar_pil_v3"
                        : (q.out_f32 ? (fastg ? "fused_u8_planar_to_f32_v3_fast" : "fused_u8_planar_to_f32_v3")
                                     : (fastg ? "fused_u8_planar_harness_v3_fast" : "fused_u8_planar_harness_v3"));
      return rcg;
    }
  }

  int rc;
  if (split) {
    p.byte_store = 1;  // (a quad's first lane stores its pixel's bytes)
    rc = C == 3 ? aa_v3_launch_c3s(tw, q.ah.scatter_max, p, q, lds) : C == 4 ? aa_v3_launch_c4s(tw, q.ah.scatter_max, p, q, lds)
                                                                             : aa_v3_launch_c1s(tw, q.ah.scatter_max, p, q, lds);
  } else if (up) {
    const int taps_h = q.ah.max_taps > 0 ? q.ah.max_taps : q.ah.ksize;
    const bool nonneg = q.aw.filter != AA_FILTER_CUBIC && q.ah.filter != AA_FILTER_CUBIC;
    rc = C == 3   ? aa_v3_launch_up_c3(tw, taps_h, nonneg, flt, p, q, lds)
         : C == 4 ? aa_v3_launch_up_c4(tw, taps_h, nonneg, flt, p, q, lds)
                  : aa_v3_launch_up_c1(tw, taps_h, nonneg, flt, p, q, lds);
  } else {
    if (flt && q.fast && tw <= 16) {  // the tolerance mode of the float-arithmetic kernels (down-scaling heights only; growing heights run exact)
      p.fast = 1;
      rc = C == 3 ? aa_v3_launch_c3ff(tw, q.ah.scatter_max, p, q, lds) : C == 4 ? aa_v3_launch_c4ff(tw, q.ah.scatter_max, p, q, lds)
                                                                                 : aa_v3_launch_c1ff(tw, q.ah.scatter_max, p, q, lds);
    } else if (tw > 16 && flt) {  // (wide windows in float arithmetic run exact in either precision mode)
      rc = C == 3 ? aa_v3_launch_c3wf(tw, q.ah.scatter_max, p, q, lds) : C == 4 ? aa_v3_launch_c4wf(tw, q.ah.scatter_max, p, q, lds)
                                                                                : aa_v3_launch_c1wf(tw, q.ah.scatter_max, p, q, lds);
    } else if (tw > 16) {
      rc = C == 3 ? aa_v3_launch_c3w(tw, q.ah.scatter_max, p, q, lds) : C == 4 ? aa_v3_launch_c4w(tw, q.ah.scatter_max, p, q, lds)
                                                                               : aa_v3_launch_c1w(tw, q.ah.scatter_max, p, q, lds);
    } else
    rc = C == 3   ? aa_v3_launch_c3(tw, q.ah.scatter_max, flt, p, q, lds)
         : C == 4 ? aa_v3_launch_c4(tw, q.ah.scatter_max, flt, p, q, lds)
                  : aa_v3_launch_c1(tw, q.ah.scatter_max, flt, p, q, lds);
  }
  const bool fastv = flt && q.fast && !up && tw <= 16;
  if (rc == 1 && q.out_f32) *variant = planar ? (fastv ? "fused_u8_planar_to_f32_v3_fast" : "fused_u8_planar_to_f32_v3")
                                              : (p.outm == 1 ? (fastv ? "fused_u8_nhwc_to_f32_nchw_v3_fast" : "fused_u8_nhwc_to_f32_nchw_v3")
                                                             : (fastv ? "fused_u8_nhwc_to_f32_nhwc_v3_fast" : "fused_u8_nhwc_to_f32_nhwc_v3"));
  else if (rc == 1) *variant = flt ? (planar ? (fastv ? "fused_u8_planar_harness_v3_fast" : "fused_u8_planar_harness_v3")
                                             : (fastv ? "fused_u8_nhwc_harness_v3_fast" : "fused_u8_nhwc_harness_v3"))
                        : (planar ? "fused_u8_planar_pil_v3" : "fused_u8_nhwc_pil_v3");
  return rc;
}
