// aa_fused_u8_mfma_impl.h — fused resample for uint8 channels_last, Pillow arithmetic, BOTH PASSES ON THE MATRIX PIPE.
//
// Fourth design of the uint8 kernel.  The third (aa_fused_u8_v3_impl.h) is bound by vector-ALU issue (83 % busy on the headline
// shape): Pillow's 22-bit integer weights make every multiply-accumulate a 4-cycle v_mul_i32_i24_sdwa plus half a v_add3.  Here
// both passes are banded integer matrix products on v_mfma_i32_16x16x64_i8 (measured first in
// tools/microbench/ubench_mfma_i8_band.hip, profiles/r03_ubench_mfma_band.txt):
//
//   horizontal   D[16 rows][16 elements] = A[16 rows][64 bytes] x Bh[64 bytes][16 elements]
//                A = 64 contiguous interleaved input bytes of 16 consecutive input rows (one ds_read_b128 per lane from an LDS
//                image of rows an odd number of 16-byte chunks apart), biased to signed (v_xor 0x80808080); Bh = the band of the W table for one
//                TILE of NPX output pixels (C * NPX <= 16 elements; 4 pixels x 3 channels), split into three signed byte digit
//                planes (w = d0 + 256 d1 + 65536 d2): three MFMAs, planes recombined Horner style with the constant
//                128 * sum(w) + 2^21 (- 2^29) entering through the C operand, clipped and packed by v_ashr_pk_i8_i32: the lane
//                then holds Pillow's uint8 intermediate (minus 128) of 4 consecutive ROWS of one element in one dword —
//   vertical     which is exactly the A operand layout of the second product, D[16 elements][16 output rows] =
//                A[16 elements][64 input rows] x Bv[64 input rows][16 output rows]: four such dwords (64 input rows = one
//                SUPER-BLOCK) are the operand, no lane ever moves data.  Bv = the band of the H table for 16 output rows,
//                three digit planes, one or two super-blocks deep.  The result lane holds 4 consecutive output bytes of one
//                output row: one dword store.
// Integer arithmetic is associative and nothing is rounded before Pillow rounds: results are bit-identical to PIL.
//
// Data movement: a workgroup of 4 waves owns one strip of <= 16 tiles (64 output columns) of one image, top to bottom.  Input rows
// are staged 16 at a time ("block") into a ring of R blocks shared by the 4 waves with LDS-DMA (buffer_load_dwordx4 ... lds; a
// DMA instruction = one row segment, 16 bytes per lane; LDS-DMA serves byte-unaligned global sources, so every row sits at the
// same phase; the vector cache looks up ~0.34 lines per clock and CU whatever they hold, so a segment must be ONE contiguous
// piece: 10 lines for 544 useful bytes, where 64-byte pieces of 16 rows touched 18), one workgroup barrier per block.  Each wave keeps the band operands of ITS 4 tiles in registers (48 VGPRs) and runs both passes
// for them.  Waves 0-2 issue the row DMAs (a deep in-order queue: R - 1 blocks in flight — the LDS ring IS the memory pipeline,
// measured: ~100 KiB must be in flight per CU); wave 3 issues the DMAs that stage the vertical band operands of the next
// output-row tiles into a double buffer (a shallow queue, so they land in time; in one in-order vmcnt stream with the rows they
// would queue behind R - 1 blocks).  Finished output-row tiles are transposed through LDS and leave as whole 16-byte pieces of
// 4 rows x 192 bytes per wave.
//
// Everything table-derived the kernel needs (band operands, constants, window offsets) is a PLAN built once per
// (W table, H table, C, shape) on device by the kernels at the end of this file: see aa_interp.h (aa_plan_*).
#pragma once
#include <stdint.h>

#include <hip/hip_runtime.h>

// developer ablations (wrong results!): 1 no stores, 2 no vertical pass, 3 DMA + barriers only, 4 no row DMA,
// 5 = 3 without Bv staging and stores, 6 = 5 without barriers, 7 = 5 with every row DMA reading the image's first rows (cache hits)
#ifndef AA_MFMA_ABL
#define AA_MFMA_ABL 0
#endif
#ifndef AA_MFMA_NOSYNC
#define AA_MFMA_NOSYNC 0  // developer knob (wrong results!): 1 = no barriers and no vmcnt waits in the block loop
#endif
#ifndef AA_MFMA_LOADERS
#define AA_MFMA_LOADERS 3  // developer knob: waves 0 .. LOADERS-1 issue the row DMAs
#endif
#ifndef AA_MFMA_DMA_MASK
#define AA_MFMA_DMA_MASK 0  // developer knob (wrong results!): low bits cleared from every row DMA's source address
#endif
#if AA_MFMA_ABL == 5
#define AA_MFMA_SKEL 5
#elif AA_MFMA_ABL == 6
#define AA_MFMA_SKEL 6
#elif AA_MFMA_ABL == 7
#define AA_MFMA_SKEL 7
#else
#define AA_MFMA_SKEL 0
#endif
#if AA_MFMA_ABL >= 5
#undef AA_MFMA_ABL
#define AA_MFMA_ABL 3
#endif

#define AA_PLAN_MAGIC 0x4E4C5041  // 'APLN'

// ---- plan layout (one flat device buffer) -------------------------------------------------------------------------
struct aa_plan_header {  // 128 bytes
  int32_t magic;
  int32_t C;                // interleaved channels
  int32_t npx;              // output pixels per tile (C * npx <= 16)
  int32_t ntiles;           // ceil(oW / npx)
  int32_t tiles_per_strip;  // 16: a workgroup's strip (4 waves x 4 tiles)
  int32_t nstrips;
  int32_t nch;              // 16-byte chunks staged per row segment = row pitch of the LDS image / 16 (odd)
  int32_t njt;              // output-row tiles: ceil(oH / 16)
  int32_t nsb;              // super-blocks of 64 input rows: ceil(H / 64)
  int32_t H, W, oH, oW;
  int32_t off_strip, off_tile, off_bh, off_ch, off_jt, off_wv, off_cv;  // byte offsets of the sections
  int32_t fits;             // 1: every tile window fits its 64 slots and every output-row tile spans <= 2 super-blocks;
                            // cleared by the build kernels otherwise (the dispatcher then declines)
  int32_t max_jt_per_blk;   // most output-row tiles whose window ends in the same 16-row block (the kernel takes <= 2)
  int32_t total_bytes;
  int32_t reserved[9];
};
// sections:
//   strip : int32 seg_first[nstrips]          row-relative byte that LDS byte 0 of a staged row holds
//   tile  : int32 aoff[ntiles]                16 * first chunk of the tile's 64-slot window
//   bh    : [ntiles][3 planes][64 lanes] 16 B horizontal band operands (B operand lane map of v_mfma_i32_16x16x64_i8)
//   ch    : int32 [ntiles][16]                128 * sum(w) + 2^21 - 2^29 per element column
//   jt    : int32 [njt][4]                    {last 16-row block of the window, super-blocks spanned (1 or 2), 0, 0}
//   wv    : [njt][2 ks][3 planes][64 lanes] 16 B  vertical band operands; ks 0 = the tile's last super-block, 1 = the one before
//   cv    : int32 [ceil16(oH)]                128 * sum(w) + 2^21 per output row

__host__ __device__ inline size_t aa_plan_align(size_t x) { return (x + 127) & ~(size_t)127; }

struct AAPlanGeom {
  int C, npx, ntiles, tps, nstrips, nch, njt, nsb;
  size_t off_strip, off_tile, off_bh, off_ch, off_jt, off_wv, off_cv, total;
};

// host arithmetic only: seg_px = input pixels the windows of one strip can cover (from the table header's measured span)
inline AAPlanGeom aa_plan_geometry(int C, int H, int W, int oH, int oW, int seg_px, int tiles_per_strip = 16) {
  AAPlanGeom g;
  g.C = C;
  g.npx = C == 3 ? 4 : (C == 4 ? 4 : 16);
  g.ntiles = (oW + g.npx - 1) / g.npx;
  g.tps = tiles_per_strip;
  g.nstrips = (g.ntiles + g.tps - 1) / g.tps;
  const int seg_bytes = seg_px * C;
  g.nch = ((seg_bytes + 15) / 16) | 1;  // odd: rows of the LDS image then start 4 * nch dwords apart and the 16 rows a
  if (g.nch < 5) g.nch = 5;             // ds_read_b128 touches fall into 16 different bank groups
  g.njt = (oH + 15) / 16;
  g.nsb = (H + 63) / 64;
  size_t o = sizeof(aa_plan_header);
  g.off_strip = o; o = aa_plan_align(o + 4 * (size_t)g.nstrips);
  g.off_tile = o; o = aa_plan_align(o + 4 * (size_t)g.ntiles);
  g.off_bh = o; o = aa_plan_align(o + (size_t)g.ntiles * 3 * 1024);
  g.off_ch = o; o = aa_plan_align(o + (size_t)g.ntiles * 64);
  g.off_jt = o; o = aa_plan_align(o + (size_t)g.njt * 16);
  g.off_wv = o; o = aa_plan_align(o + (size_t)g.njt * 6 * 1024);
  g.off_cv = o; o = aa_plan_align(o + (size_t)((oH + 15) & ~15) * 4);
  g.total = o;
  return g;
}

// ---- plan build kernels ------------------------------------------------------------------------------------------------
// Pillow table arrays: xmin[out], xsize[out], w[out * ksize] (int32, 22-bit fixed point)
struct AAPilAxis { const int32_t *xmin, *xsize, *w; int ksize, in_size, out_size; };

__device__ inline void aa_split_digits(int w, int &d0, int &d1, int &d2) {
  d0 = ((w + 128) & 255) - 128;
  const int w1 = (w - d0) >> 8;
  d1 = ((w1 + 128) & 255) - 128;
  d2 = (w1 - d1) >> 8;
}

// one workgroup of 64 lanes per tile
__global__ void aa_plan_build_h(char *plan, AAPilAxis ax) {
  aa_plan_header *hd = (aa_plan_header *)plan;
  const int tile = blockIdx.x, lane = threadIdx.x, g = lane >> 4, n = lane & 15;
  const int C = hd->C, npx = hd->npx;
  const int strip = tile / hd->tiles_per_strip;
  const int px_strip = strip * hd->tiles_per_strip * npx;
  const int seg_first = ax.xmin[px_strip] * C;
  const int px0 = tile * npx;
  int ch0 = (ax.xmin[px0] * C - seg_first) >> 4;
  if (ch0 > hd->nch - 4) ch0 = hd->nch - 4;
  if (ch0 < 0) ch0 = 0;
  const int pl = (px0 + npx - 1 < ax.out_size ? px0 + npx - 1 : ax.out_size - 1);  // last valid pixel of the tile
  const int win_end = (ax.xmin[pl] + ax.xsize[pl]) * C - seg_first;                // one past the last byte the tile reads
  if (lane == 0) {
    ((int32_t *)(plan + hd->off_tile))[tile] = ch0 * 16;
    if (tile % hd->tiles_per_strip == 0) ((int32_t *)(plan + hd->off_strip))[strip] = seg_first;
    if (win_end > ch0 * 16 + 64 || ax.xmin[px0] * C - seg_first < ch0 * 16 || win_end > hd->nch * 16) atomicAnd(&hd->fits, 0);
  }
  int8_t b[3][16];
  for (int p = 0; p < 3; p++)
    for (int j = 0; j < 16; j++) b[p][j] = 0;
  long long sumw = 0;
  const int px = px0 + n / C, c = n % C;
  if (n < C * npx && px < ax.out_size) {
    const int xm = ax.xmin[px], xs = ax.xsize[px];
    for (int k = 0; k < xs && k < ax.ksize; k++) sumw += ax.w[(size_t)px * ax.ksize + k];
    for (int j = 0; j < 16; j++) {
      const int rb = seg_first + ch0 * 16 + g * 16 + j - c;  // byte of the row, minus the channel
      if (rb < 0 || rb % C != 0) continue;
      const int k = rb / C - xm;
      if (k < 0 || k >= xs || k >= ax.ksize) continue;
      int d0, d1, d2;
      aa_split_digits(ax.w[(size_t)px * ax.ksize + k], d0, d1, d2);
      b[0][j] = (int8_t)d0; b[1][j] = (int8_t)d1; b[2][j] = (int8_t)d2;
    }
  }
  for (int p = 0; p < 3; p++) {
    int4 v;
    memcpy(&v, b[p], 16);
    ((int4 *)(plan + hd->off_bh))[((size_t)tile * 3 + p) * 64 + lane] = v;
  }
  if (g == 0) ((int32_t *)(plan + hd->off_ch))[tile * 16 + n] = (int32_t)(128 * sumw + (1 << 21) - (1 << 29));
}

// one workgroup of 64 lanes per (output-row tile, ks)
__global__ void aa_plan_build_v(char *plan, AAPilAxis ay) {
  aa_plan_header *hd = (aa_plan_header *)plan;
  const int jt = blockIdx.x >> 1, ks = blockIdx.x & 1, lane = threadIdx.x, g = lane >> 4, c = lane & 15;
  const int oy0 = jt * 16;
  const int oyl = (oy0 + 15 < ay.out_size ? oy0 + 15 : ay.out_size - 1);
  const int ya = ay.xmin[oy0];
  int yb = 0;
  for (int o = oy0; o <= oyl; o++) { const int e = ay.xmin[o] + (ay.xsize[o] > 1 ? ay.xsize[o] : 1); yb = e > yb ? e : yb; }
  const int sbA = ya / 64, sbB = (yb - 1) / 64;
  const int ksv = sbB - sbA + 1;
  if (lane == 0 && ks == 0) {
    int32_t *ji = (int32_t *)(plan + hd->off_jt) + jt * 4;
    ji[0] = (yb - 1) / 16; ji[1] = ksv; ji[2] = 0; ji[3] = 0;
    if (ksv > 2) atomicAnd(&hd->fits, 0);
  }
  int8_t b[3][16];
  for (int p = 0; p < 3; p++)
    for (int j = 0; j < 16; j++) b[p][j] = 0;
  const int oy = oy0 + c;
  long long sumw = 0;
  if (oy < ay.out_size) {
    const int ym = ay.xmin[oy], ys = ay.xsize[oy];
    for (int k = 0; k < ys && k < ay.ksize; k++) sumw += ay.w[(size_t)oy * ay.ksize + k];
    for (int j = 0; j < 16; j++) {
      const int y = 64 * (sbB - ks) + 16 * (j >> 2) + 4 * g + (j & 3);
      const int k = y - ym;
      if (sbB - ks < 0 || k < 0 || k >= ys || k >= ay.ksize) continue;
      int d0, d1, d2;
      aa_split_digits(ay.w[(size_t)oy * ay.ksize + k], d0, d1, d2);
      b[0][j] = (int8_t)d0; b[1][j] = (int8_t)d1; b[2][j] = (int8_t)d2;
    }
  }
  for (int p = 0; p < 3; p++) {
    int4 v;
    memcpy(&v, b[p], 16);
    ((int4 *)(plan + hd->off_wv))[(((size_t)jt * 2 + ks) * 3 + p) * 64 + lane] = v;
  }
  if (g == 0 && ks == 0) ((int32_t *)(plan + hd->off_cv))[oy0 + c] = oy < ay.out_size ? (int32_t)(128 * sumw + (1 << 21)) : 0;
}

// after aa_plan_build_v: most output-row tiles completing in one 16-row block (one thread)
__global__ void aa_plan_finish(char *plan) {
  aa_plan_header *hd = (aa_plan_header *)plan;
  const int32_t *ji = (const int32_t *)(plan + hd->off_jt);
  int best = 0, run = 0, prev = -1;
  for (int jt = 0; jt < hd->njt; jt++) {
    const int xe = ji[jt * 4];
    if (xe < prev) atomicAnd(&hd->fits, 0);  // (window ends are non-decreasing)
    run = xe == prev ? run + 1 : 1;
    prev = xe;
    best = run > best ? run : best;
  }
  hd->max_jt_per_blk = best;
  if (best > 2) atomicAnd(&hd->fits, 0);
}

// ---- the kernel -----------------------------------------------------------------------------------------------------------
struct FusedU8MfmaParams {
  const uint8_t *in;
  uint8_t *out;
  const char *plan;
  int H, W, oH, oW;
  int nstrips, tps, nch, njt, nsb, ntiles;
  int off_strip, off_tile, off_bh, off_ch, off_jt, off_wv, off_cv;
  unsigned plan_bytes;
  int in_mis;  // (input pointer & 15): the kernel gets the pointer rounded down to 16 B
  unsigned long long img_in_bytes, img_out_bytes, total_in_bytes, total_out_bytes;
  long long n_images;
  int x4;      // output rows, strips and the output pointer are 16-byte aligned: finished rows leave as dwordx4 stores
};

// bytes of dynamic LDS the kernel needs
inline int aa_mfma_out_pitch(int strip_bytes) { return (((strip_bytes + 15) / 16) | 1) * 16; }  // odd number of 16-byte pieces
inline size_t aa_mfma_lds_bytes(int R, int nch, int oH, int njt, int strip_bytes) {
  return (size_t)R * nch * 256 + 2 * 6144 + 2 * 16 * (size_t)aa_mfma_out_pitch(strip_bytes) + (size_t)((oH + 15) & ~15) * 4 + (size_t)njt * 8;
}

namespace aa_mfma {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

// clip8((v >> 22)) of four values packed into one dword, unsigned (Pillow's clip8) or signed (the same minus 128 when 2^29 was
// subtracted from v).  NEVER feed these an MFMA result directly: hipcc pads no MFMA->VALU wait states in front of inline asm.
__device__ inline unsigned pack4_u8(int a0, int a1, int a2, int a3) {
  unsigned d;
  asm("v_ashr_pk_u8_i32 %0, %1, %2, 22\n\tv_ashr_pk_u8_i32 %0, %3, %4, 22 op_sel:[0,0,0,1]" : "=&v"(d) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
  return d;
}
__device__ inline unsigned pack4_i8(int a0, int a1, int a2, int a3) {
  unsigned d;
  asm("v_ashr_pk_i8_i32 %0, %1, %2, 22\n\tv_ashr_pk_i8_i32 %0, %3, %4, 22 op_sel:[0,0,0,1]" : "=&v"(d) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
  return d;
}

__device__ inline void wait_vmcnt_le(int n) {  // wait until at most n vector-memory operations are outstanding (rounding n DOWN only waits longer)
  if (n >= 24) { asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); return; }
  if (n >= 20) { asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); return; }
  if (n >= 16) { asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); return; }
  switch (n) {
    case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); return;
    case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); return;
    case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); return;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); return;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); return;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); return;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); return;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); return;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); return;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); return;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); return;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); return;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); return;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); return;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); return;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return;
  }
}

// C: channels; EPT = C * NPX elements per tile (12 for C = 3); R: ring of staged 16-row blocks; X4: dwordx4 output stores
// T: tiles per wave (a strip = 4 T tiles)
template <int C, int NPX, int R, bool X4, int T>
__global__ void __launch_bounds__(256) fused_u8_nhwc_mfma_kernel(const FusedU8MfmaParams p) {
  constexpr int EPT = C * NPX;        // useful element columns per tile
  constexpr int DPT = (EPT + 3) / 4;  // dwords per tile and output row
  constexpr int SB = 4 * T * EPT;     // bytes of an output row a strip covers (192 for T = 4)
  constexpr int kOutPitch = (((SB + 15) / 16) | 1) * 16;  // bytes per row of the LDS output tile: an odd number of 16-byte pieces
                                                           // (aligned 16-byte reads, conflict-free dword writes)
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, n16 = lane & 15;

  // XCD-aware mapping: consecutive workgroup ids go round-robin to the 8 XCDs; within an XCD consecutive ids walk the strips of
  // one image first, so neighbouring strips (shared input sectors, adjacent output pieces) meet in the same L2
  const int xcd = blockIdx.x & 7, kk = blockIdx.x >> 3;
  const int strip = kk % p.nstrips;
  const long long n = (long long)(kk / p.nstrips) * 8 + xcd;
  if (n >= p.n_images) return;

  // LDS map (byte offsets; the kernel has no static LDS, so the dynamic array starts at 0): ring of R blocks | 2 Bv operand
  // buffers | 2 output tiles | per-output-row constants | table of output-row tiles
  const int blk_bytes = p.nch * 256;
  const unsigned wv_base = (unsigned)(R * blk_bytes);
  const unsigned ot_base = wv_base + 2u * 6144u;
  const unsigned cv_base = ot_base + 2u * 16u * kOutPitch;
  const unsigned jt_base = cv_base + (unsigned)((p.oH + 15) & ~15) * 4u;  // {last block of the window, super-blocks spanned} per output-row tile
  typedef const __attribute__((address_space(3))) v4i lds_v4i;
  typedef __attribute__((address_space(3))) int32_t lds_i32;
  typedef __attribute__((address_space(3))) unsigned lds_u32;

  // ---- plan views ----
  const int32_t *strip_tab = (const int32_t *)(p.plan + p.off_strip);
  const int32_t *tile_tab = (const int32_t *)(p.plan + p.off_tile);
  const v4i *bh = (const v4i *)(p.plan + p.off_bh);
  const int32_t *chh = (const int32_t *)(p.plan + p.off_ch);
  const int32_t *jti = (const int32_t *)(p.plan + p.off_jt);
  const int32_t *cvg = (const int32_t *)(p.plan + p.off_cv);

  const int seg_first = __builtin_amdgcn_readfirstlane(strip_tab[strip]);
  const int tile0 = strip * p.tps + wv * T;  // this wave's first tile
  v4i B[T][3], C0[T];
  unsigned aoff[T];
#pragma unroll
  for (int t = 0; t < T; t++) {
    const int tg = tile0 + t < p.ntiles ? tile0 + t : p.ntiles - 1;
#pragma unroll
    for (int pl = 0; pl < 3; pl++) B[t][pl] = bh[((size_t)tg * 3 + pl) * 64 + lane];
    const int c = chh[tg * 16 + n16];
    C0[t] = v4i{c, c, c, c};
    aoff[t] = (unsigned)tile_tab[tg] + (unsigned)n16 * (unsigned)(p.nch * 16) + (unsigned)g * 16u;  // row n16 of the block, chunk ch0 + g
  }
  for (int i = threadIdx.x; i < ((p.oH + 15) & ~15); i += 256) *(lds_i32 *)(uintptr_t)(cv_base + 4u * i) = cvg[i];
  // (the table of output-row tiles is read from LDS in the loop: a vector load there would have to drain vmcnt, i.e. every DMA in flight)
  for (int i = threadIdx.x; i < p.njt * 2; i += 256) *(lds_i32 *)(uintptr_t)(jt_base + 4u * i) = jti[(i >> 1) * 4 + (i & 1)];
  auto jt_info = [&](int j, int which) -> int {  // uniform
    return __builtin_amdgcn_readfirstlane(*(lds_i32 *)(uintptr_t)(jt_base + 8u * (unsigned)j + 4u * (unsigned)which));
  };
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // from here on vmcnt counts this wave's DMAs and stores

  // ---- input: range-checked view of this image ----
  const unsigned long long img_off = (unsigned long long)p.in_mis + (unsigned long long)n * p.img_in_bytes;
  const unsigned long long base_off = img_off & ~15ull;
  unsigned long long remaining = p.total_in_bytes - base_off;
  remaining = (remaining + 3ull) & ~3ull;
  if (remaining > 0xFFFFFFFCull) remaining = 0xFFFFFFFCull;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(p.in + base_off), 0, (unsigned)remaining, 0x00020000);
  const __amdgpu_buffer_rsrc_t prsrc = __builtin_amdgcn_make_buffer_rsrc((void *)p.plan, 0, p.plan_bytes, 0x00020000);
  const unsigned row_bytes = (unsigned)p.W * C;
  // a DMA instruction stages rpi consecutive rows (the vector memory path takes ~40 cycles per LDS-DMA instruction whatever it
  // carries — measured — so every instruction should carry close to 64 lanes): lane = (row of the instruction, chunk of the row)
  const int rpi = p.nch <= 32 ? 64 / p.nch : 1;
  const int lrow = lane / p.nch, lch = lane - lrow * p.nch;
  const unsigned voff = (unsigned)lrow * row_bytes + (unsigned)lch * 16u;
  const unsigned a_img = (unsigned)(img_off - base_off) + (unsigned)seg_first;
  const int row_pitch = p.nch * 16;  // (waves 0-2 issue the instructions 0, 1, 2, ... of a block in turn)
  const int nblk = p.nsb * 4;

  // ---- output ----
  const unsigned long long out_off = (unsigned long long)n * p.img_out_bytes;
  unsigned long long out_rem = p.total_out_bytes - out_off;
  if (out_rem > 0xFFFFFFFFull) out_rem = 0xFFFFFFFFull;
  const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(p.out + out_off), 0, (unsigned)out_rem, 0x00020000);
  const unsigned out_row_bytes = (unsigned)p.oW * C;

  int issued = 0;   // vector-memory operations this wave has issued (uniform).  Never count an operation that may not be issued:
                    // a count too high lets a wait pass too early
  int marks = 0;    // lane s: `issued` right after this wave's DMAs of the block in ring slot s
  int markw[2] = {0, 0};  // wave 3: `issued` right after the DMAs of the Bv operands in buffer 0 / 1

  auto dma_block = [&](int x, int slot) {  // this wave's share of block x (rows 16 x .. 16 x + 15) into ring slot `slot`
    if (AA_MFMA_ABL != 4 && wv < AA_MFMA_LOADERS) {
      uint8_t *dst = lds + slot * blk_bytes;
      const unsigned arow = a_img + (unsigned)(16 * (AA_MFMA_SKEL == 7 ? (x & 1) : x)) * row_bytes;
      for (int m = wv * rpi; m < 16; m += AA_MFMA_LOADERS * rpi) {  // rows m .. m + rpi - 1 (lane 0 always takes part: the instruction is certainly issued)
        const int nr = 16 - m < rpi ? 16 - m : rpi;
        if (lrow < nr) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void *)(dst + m * row_pitch), 16, voff, (arow + (unsigned)m * row_bytes) & ~(unsigned)AA_MFMA_DMA_MASK, 0, 0);
        issued++;
      }
    }
    marks = (lane == slot) ? issued : marks;
  };

  // the horizontal-pass results of two super-blocks: hq[s][t] = the A operand of the vertical pass (component b = block b)
  v4i hq[2][T];
#pragma unroll
  for (int s = 0; s < 2; s++)
#pragma unroll
    for (int t = 0; t < T; t++) hq[s][t] = v4i{0, 0, 0, 0};

  __syncthreads();  // the LDS tables
  // prologue: R - 1 blocks in flight
  for (int x = 0; x < R - 1; x++)
    if (x < nblk) dma_block(x, x);

  int jt = 0;                  // next output-row tile to complete ...
  int jt_xe = jt_info(0, 0);   // ... the block its window ends in ...
  int jt_ks = jt_info(0, 1);   // ... and the super-blocks it spans
  int jw = 0;                  // next output-row tile whose Bv operands are to be staged
  int js = 0;                  // next output-row tile to be stored from its LDS output tile
  int slot = 0;                // ring slot of block x

  // finished output-row tiles [js, jt) sit in the LDS output tiles (buffer = tile & 1): wave w stores rows 4 w .. 4 w + 3
  auto store_pending = [&]() {
    for (; js < jt; js++) {
      const unsigned ob = ot_base + (unsigned)(js & 1) * 16u * kOutPitch;
      if constexpr (X4) {
        const int row = 4 * wv + lane / (SB / 16), piece = lane % (SB / 16);
        const int oy = js * 16 + row;
        const unsigned col = (unsigned)(strip * SB + piece * 16);
        const bool ok = lane < 4 * (SB / 16) && oy < p.oH && col + 16 <= out_row_bytes;
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 v = *(const __attribute__((address_space(3))) u32x4 *)(uintptr_t)(ob + (unsigned)(row * kOutPitch + piece * 16));
        if (ok && AA_MFMA_ABL != 1) __builtin_amdgcn_raw_buffer_store_b128(v, orsrc, col, (unsigned)oy * out_row_bytes, 0);
        if (AA_MFMA_ABL != 1 && js * 16 + 4 * wv < p.oH) issued++;  // (lane 0 holds row 4 wv of the tile, piece 0: when that row exists the store is certainly issued)
      } else {
#pragma unroll
        for (int i = 0; i < (4 * SB / 4 + 63) / 64; i++) {
          const int idx = i * 64 + lane;
          const int row = 4 * wv + idx / (SB / 4), dwi = idx % (SB / 4);
          const int oy = js * 16 + row;
          const unsigned col = (unsigned)(strip * SB + dwi * 4);
          const bool ok = idx < 4 * (SB / 4) && oy < p.oH && col + 4 <= out_row_bytes;
          const unsigned v = *(lds_u32 *)(uintptr_t)(ob + (unsigned)(row * kOutPitch + dwi * 4));
          if (ok && AA_MFMA_ABL != 1) __builtin_amdgcn_raw_buffer_store_b32(v, orsrc, col, (unsigned)oy * out_row_bytes, 0);
        }
        // (not counted: rows beyond oH make some of these instructions empty; an uncounted store only makes waits longer)
      }
    }
  };

  for (int sb2 = 0; sb2 < p.nsb; sb2 += 2) {
#pragma unroll
    for (int s = 0; s < 2; s++) {
      const int sb = sb2 + s;
      if (sb >= p.nsb) break;
#pragma unroll
      for (int b = 0; b < 4; b++) {
        const int x = sb * 4 + b;
        // block x has landed: this wave's own DMAs of it, and (wave 3) the Bv operands of a tile that completes in this block
        int allow = issued - __builtin_amdgcn_readlane(marks, slot);
        if (wv == 3 && jt < p.njt && jt_xe == x) {
          const int aw = issued - markw[jt & 1];
          allow = aw < allow ? aw : allow;
          if (jt + 1 < p.njt && jt_info(jt + 1, 0) == x) {
            const int aw2 = issued - markw[(jt + 1) & 1];
            allow = aw2 < allow ? aw2 : allow;
          }
        }
        if (!AA_MFMA_NOSYNC) wait_vmcnt_le(allow);
        if (AA_MFMA_SKEL != 6 && !AA_MFMA_NOSYNC) __builtin_amdgcn_s_barrier();
        // every wave has finished block x - 1 and the vertical pass behind it: its ring slot is free, output tiles written there
        // are complete, and the Bv buffer of tile jt - 1 is free
        {
          const int xn = x + R - 1;
          int nslot = slot + R - 1;
          nslot = nslot >= R ? nslot - R : nslot;
          if (xn < nblk) dma_block(xn, nslot);
        }
        while (jw < p.njt && jw < jt + 2) {  // stage the Bv operands of the next two output-row tiles (6 KiB each)
          if (wv == 3 && AA_MFMA_SKEL == 0) {
            for (int q = 0; q < 6; q++) {
              __builtin_amdgcn_raw_ptr_buffer_load_lds(prsrc, (lds_void *)(lds + wv_base + (jw & 1) * 6144 + q * 1024), 16, (unsigned)lane * 16u,
                                                       (unsigned)p.off_wv + (unsigned)(jw * 6 + q) * 1024u, 0, 0);
              issued++;
            }
            if (jw & 1) markw[1] = issued; else markw[0] = issued;
          }
          jw++;
        }
        if (AA_MFMA_SKEL == 0) store_pending();
        if (AA_MFMA_ABL != 3) {
          // ---- horizontal pass of block x for this wave's tiles, in phases over the 4 tiles so that independent MFMAs sit
          //      back to back and every dependent step finds its operands ready ----
          const unsigned blk = (unsigned)(slot * blk_bytes);
          const v4i z = {0, 0, 0, 0};
          v4i a[T], d2[T], d0[T];
#pragma unroll
          for (int t = 0; t < T; t++) a[t] = *(lds_v4i *)(uintptr_t)(blk + aoff[t]);
#pragma unroll
          for (int t = 0; t < T; t++) a[t] ^= (int)0x80808080;
#pragma unroll
          for (int t = 0; t < T; t++) d2[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[t], B[t][2], z, 0, 0, 0);
#pragma unroll
          for (int t = 0; t < T; t++) d0[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[t], B[t][0], C0[t], 0, 0, 0);
#pragma unroll
          for (int t = 0; t < T; t++) d2[t] <<= 8;
#pragma unroll
          for (int t = 0; t < T; t++) d2[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[t], B[t][1], d2[t], 0, 0, 0);
#pragma unroll
          for (int t = 0; t < T; t++) {
            const v4i val = (d2[t] << 8) + d0[t];
            hq[s][t][b] = (int)pack4_i8(val.x, val.y, val.z, val.w);
          }
        }
        // ---- vertical pass: the output-row tiles whose window ends in this block (components of hq[s] that belong to later
        //      blocks still hold rows of two super-blocks ago: their weights are zero) ----
        while (jt < p.njt && jt_xe == x) {
          if (AA_MFMA_ABL != 3 && AA_MFMA_ABL != 2) {
            const unsigned wb = wv_base + (unsigned)((jt & 1) * 6144 + lane * 16);
            v4i Bv[2][3];
#pragma unroll
            for (int pl = 0; pl < 3; pl++) Bv[0][pl] = *(lds_v4i *)(uintptr_t)(wb + pl * 1024);
            if (jt_ks == 2) {
#pragma unroll
              for (int pl = 0; pl < 3; pl++) Bv[1][pl] = *(lds_v4i *)(uintptr_t)(wb + (3 + pl) * 1024);
            }
            const int cv = *(lds_i32 *)(uintptr_t)(cv_base + 4u * (unsigned)(jt * 16 + n16));
            const v4i Cv = {cv, cv, cv, cv};
            const unsigned ob = ot_base + (unsigned)(jt & 1) * 16u * kOutPitch + (unsigned)(n16 * kOutPitch + wv * (T * EPT) + g * 4);
            const v4i z = {0, 0, 0, 0};
            const bool two = jt_ks == 2;  // (wave-uniform)
            v4i e2[T], e0[T];
#pragma unroll
            for (int t = 0; t < T; t++) e2[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(hq[s][t], Bv[0][2], z, 0, 0, 0);
#pragma unroll
            for (int t = 0; t < T; t++) e0[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(hq[s][t], Bv[0][0], Cv, 0, 0, 0);
            if (two) {
#pragma unroll
              for (int t = 0; t < T; t++) e2[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(hq[s ^ 1][t], Bv[1][2], e2[t], 0, 0, 0);
#pragma unroll
              for (int t = 0; t < T; t++) e0[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(hq[s ^ 1][t], Bv[1][0], e0[t], 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < T; t++) e2[t] <<= 8;
#pragma unroll
            for (int t = 0; t < T; t++) e2[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(hq[s][t], Bv[0][1], e2[t], 0, 0, 0);
            if (two) {
#pragma unroll
              for (int t = 0; t < T; t++) e2[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(hq[s ^ 1][t], Bv[1][1], e2[t], 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < T; t++) {
              const v4i val = (e2[t] << 8) + e0[t];
              const unsigned dw = pack4_u8(val.x, val.y, val.z, val.w);
              if (g < DPT) *(lds_u32 *)(uintptr_t)(ob + (unsigned)(t * EPT)) = dw;
            }
          }
          jt++;
          if (jt < p.njt) {
            jt_xe = jt_info(jt, 0);
            jt_ks = jt_info(jt, 1);
          } else jt_xe = -1;
        }
        slot = slot + 1 >= R ? 0 : slot + 1;
      }
    }
  }
  __builtin_amdgcn_s_barrier();
  store_pending();
}

}  // namespace aa_mfma
