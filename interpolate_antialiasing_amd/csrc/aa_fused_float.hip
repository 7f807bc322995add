// aa_fused_float.hip — fused single-launch resample for fp32 NCHW tensors (BASELINE configs 0 and 2), reference arithmetic.
//
// Second design (round 2).  The first one read every tap of every output from LDS with its own ds_read_b32: for the
// bicubic 1024 -> 224 case (21 taps, window starts 32/7 dwords apart) that is 24 four-and-a-half-way bank-conflicted LDS
// instructions per row, and its accumulators were indexed dynamically, which the compiler turned into scratch memory
// (16-32 B per lane, rewritten on every row: the 3.4x WRITE_SIZE of the round-1 counters).  This one:
//   * one wave = one strip of <= 64 output columns of one band of rows of one (n, c) plane, no barrier, no shared LDS;
//   * input-row segments are staged into a private G-slot LDS ring by LDS-DMA (`buffer_load_dwordx4 ... lds`, range
//     checked).  The DMA source starts at the segment's first float rounded down to a multiple of FOUR FLOATS OF THE ROW
//     (dword-aligned in memory, not necessarily 16-byte aligned), so the LDS image of every row has the same phase: the
//     float at row position x always lands at LDS offset 4 * (x - seg0), whatever W is;
//   * horizontal pass: every lane reads its window with 16-byte ALIGNED ds_read_b128 (NQ of them: a third of a conflict
//     group each, 4-6x fewer LDS cycles than per-tap reads).  The window therefore starts up to 3 floats before the
//     lane's first tap; the lane's weights are loaded shifted by that amount once, at kernel start.  Positions outside
//     the lane's own [first tap, last tap] are SKIPPED, not added with a zero weight: acc = in_window ? acc + d*w : acc
//     with wave-level lane masks kept in scalar registers (one v_cndmask per position), so a non-finite neighbour never
//     leaks in and the sum is the reference's, bit for bit (step_two_dot_two/aa_interpolation_impl.h:60-87: tap 0 first,
//     then taps 1..xsize-1 in order, product and sum rounded separately; this file is built with -ffp-contract=off).
//     The accumulator starts at -0.0f: (-0) + x == x exactly for every x, so the first tap is an assignment;
//   * vertical pass in registers, scatter form: row r's result is multiplied by the weights it has in the outputs it
//     feeds (scatter record of the H table, one scalar load per row) and added to their accumulators, which are a
//     compile-time-indexed register array.  Rows arrive in increasing order = the reference's tap order (:29-58);
//   * a finished output row is one coalesced 256-byte store per wave (strips are 64 columns: whole 128-byte lines).
// Roofline: HBM (fp32 config A: 5 514 576 B/image, config 2: 13 185 024 B/image; ~1.8-2.8 flop/B).

#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "aa_common.h"

// This file is compiled twice (Makefile): as it stands — the reference's arithmetic, bit for bit — and with -DAA_F32_FAST_BUILD into
// aa_fused_float_fast.o: the opt-in TOLERANCE mode (AA_FLAG_FAST, `precision="fast"`): every window position is one fused
// multiply-add with a zero weight outside the lane's own taps (no separately rounded product and sum, no v_cndmask, no lane masks)
// and the vertical pass accumulates with FMAs too.  Results then differ from the reference's by rounding only (BASELINE.json's
// north star allows 1e-4 relative; measured ~1e-7), and a non-finite value poisons every output whose 16-byte ALIGNED window holds it
// (0 * inf), not only those whose taps do.  The fast build holds the plane kernels for fp32 / fp16 / bf16 only.
#ifdef AA_F32_FAST_BUILD
#define AA_F32_FAST 1
#define aa_fused_float_nchw_applicable aa_fused_float_nchw_fast_applicable
#define aa_try_fused_float_nchw aa_try_fused_float_nchw_fast
#else
#define AA_F32_FAST 0
#endif

namespace {

typedef __attribute__((address_space(3))) void lds_void;
#ifndef AA_F32_RB
#define AA_F32_RB 16  // bytes per aligned window read of fp32 PLANES.  8 (2 floats per read: 22 window positions instead of 28 for the 21-tap
                      // bicubic of config 2) was built and measured SLOWER: config 2 0.188-0.198 -> 0.227 ms, tolerance mode 0.18-0.195 -> 0.21 —
                      // 11 ds_read_b64 per row instead of 7 ds_read_b128 cost more than 6 fewer multiply-add-select groups save.  16-bit
                      // elements are the other way round (their positions carry a conversion each): they read 8 bytes
#endif
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
template <int BYTES> struct ReadUnit { typedef u32x4 type; };  // one aligned LDS read of a window
template <> struct ReadUnit<8> { typedef u32x2 type; };

__device__ inline float fma_real(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ inline double fma_real(double a, double b, double c) { return __builtin_fma(a, b, c); }

struct FusedF32Params {
  int H, W, oH, oW;  // W, oW: ELEMENTS per row (pixels * channel stride for interleaved channels)
  int Wp, oWp;       // pixels per row
  int ksize_w, ksize_h;
  int ybands, nstrips, strips_per_block, strip_w;
  int nseg, seg_bytes;
  int sc_off;
  unsigned row_pitch;  // bytes between consecutive input rows (= W * element size for a dense tensor; larger for a cropped view)
  int store_nt;  // outputs far larger than the caches are stored with the streaming (nt) policy: -3 .. -8 % (they are written once and
                 // never read here; with the default policy they displace input rows that neighbouring strips and bands re-read)
  unsigned long long plane_in_bytes, plane_out_bytes, total_in_bytes, total_out_bytes;
  long long n_groups;  // (plane, band) groups = planes * ybands
};

__device__ inline void wait_vmcnt_f(int n) {  // rounding n DOWN only waits longer
  if (n >= 14) { asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); return; }
  if (n >= 12) { asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); return; }
  if (n >= 8) { asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); return; }
  if (n >= 6) { asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); return; }
  if (n >= 4) { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); return; }
  if (n >= 3) { asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); return; }
  if (n >= 2) { asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); return; }
  if (n >= 1) { asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); return; }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

#ifndef AA_F32_ANDMASK
#define AA_F32_ANDMASK 0  // developer knob, kept off.  0: positions outside the lane's own taps are skipped with v_cndmask on the sum, lane
                          // masks in scalar registers.  1: the (negated) product is AND-ed with a per-lane 0 / ~0 register and SUBTRACTED.
#endif
// skip a window position exactly: a - ((x * -w) & m).  m = ~0: a - (-(x w)) is a + x w bit for bit (IEEE subtraction is addition of
// the negation); m = 0: whatever x * -w was (a non-finite neighbour included) becomes +0.0 and a - (+0.0) == a for EVERY a, -0.0 and
// NaN included.  On gfx950 v_and_b32 and v_sub_f32 issue at the fast rate (3.3 cycles per wave), v_cndmask_b32 at 5.5
// (profiles/r02_ubench_valu_issue_rates.txt): 9.9 instead of 12.2 issue cycles per position — but the masks then live in vector
// registers (one per window position: 42 -> 54 registers for 12 positions, 79 -> 106 for 28) and the measured result, bit-identical,
// is a wash: 21-tap bicubic 0.1995 -> 0.192 ms at its best band count, fp16 bilinear unchanged, config A fp32 0.254 -> 0.271 ms
// (same box).  The select form stays.
__device__ inline float sub_masked(float a, float negprod, unsigned m) { return a - __uint_as_float(__float_as_uint(negprod) & m); }
__device__ inline double sub_masked(double a, double negprod, unsigned m) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(negprod);
  const unsigned lo = (unsigned)u & m, hi = (unsigned)(u >> 32) & m;
  return a - __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// dst = mask[lane] ? b : a, the lane mask in a scalar register pair (no per-row compare)
__device__ inline float select_by_mask(float a, float b, unsigned long long mask) {
  float d;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(mask));
  return d;
}
__device__ inline double select_by_mask(double a, double b, unsigned long long mask) {  // two halves
  const unsigned long long ua = __double_as_longlong(a), ub = __double_as_longlong(b);
  unsigned lo, hi;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(lo) : "v"((unsigned)ua), "v"((unsigned)ub), "s"(mask));
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(hi) : "v"((unsigned)(ua >> 32)), "v"((unsigned)(ub >> 32)), "s"(mask));
  return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}
template <int DT> struct RealOf { typedef float type; };
template <> struct RealOf<AA_F64> { typedef double type; };

// element <-> float: 16-bit floats are storage types only: fp32 arithmetic, fp32 intermediate, ONE rounding to
// nearest even at the store — exactly half(reference_fp32(float(x))), like the generic path's Store<> (aa_generic.hip)
template <int DT> __device__ inline float elem_to_f32(unsigned bits);  // bits: the element in the low 16 (or all 32) bits
template <> __device__ inline float elem_to_f32<AA_F32>(unsigned bits) { return __uint_as_float(bits); }
template <> __device__ inline float elem_to_f32<AA_F16>(unsigned bits) {
  union { unsigned short u; _Float16 h; } c;
  c.u = (unsigned short)bits;
  return (float)c.h;
}
template <> __device__ inline float elem_to_f32<AA_BF16>(unsigned bits) { return __uint_as_float(bits << 16); }
// fp16 windows: v_fma_mix_f32 takes the half straight from either half of the packed register (one instruction instead of a conversion
// and a multiply).  fma(float(h), w, c) with c = -0.0f IS the separately rounded product the reference computes — the conversion is
// exact and adding -0 changes neither a value nor the sign of a zero product — and with c = the accumulator it is the tolerance mode's
// fused multiply-add.  (Halves that are denormal follow the same FP16 denormal mode as v_cvt_f32_f16.)
template <int HI>
__device__ inline float fma_mix_f16(unsigned packed, float w, float c) {
  float d;
  if constexpr (HI != 0) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(packed), "v"(w), "v"(c));
  else asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(packed), "v"(w), "v"(c));
  return d;
}
template <int DT> __device__ inline unsigned f32_to_elem(float a);
template <> __device__ inline unsigned f32_to_elem<AA_F32>(float a) { return __float_as_uint(a); }
template <> __device__ inline unsigned f32_to_elem<AA_F16>(float a) {
  union { unsigned short u; _Float16 h; } c;
  c.h = (_Float16)a;
  return c.u;
}
template <> __device__ inline unsigned f32_to_elem<AA_BF16>(float a) {  // round to nearest even, NaN stays NaN
  const unsigned u = __float_as_uint(a);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x0040u;
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

// NQ: aligned 16-byte LDS reads per window (window positions TWP = EPQ*NQ >= max taps + EPQ - 1, EPQ = elements per 16
// bytes); G: staged rows per wave; NDMA: LDS-DMA instructions per staged row (segments of up to 64 * NDMA 16-byte pieces);
// MAXC: outputs one input row can feed; DT: element type of the planes (AA_F32, AA_F16, AA_BF16).
// CS: channel stride.  1: planes (NCHW), the aligned-window form described above.  3 / 4: interleaved channels (fp32
// channels_last, s2.2:752): a "plane" is an image, a row holds W*CS floats, a lane owns one output ELEMENT (pixel ox = e / CS,
// channel e % CS) so that a wave's row piece is still 256 contiguous bytes; its taps sit CS floats apart, so the window is
// read tap by tap (ds_read_b32 at the exact tap address: no shift, TWP = 4*NQ taps) — same arithmetic, same order.
template <int NQ, int G, int NDMA, int MAXC, int DT, int CS = 1>
__global__ void __launch_bounds__(512)
fused_f32_nchw_kernel(const void *__restrict__ in, void *__restrict__ out, const char *__restrict__ tab_w,
                      const char *__restrict__ tab_h, const FusedF32Params p) {
  typedef typename RealOf<DT>::type real;  // arithmetic type: double for AA_F64 planes (AA_TABLE_F64 tables), else float
  constexpr int ES = DT == AA_F64 ? 8 : (DT == AA_F32 ? 4 : 2);  // element bytes
  // bytes per aligned LDS read of a window: 16, but 8 for 16-bit elements (round 3) — a window starts anywhere on the read grid, so it
  // spans its taps + up to EPQ - 1 wasted positions, each costing its conversion, multiply, add and select: 4 elements per read
  // instead of 8 cut the positions of a 7-tap window from 16 to 12, of an 11-tap one from 24 to 16, of a 21-tap one from 40 to 28
  constexpr int RB = ES == 2 ? 8 : ((ES == 4 && CS == 1) ? AA_F32_RB : 16);
  typedef typename ReadUnit<RB>::type unit_t;
  constexpr int EPQ = RB / ES;              // elements per aligned read
  constexpr int TWP = EPQ * NQ;
  constexpr int TW = CS == 1 ? TWP - (EPQ - 1) : TWP;  // taps a lane can hold
  // more than 28 window positions: their lane masks no longer fit the scalar registers (two per position), so the AND form is used —
  // one 0 / ~0 VECTOR register per position (see sub_masked)
  constexpr bool ANDM = AA_F32_ANDMASK != 0 || TWP > 28;
  static_assert(CS == 1 || DT == AA_F32, "interleaved channels: fp32 only");
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];

  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // XCD-aware index mapping (see aa_fused_u8_v3_impl.h): workgroup id = 8 * k + xcd; within an XCD consecutive k walk
  // the strips of one (plane, band) group first, so neighbouring strips meet in the same L2
  const int sgroups = (p.nstrips + p.strips_per_block - 1) / p.strips_per_block;
  const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
  const int strip = (k % sgroups) * p.strips_per_block + wv;
  const long long grp = (long long)(k / sgroups) * 8 + xcd;
  if (grp >= p.n_groups) return;  // the grid is padded to whole rounds of 8 groups
  if (strip >= p.nstrips) return;
  const int yb = (int)(grp % p.ybands);
  const int plane = (int)(grp / p.ybands);  // n * C + c
  const int ox0 = strip * p.strip_w;
  const int bw = min(p.strip_w, p.oW - ox0);
  const int oy0 = (int)((long long)yb * p.oH / p.ybands);
  const int oy1 = (int)((long long)(yb + 1) * p.oH / p.ybands);

  const int32_t *__restrict__ xmin_w = (const int32_t *)(tab_w + aa_table_xmin_off());
  const int32_t *__restrict__ xsize_w = (const int32_t *)(tab_w + aa_table_xsize_off(p.oWp));
  const real *__restrict__ kw = (const real *)(tab_w + aa_table_w_off(p.oWp));
  const int32_t *__restrict__ ymin_h = (const int32_t *)(tab_h + aa_table_xmin_off());
  const int32_t *__restrict__ ysize_h = (const int32_t *)(tab_h + aa_table_xsize_off(p.oH));
  const int32_t *__restrict__ sc_rec = (const int32_t *)(tab_h + p.sc_off);

  // input rows this band needs: [r_begin, r_stop)
  const int r_begin = __builtin_amdgcn_readfirstlane(ymin_h[oy0]);
  const int ylm = __builtin_amdgcn_readfirstlane(ymin_h[oy1 - 1]);
  const int yls = __builtin_amdgcn_readfirstlane(ysize_h[oy1 - 1]);
  const int r_stop = ylm + (yls > 1 ? yls : 1);
  const int n_rows = r_stop - r_begin;
  const int n_groups = (n_rows + G - 1) / G;

  // ---- per-lane horizontal-pass state ------------------------------------------------------------------------
  const bool active = lane < bw;
  const int oe = ox0 + (active ? lane : 0);  // (lanes beyond the strip compute a duplicate of lane 0 and never store)
  const int ox = CS == 1 ? oe : oe / CS;     // output pixel; channel oe % CS
  const int xm = xmin_w[ox];
  int xs = xsize_w[ox];
  xs = xs > 1 ? xs : 1;  // tap 0 is unconditional in the reference (s2.2:68-73)
  xs = xs < TW ? xs : TW;
  int lead = xm + TW - p.Wp;  // right-align windows whose unused tail would leave the row
  lead = lead > 0 ? lead : 0;
  const int start = CS == 1 ? xm - lead : (xm - lead) * CS + (oe - ox * CS);  // row position (elements) of the first readable one
  const int astart = CS == 1 ? (start & ~(EPQ - 1)) : start;  // planes: rounded down to the 16-byte grid of the row image
  const int tap0 = CS == 1 ? (start & (EPQ - 1)) + lead : lead;  // window position of the reference's tap 0
  real wreg[TWP];
  unsigned long long inwin[TWP];  // lane masks (scalar registers): position q belongs to the lane's own taps
  unsigned mk[TWP];               // the same per lane: ~0 / 0
#pragma unroll
  for (int q = 0; q < TWP; q++) {
    const int j = q - tap0;
    const bool mine = j >= 0 && j < xs;
    wreg[q] = (mine && j < p.ksize_w) ? kw[(size_t)ox * p.ksize_w + j] : (real)0;
    inwin[q] = __ballot(mine);
    if (ANDM && !AA_F32_FAST) wreg[q] = -wreg[q];  // (the product is subtracted, see sub_masked)
    mk[q] = mine ? 0xFFFFFFFFu : 0u;
    asm volatile("" : "+v"(mk[q]));  // (a plain register to the compiler: or it turns the AND back into a v_cndmask on a lane mask)
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // table loads done: from here on vmcnt counts DMAs and stores
  // lane 0 is always active and its PIXEL has the smallest window start (interleaved channels: take its channel 0, a
  // neighbouring pixel with the same start and a lower channel sits before lane 0's own element)
  const int seg0 = __builtin_amdgcn_readfirstlane(CS == 1 ? astart : astart - (oe - ox * CS)) & ~(EPQ - 1);
  const unsigned lane_lds = (unsigned)(wv * G * p.seg_bytes + (astart - seg0) * ES);  // multiple of RB

  const unsigned long long plane_off = (unsigned long long)plane * p.plane_in_bytes;
  unsigned long long remaining = p.total_in_bytes - plane_off;
  // The range check works per dword, and rows of 16-bit elements with an odd W start on odd halves: in the LAST row of the LAST
  // plane the dword holding the tensor's final element then straddles the end of the tensor and is refused (zeros).  It must be:
  // extending the range by the two bytes beyond lets the load touch memory that is not the tensor's, and when the tensor ends on
  // the last byte of a mapped page that is a memory fault (seen once in 90 000 fuzz problems: a bf16 tensor of 19 x 512 bytes at
  // the end of an allocator block).  The final element is fetched on its own instead (patch_last below).
  if (remaining > 0xFFFFFFFCull) remaining = 0xFFFFFFFCull;
  const int fix_row = (ES == 2 && (p.Wp & 1) && (long long)plane + 1 == p.n_groups / p.ybands) ? p.H - 1 : -1;
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void *)((const uint8_t *)in + plane_off), 0, (unsigned)remaining, 0x00020000);
  const unsigned row_bytes = p.row_pitch;
  const int lds_base = wv * G * p.seg_bytes;
  const unsigned voff = (unsigned)lane * 16u;

  const unsigned long long out_off = (unsigned long long)plane * p.plane_out_bytes;
  unsigned long long out_rem = p.total_out_bytes - out_off;
  if (out_rem > 0xFFFFFFFFull) out_rem = 0xFFFFFFFFull;
  const __amdgpu_buffer_rsrc_t orsrc =
      __builtin_amdgcn_make_buffer_rsrc((void *)((uint8_t *)out + out_off), 0, (unsigned)out_rem, 0x00020000);
  const unsigned out_row_bytes = (unsigned)p.oW * (unsigned)ES;
  const unsigned store_voff = (unsigned)(ox0 + lane) * (unsigned)ES;

  // byte offset (from the plane) of the CURRENT row's segment
  unsigned a = (unsigned)seg0 * (unsigned)ES + (unsigned)r_begin * row_bytes;

  // ---- vertical-pass state: MAXC accumulators, A[k] belongs to output row o_base + k -----------------------------
  real A[MAXC];
#pragma unroll
  for (int k2 = 0; k2 < MAXC; k2++) A[k2] = (real)-0.0;
  int o_base = oy0;

  auto dma = [&](unsigned a_row, int slot) {
    const int dst = lds_base + slot * p.seg_bytes;
#pragma unroll
    for (int i = 0; i < NDMA; i++) {
      if (lane + 64 * i < p.nseg)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void *)(lds + dst + 1024 * i), 16, voff + 1024u * i, a_row, 0, 0);
    }
  };
  struct Scatter { int first; int cc; real w[MAXC]; };
  auto load_scatter = [&](int r) -> Scatter {  // one record: {first, count | completes << 16, w[6]} (32 bytes; 64 with double weights); r == H: sentinel
    Scatter s;
    const int32_t *rec = (const int32_t *)((const char *)sc_rec + (unsigned)r * (DT == AA_F64 ? 64u : 32u));
    s.first = __builtin_amdgcn_readfirstlane(rec[0]);
    s.cc = __builtin_amdgcn_readfirstlane(rec[1]);
#pragma unroll
    for (int k2 = 0; k2 < MAXC; k2++) {
      if constexpr (DT == AA_F64) {
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane(rec[2 + 2 * k2]), hi = (unsigned)__builtin_amdgcn_readfirstlane(rec[3 + 2 * k2]);
        s.w[k2] = __longlong_as_double(((unsigned long long)hi << 32) | lo);
      } else {
        s.w[k2] = __int_as_float(__builtin_amdgcn_readfirstlane(rec[2 + k2]));
      }
    }
    return s;
  };
  auto emit = [&](int oy) {  // accumulator 0 is complete: store it, slide the others down
    if constexpr (DT == AA_F64) {
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
      const unsigned long long bits = __double_as_longlong(A[0]);
      const u32x2 t = {(unsigned)bits, (unsigned)(bits >> 32)};
      if (active) {
        if (p.store_nt) __builtin_amdgcn_raw_buffer_store_b64(t, orsrc, store_voff, (unsigned)oy * out_row_bytes, 2);
        else __builtin_amdgcn_raw_buffer_store_b64(t, orsrc, store_voff, (unsigned)oy * out_row_bytes, 0);
      }
    } else if constexpr (DT == AA_F32) {
      if (active) {
        if (p.store_nt) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(A[0]), orsrc, store_voff, (unsigned)oy * out_row_bytes, 2);
        else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(A[0]), orsrc, store_voff, (unsigned)oy * out_row_bytes, 0);
      }
    } else {
      if (active) __builtin_amdgcn_raw_buffer_store_b16((unsigned short)f32_to_elem<DT>(A[0]), orsrc, store_voff, (unsigned)oy * out_row_bytes, 0);
    }
#pragma unroll
    for (int k2 = 0; k2 + 1 < MAXC; k2++) A[k2] = A[k2 + 1];
    A[MAXC - 1] = (real)-0.0;
  };
  float neg_zero = -0.0f;
  asm volatile("" : "+v"(neg_zero));  // (a register operand for fma_mix_f16)
  // one input row: window from LDS, reference-order accumulation over the lane's own taps, scatter into the open outputs
  auto row_step = [&](int slot, const Scatter &sc) {
    const __attribute__((address_space(3))) unit_t *src =
        (const __attribute__((address_space(3))) unit_t *)(uintptr_t)(lane_lds + (unsigned)(slot * p.seg_bytes));
    unit_t d[NQ];
    float dt[CS == 1 ? 1 : TWP];  // (interleaved channels: the taps, CS floats apart)
    if constexpr (CS == 1) {
#pragma unroll
      for (int q = 0; q < NQ; q++) d[q] = src[q];
    } else {
      const __attribute__((address_space(3))) float *st = (const __attribute__((address_space(3))) float *)src;
#pragma unroll
      for (int q = 0; q < TWP; q++) dt[q] = st[q * CS];
    }
    real acc = (real)-0.0;
#pragma unroll
    for (int q = 0; q < TWP; q++) {
      if constexpr (DT == AA_F16) {  // (see fma_mix_f16)
        const unsigned pk = d[q >> 2][(q >> 1) & 1];
        if constexpr (AA_F32_FAST != 0) {
          acc = (q & 1) ? fma_mix_f16<1>(pk, wreg[q], acc) : fma_mix_f16<0>(pk, wreg[q], acc);
        } else {
          const float prod = (q & 1) ? fma_mix_f16<1>(pk, wreg[q], neg_zero) : fma_mix_f16<0>(pk, wreg[q], neg_zero);
          if constexpr (ANDM) {
            acc = sub_masked(acc, prod, mk[q]);
          } else {
            const float sum = acc + prod;
            acc = select_by_mask(acc, sum, inwin[q]);
          }
        }
        continue;
      }
      real dq;  // window position q as a real
      if constexpr (CS != 1) dq = dt[q];
      else if constexpr (DT == AA_F64) dq = __longlong_as_double(((unsigned long long)d[q >> 1][2 * (q & 1) + 1] << 32) | d[q >> 1][2 * (q & 1)]);
      else if constexpr (DT == AA_F32) dq = __uint_as_float(d[q / EPQ][q % EPQ]);
      else dq = elem_to_f32<DT>(d[q >> 2][(q >> 1) & 1] >> (16 * (q & 1)));
      if constexpr (AA_F32_FAST != 0) {  // tolerance mode: the weight is zero outside the lane's own taps
        acc = fma_real(dq, wreg[q], acc);
        continue;
      }
      const real prod = dq * wreg[q];
      if constexpr (ANDM) {
        acc = sub_masked(acc, prod, mk[q]);
      } else {
        const real sum = acc + prod;
        acc = select_by_mask(acc, sum, inwin[q]);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the window is in registers: the caller may refill the slot)
    const int cnt = sc.cc & 0xFFFF;
    const int idx0 = sc.first - o_base;  // 0 in steady state; negative while the band's first rows still feed outputs
                                         // of the previous band
    if (__builtin_expect(idx0 == 0, 1)) {
#pragma unroll
      for (int k2 = 0; k2 < MAXC; k2++)
        if (k2 < cnt) A[k2] = AA_F32_FAST ? fma_real(acc, sc.w[k2], A[k2]) : A[k2] + acc * sc.w[k2];  // wave-uniform: only the outputs whose window holds this row
    } else if (idx0 < 0 && idx0 > -MAXC) {
#pragma unroll
      for (int s = 1; s < MAXC; s++) {
        if (idx0 == -s) {
#pragma unroll
          for (int k2 = s; k2 < MAXC; k2++)
            if (k2 < cnt) A[k2 - s] = AA_F32_FAST ? fma_real(acc, sc.w[k2], A[k2 - s]) : A[k2 - s] + acc * sc.w[k2];
        }
      }
    }
    const int sc_end = sc.first + (sc.cc >> 16);     // outputs [first, sc_end) take their LAST row here
    const int e_end = sc_end < oy1 ? sc_end : oy1;  // (outputs below o_base belong to the previous band)
    while (o_base < e_end) {
      emit(o_base);
      o_base++;
    }
  };

  // 16-bit elements, odd W: the staged image of the tensor's very last row lacks its final element (see fix_row); lane 0 reads
  // that element with an ordinary 2-byte load and puts it (and a zero for the position beyond the row) into the slot
  auto patch_last = [&](int slot) {
    const int pos = p.Wp - 1 - seg0;  // position inside the strip's segment (even: seg0 is a multiple of 4, W is odd)
    if (pos < 0 || pos >= p.nseg * (16 / ES)) return;
    if (lane == 0) {
      const unsigned short v = *(const unsigned short *)((const uint8_t *)in + plane_off + (unsigned long long)(p.H - 1) * row_bytes +
                                                         (unsigned long long)(p.Wp - 1) * 2u);
      *(unsigned *)(lds + lds_base + slot * p.seg_bytes + pos * 2) = (unsigned)v;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  };

  for (int i = 0; i < G; i++)
    if (i < n_rows) dma(a + (unsigned)i * row_bytes, i);
  int r = r_begin;
  for (int g = 0; g < n_groups; g++) {
    const int x0 = g * G;
    if (x0 + 2 * G <= n_rows) {
#pragma unroll
      for (int i = 0; i < G; i++) {
        // row x must have landed: rows x+1 .. x+G-1 (and any output stores) were issued after it
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA * (G - 1)) : "memory");
        const Scatter sc = load_scatter(r);
        row_step(i, sc);
        dma(a + (unsigned)G * row_bytes, i);
        a += row_bytes;
        r++;
      }
    } else {
      for (int i = 0; i < G; i++) {
        const int x = x0 + i;
        if (x >= n_rows) break;
        int younger = n_rows - 1 - x;
        younger = younger < G - 1 ? younger : G - 1;
        wait_vmcnt_f(younger * NDMA);
        if (ES == 2 && r == fix_row) patch_last(i);  // (the band's last row: never inside the unrolled groups above)
        const Scatter sc = load_scatter(r);
        row_step(i, sc);
        if (x + G < n_rows) dma(a + (unsigned)G * row_bytes, i);
        a += row_bytes;
        r++;
      }
    }
  }
}

int pick_ybands_f(int64_t items_per_band, double slots, int taps_h, int taps_w, int64_t H, int64_t oH, int waves_per_cu) {
  const int64_t max_yb = oH / 8 > 1 ? oH / 8 : 1;
  int64_t ybands = 1;
  double best = 1e30;
  // Wide windows (> 12 taps: rows heavy in arithmetic and staging, 20 halo rows per band): such a kernel reaches its rate with about 12
  // waves on a CU, so a last (or only) round that fills `sat` of the slots costs no more than its work — and fewer, taller bands save halo
  // rows.  Measured, config 2 (21-tap bicubic, 24 waves fit a CU), ms by band count: 2: 0.261, 3: 0.229, 4: 0.193, 5: 0.204, 6: 0.206,
  // 8 (one full round, the old choice): 0.200, 10: 0.218; tolerance mode 4: 0.180, 8: 0.194.  Narrow windows need every wave they can
  // get to hide memory latency (config A fp32, 26 waves fit: 2 bands 0.310, 10 bands 0.268): sat = 1, the plain round model.
  const double sat = (taps_w > 12 && waves_per_cu > 12) ? 12.0 / waves_per_cu : 1.0;
  for (int64_t yb = 1; yb <= max_yb && yb <= 64; yb++) {
    const double rounds = (double)items_per_band * yb / slots;
    const double whole = floor(rounds), part = rounds - whole;
    const double units = whole + (part > 1e-9 ? (part > sat ? part : sat) : 0.0);  // time, in full-occupancy rounds
    const double halo = 1.0 + (double)(yb - 1) * taps_h / (double)H;
    const double cost = halo * units / rounds;
    if (cost < best - 1e-9) {
      best = cost;
      ybands = yb;
    }
  }
  // Narrow windows (bilinear-class shapes, memory-bound): twice the bands of the round model, while the launch is fewer than 8
  // rounds — shorter work items even out the end of the kernel, and their extra halo rows cost little where the vector ALUs are
  // half idle.  Measured (ms, model | doubled): [256,3,438,906] fp32 NCHW 0.287 | 0.271, channels_last 0.283 | 0.264, [64,3,1024,1024]
  // fp16 bilinear 0.163 | 0.150; the 21-tap bicubic config (vector-ALU bound) loses with more bands (0.20 | 0.22) and keeps the model.
  // (Round 3 tried to keep the doubling to launches of 1.5 rounds and more — fp16 bilinear 1024 -> 224 prefers 8 bands, 0.102 ms, to its 18,
  // 0.107 — and lost more elsewhere: fp16 [128,3,438,906] -> (196,320) 0.099 -> 0.110, -> (196,1200) 0.228 -> 0.284.  The rule stays.)
  if (taps_w <= 12 && (double)items_per_band * ybands / slots < 8.0) ybands = 2 * ybands < max_yb ? 2 * ybands : max_yb;
  if (const char *e = aa_knob("AA_FUSED_YBANDS")) {
    const int64_t v = atoll(e);
    if (v >= 1 && v <= max_yb) ybands = v;
  }
  return (int)ybands;
}

template <int NQ, int G, int NDMA, int MAXC, int DT, int CS = 1>
int launch_k(FusedF32Params p, const AAProblem &q) {
  auto kern = fused_f32_nchw_kernel<NQ, G, NDMA, MAXC, DT, CS>;
  const size_t lds = (size_t)G * p.seg_bytes;  // per strip (wave)
  auto resident = [&](int s) {  // workgroups of s strips a CU holds (-1: their rings do not fit a workgroup's LDS)
    if (lds * s > 64 * 1024) return -1;  // (never for s == 1: a strip's ring is at most 16 KiB)
    int nb = aa_resident_blocks(kern, 64 * s, lds * s);
    if (nb <= 0) {  // a failed query only costs the heuristic its input: estimate from LDS and wave slots
      nb = (int)((160 * 1024) / (lds * s > 0 ? lds * s : 1));
      if (nb > 32 / s) nb = 32 / s;
      if (nb < 1) nb = 1;
    }
    return nb;
  };
  // strips of a band share a workgroup unless single-strip workgroups put more waves on a CU (see aa_fused_u8_v3_impl.h)
  int spb = p.strips_per_block;
  if (spb > 1 && (resident(spb) < 0 || resident(1) > resident(spb) * spb)) spb = 1;
  if (const char *e = aa_knob("AA_F32_SPB")) {  // experiment knob
    const int v = atoi(e);
    if (v >= 1 && v <= 8 && resident(v) > 0) spb = v;
  }
  p.strips_per_block = spb;
  const int sgroups = (p.nstrips + spb - 1) / spb;
  const size_t lds_blk = lds * spb;
  const int taps_h = q.ah.max_taps > 0 ? q.ah.max_taps : q.ah.ksize;
  const int64_t planes = CS == 1 ? q.N * q.C : q.N;
  const int taps_w = q.aw.max_taps > 0 ? q.aw.max_taps : q.aw.ksize;
  p.ybands = pick_ybands_f(planes * sgroups, (double)aa_device_cu_count() * resident(spb), taps_h, taps_w, q.H, q.oH, DT == AA_F32 ? resident(spb) * spb : 0);  // (the saturation model is measured for fp32 only: fp16 bicubic thumbnails lose 20 % with it)
  p.n_groups = planes * (int64_t)p.ybands;
  const int64_t grid = (p.n_groups + 7) / 8 * 8 * sgroups;
  if (grid > 0x7FFFFFFF) return 0;
  if (aa_knob("AA_F32_DEBUG"))
    fprintf(stderr, "f32: NQ=%d G=%d MAXC=%d DT=%d CS=%d nstrips=%d spb=%d resident=%d ybands=%d planes=%lld grid=%lld lds=%zu\n", NQ, G, MAXC, DT, CS,
            p.nstrips, spb, resident(spb), p.ybands, (long long)planes, (long long)grid, lds_blk);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * spb), lds_blk, q.stream, q.in, q.out,
                     (const char *)q.aw.table_dev, (const char *)q.ah.table_dev, p);
  AA_HIP_CHECK_LAUNCH();
  return 1;
}

template <int NQ, int G, int NDMA, int DT, int CS = 1>
int launch_m(int maxc, const FusedF32Params &p, const AAProblem &q) {
  if (maxc <= 2) return launch_k<NQ, G, NDMA, 2, DT, CS>(p, q);
  if (maxc <= 3) return launch_k<NQ, G, NDMA, 3, DT, CS>(p, q);
  if (maxc <= 4) return launch_k<NQ, G, NDMA, 4, DT, CS>(p, q);
  return launch_k<NQ, G, NDMA, 6, DT, CS>(p, q);
}

// staged rows per wave: 8 while a row segment is one DMA instruction (<= 1 KiB), 4 beyond (rings stay <= 8 KiB per wave)
template <int NQ, int DT, int CS = 1>
int launch_q(int maxc, const FusedF32Params &p, const AAProblem &q) {
#ifndef AA_F32_G_WIDE
#define AA_F32_G_WIDE 4  // staged rows of segments beyond one DMA.  8 measured the same within noise on config 2 (exact 0.192 | 0.193 ms, tolerance mode
                         // 0.162-0.176 | 0.166-0.180): the deeper ring buys nothing
#endif
  return p.nseg <= 64 ? launch_m<NQ, 8, 1, DT, CS>(maxc, p, q) : launch_m<NQ, AA_F32_G_WIDE, 2, DT, CS>(maxc, p, q);
}

template <int CS>
int launch_interleaved(int nq, int maxc, const FusedF32Params &p, const AAProblem &q) {
  switch (nq) {
    case 2: return launch_q<2, AA_F32, CS>(maxc, p, q);
    case 3: return launch_q<3, AA_F32, CS>(maxc, p, q);
    case 4: return launch_q<4, AA_F32, CS>(maxc, p, q);
    case 5: return launch_q<5, AA_F32, CS>(maxc, p, q);
    case 7: return launch_q<7, AA_F32, CS>(maxc, p, q);
    default: return launch_q<9, AA_F32, CS>(maxc, p, q);  // (36 taps: test.py's bicubic 906 -> 120 thumbnails, 33 taps)
  }
}

// window quads for a table whose widest window has `taps` taps: EPQ * NQ - (EPQ - 1) >= taps.  fp32: 2,3,4,5,7 quads of 4
// floats (5 .. 25 taps) and 9 (33 taps; beyond 28 window positions the lane masks live in vector registers, see ANDM); 16-bit floats read
// 8 bytes = 4 elements at a time and use the same table
int quads_for(int taps, int epq) {
  if (epq == 1) {  // fp32 planes read 2 floats at a time: taps <= 2 * NQ - 1
    const int opts[] = {3, 4, 5, 6, 8, 11, 14, 17};
    for (int o : opts)
      if (taps <= 2 * o - 1) return o;
    return 0;
  }
  if (epq == 2) {  // doubles: 2 per aligned read, taps <= 2 * NQ - 1
    const int opts[] = {2, 4, 6, 8, 11};
    for (int o : opts)
      if (taps <= 2 * o - 1) return o;
    return 0;
  }
  if (epq == 4) {
    const int opts[] = {2, 3, 4, 5, 7, 9, 11};  // (9: 33 taps — test.py's bicubic 906 -> 120 thumbnails; 11: 41 — 4K -> 224 bilinear)
    for (int o : opts)
      if (taps <= 4 * o - 3) return o;
    return 0;
  }
  return 0;
}

struct F32Geometry { int nq, nstrips, strip_w, nseg, cs; };

bool f32_geometry(int dtype, int layout, int64_t C, int64_t W, const aa_axis &aw, F32Geometry *g) {
  const int es = dtype == AA_F64 ? 8 : (dtype == AA_F32 ? 4 : 2);
  const int epq = es == 2 ? 4 : (es == 4 ? AA_F32_RB / 4 : 2);  // elements per aligned window read of a plane (see RB in the kernel)
  const int pe = 16 / es;                 // elements per staged 16-byte piece
  const int taps_w = aw.max_taps > 0 ? aw.max_taps : aw.ksize;
  g->cs = (layout == AA_NHWC && C > 1) ? (int)C : 1;
  if (aw.span64p1 <= 0) return false;
  if (g->cs != 1) {  // interleaved channels (fp32, 3 or 4 of them): a lane per output element, taps read one by one
    if (dtype != AA_F32 || (C != 3 && C != 4) || aw.span4p1 <= 0) return false;
    g->nq = taps_w <= 8 ? 2 : (taps_w <= 12 ? 3 : (taps_w <= 16 ? 4 : (taps_w <= 20 ? 5 : (taps_w <= 28 ? 7 : (taps_w <= 36 ? 9 : 0)))));
    if (g->nq == 0 || W < 4 * g->nq) return false;
    const int64_t oWe = aw.out_size * C;
    g->strip_w = 64;
    g->nstrips = (int)((oWe + 63) / 64);
    // pixels the strip's 64 elements span: ceil(63 / C) + 1; the spread of their window starts, bounded through the measured
    // spread of 4 neighbours (3 steps) and of 64
    const int steps = (63 / (int)C + 1 + 2) / 3;
    int spread = steps * (aw.span4p1 - 1);
    if (spread > aw.span64p1 - 1) spread = aw.span64p1 - 1;
    const int span = (spread + 4 * g->nq) * (int)C + 3 + (int)C;  // elements (+3: segment start rounded down to 4)
    g->nseg = (span + 3) / 4 + 1;
    if (g->nseg > 128) {  // strong down-scaling (test.py's 906 -> 120 thumbnails): strips of 32 elements, as for planes below
      const int steps32 = (31 / (int)C + 1 + 2) / 3;
      int spread32 = steps32 * (aw.span4p1 - 1);
      if (spread32 > aw.span64p1 - 1) spread32 = aw.span64p1 - 1;
      g->strip_w = 32;
      g->nstrips = (int)((oWe + 31) / 32);
      g->nseg = ((spread32 + 4 * g->nq) * (int)C + 3 + (int)C + 3) / 4 + 1;
      if (g->nseg > 128) {  // stronger still (3840 -> 224): strips of 16 elements
        const int steps16 = (15 / (int)C + 1 + 2) / 3;
        int spread16 = steps16 * (aw.span4p1 - 1);
        if (spread16 > aw.span64p1 - 1) spread16 = aw.span64p1 - 1;
        g->strip_w = 16;
        g->nstrips = (int)((oWe + 15) / 16);
        g->nseg = ((spread16 + 4 * g->nq) * (int)C + 3 + (int)C + 3) / 4 + 1;
      }
    }
    return g->nseg <= 128;
  }
  g->nq = quads_for(taps_w, es == 4 && epq == 2 ? 1 : epq);  // (1: the fp32 table of 8-byte reads)
  if (g->nq == 0 || W < epq * g->nq - (epq - 1)) return false;
  const int64_t oW = aw.out_size;
  g->strip_w = 64;  // whole 128-byte lines per stored fp32 row piece (the last strip may be shorter)
  g->nstrips = (int)((oW + 63) / 64);
  // elements a strip's windows cover: the spread of 64 window starts (+EPQ-1: the first one rounded down to the 16-byte
  // grid) + one window; in 16-byte pieces
  g->nseg = (aw.span64p1 + (epq - 1) + epq * g->nq + (pe - 1)) / pe;
  if (g->nseg > 128 && aw.span4p1 > 0) {  // strong down-scaling: strips of 32 columns (half the lanes idle; such a shape is bound by its input stream)
    const int by4 = 11 * (aw.span4p1 - 1) + 1;
    const int span32 = by4 < aw.span64p1 ? by4 : aw.span64p1;
    g->strip_w = 32;
    g->nstrips = (int)((oW + 31) / 32);
    g->nseg = (span32 + (epq - 1) + epq * g->nq + (pe - 1)) / pe;
    if (g->nseg > 128) {  // stronger still (1920 -> 128: 15 x): strips of 16 columns — the input stream is what such a shape costs
      const int by4s = 5 * (aw.span4p1 - 1) + 1;
      const int span16 = by4s < aw.span64p1 ? by4s : aw.span64p1;
      g->strip_w = 16;
      g->nstrips = (int)((oW + 15) / 16);
      g->nseg = (span16 + (epq - 1) + epq * g->nq + (pe - 1)) / pe;
    }
  }
  return g->nseg <= 128;
}

}  // namespace

bool aa_fused_float_nchw_applicable(int dtype, int layout, int64_t N, int64_t C, int64_t H, int64_t W, const aa_axis *ah,
                                    const aa_axis *aw) {
  if (dtype != AA_F32 && dtype != AA_F16 && dtype != AA_BF16 && dtype != AA_F64) return false;
  if (AA_F32_FAST && (dtype == AA_F64 || (layout == AA_NHWC && C > 1))) return false;  // (the tolerance build: planes of fp32 / fp16 / bf16)
  if (layout != AA_NCHW && layout != AA_NHWC) return false;
  const int want_kind = dtype == AA_F64 ? AA_TABLE_F64 : AA_TABLE_F32;
  if (!ah || !aw || ah->kind != want_kind || aw->kind != want_kind) return false;
  if (ah->scatter_off <= 0 || ah->scatter_max <= 0 || ah->scatter_max > 6) return false;
  if (H < ah->out_size) return false;
  F32Geometry g;
  if (!f32_geometry(dtype, layout, C, W, *aw, &g)) return false;
  if ((uint64_t)H * W * 8 * g.cs > 0xFFFFFFF0ull) return false;
  if ((uint64_t)ah->out_size * aw->out_size * 8 * g.cs > 0xFFFFFFF0ull) return false;
  if (!aa_grid_fits(N * C * g.nstrips)) return false;
  return true;
}

int aa_try_fused_float_nchw(const AAProblem &q, const char **variant) {
  if (!aa_fused_float_nchw_applicable(q.dtype, q.layout, q.N, q.C, q.H, q.W, &q.ah, &q.aw)) return 0;
  const int es = q.dtype == AA_F64 ? 8 : (q.dtype == AA_F32 ? 4 : 2);
  if (((uintptr_t)q.out & (es - 1)) != 0 || ((uintptr_t)q.in & (es - 1)) != 0) return 0;
  F32Geometry g;
  f32_geometry(q.dtype, q.layout, q.C, q.W, q.aw, &g);

  FusedF32Params p;
  p.H = (int)q.H; p.W = (int)q.W * g.cs; p.oH = (int)q.oH; p.oW = (int)q.oW * g.cs;
  p.Wp = (int)q.W; p.oWp = (int)q.oW;
  p.ksize_w = q.aw.ksize; p.ksize_h = q.ah.ksize;
  p.plane_in_bytes = (unsigned long long)q.H * q.W * es * (g.cs == 1 ? 1 : q.C);
  p.row_pitch = (unsigned)(q.W * es * (g.cs == 1 ? 1 : q.C));
  if (q.in_row_pitch) {  // a pitched view: rows / planes (images) these many bytes apart
    if ((uint64_t)q.H * (uint64_t)q.in_row_pitch > 0x7FFFFFF0ull || (q.in_row_pitch & (es - 1)) || (q.in_img_pitch & (es - 1))) return 0;
    p.row_pitch = (unsigned)q.in_row_pitch;
    p.plane_in_bytes = (unsigned long long)q.in_img_pitch;
  }
  p.plane_out_bytes = (unsigned long long)q.oH * q.oW * es * (g.cs == 1 ? 1 : q.C);
  const unsigned long long planes = (unsigned long long)(g.cs == 1 ? q.N * q.C : q.N);
  p.total_in_bytes = q.in_row_pitch ? p.plane_in_bytes * (planes - 1) + (unsigned long long)(q.H - 1) * p.row_pitch + (unsigned long long)q.W * es * (g.cs == 1 ? 1 : q.C)
                                    : p.plane_in_bytes * planes;
  p.total_out_bytes = p.plane_out_bytes * planes;
  p.sc_off = q.ah.scatter_off;
  p.store_nt = p.total_out_bytes > (64ull << 20) ? 1 : 0;
  p.nstrips = g.nstrips;
  p.strip_w = g.strip_w;
  p.strips_per_block = p.nstrips <= 8 ? p.nstrips : 4;
  p.nseg = g.nseg;
  p.seg_bytes = p.nseg * 16;
  p.ybands = 1;
  p.n_groups = 0;

  int rc = 0;
  const int mc = q.ah.scatter_max;
#if !AA_F32_FAST
  if (g.cs == 3) rc = launch_interleaved<3>(g.nq, mc, p, q);
  else if (g.cs == 4) rc = launch_interleaved<4>(g.nq, mc, p, q);
  else
#endif
  if (q.dtype == AA_F32) {
    switch (g.nq) {
#if AA_F32_RB == 8
      case 3: rc = launch_q<3, AA_F32>(mc, p, q); break;
      case 4: rc = launch_q<4, AA_F32>(mc, p, q); break;
      case 5: rc = launch_q<5, AA_F32>(mc, p, q); break;
      case 6: rc = launch_q<6, AA_F32>(mc, p, q); break;
      case 8: rc = launch_q<8, AA_F32>(mc, p, q); break;
      case 11: rc = launch_q<11, AA_F32>(mc, p, q); break;
      case 14: rc = launch_q<14, AA_F32>(mc, p, q); break;
      default: rc = launch_q<17, AA_F32>(mc, p, q); break;
#else
      case 2: rc = launch_q<2, AA_F32>(mc, p, q); break;
      case 3: rc = launch_q<3, AA_F32>(mc, p, q); break;
      case 4: rc = launch_q<4, AA_F32>(mc, p, q); break;
      case 5: rc = launch_q<5, AA_F32>(mc, p, q); break;
      case 7: rc = launch_q<7, AA_F32>(mc, p, q); break;
      case 9: rc = launch_q<9, AA_F32>(mc, p, q); break;
      default: rc = launch_q<11, AA_F32>(mc, p, q); break;
#endif
    }
#if !AA_F32_FAST
  } else if (q.dtype == AA_F64) {
    switch (g.nq) {
      case 2: rc = launch_q<2, AA_F64>(mc, p, q); break;
      case 4: rc = launch_q<4, AA_F64>(mc, p, q); break;
      case 6: rc = launch_q<6, AA_F64>(mc, p, q); break;
      case 8: rc = launch_q<8, AA_F64>(mc, p, q); break;
      default: rc = launch_q<11, AA_F64>(mc, p, q); break;
    }
#endif
  } else if (q.dtype == AA_F16) {
    switch (g.nq) {
      case 2: rc = launch_q<2, AA_F16>(mc, p, q); break;
      case 3: rc = launch_q<3, AA_F16>(mc, p, q); break;
      case 4: rc = launch_q<4, AA_F16>(mc, p, q); break;
      case 5: rc = launch_q<5, AA_F16>(mc, p, q); break;
      case 7: rc = launch_q<7, AA_F16>(mc, p, q); break;
      case 9: rc = launch_q<9, AA_F16>(mc, p, q); break;
      default: rc = launch_q<11, AA_F16>(mc, p, q); break;
    }
  } else {
    switch (g.nq) {
      case 2: rc = launch_q<2, AA_BF16>(mc, p, q); break;
      case 3: rc = launch_q<3, AA_BF16>(mc, p, q); break;
      case 4: rc = launch_q<4, AA_BF16>(mc, p, q); break;
      case 5: rc = launch_q<5, AA_BF16>(mc, p, q); break;
      case 7: rc = launch_q<7, AA_BF16>(mc, p, q); break;
      case 9: rc = launch_q<9, AA_BF16>(mc, p, q); break;
      default: rc = launch_q<11, AA_BF16>(mc, p, q); break;
    }
  }
#if AA_F32_FAST
  if (rc == 1) *variant = q.dtype == AA_F32 ? "fused_f32_nchw_fast" : (q.dtype == AA_F16 ? "fused_f16_nchw_fast" : "fused_bf16_nchw_fast");
#else
  if (rc == 1 && g.cs != 1) *variant = "fused_f32_nhwc";
  else if (rc == 1 && q.dtype == AA_F64) *variant = "fused_f64_nchw";
  else if (rc == 1) *variant = q.dtype == AA_F32 ? "fused_f32_nchw" : (q.dtype == AA_F16 ? "fused_f16_nchw" : "fused_bf16_nchw");
#endif
  return rc;
}
