// aa_fused_float.hip — fused single-launch resample for fp32 NCHW tensors (BASELINE configs 0 and 2), reference arithmetic.
//
// Same wave-autonomous streaming design as aa_fused_u8_v3.hip, on floats:
//   * one wave = one strip of <=64 output columns of one band of one (n, c) plane; the strips of a band share a
//     workgroup only so that segment edges are served from L1/L2 (no barrier, no shared LDS);
//   * input-row segments (the floats the strip's 64 windows cover) are staged into a private G-slot LDS ring by LDS-DMA
//     (`buffer_load_dwordx4 ... lds`, range-checked), G-2 rows in flight behind a counted vmcnt;
//   * horizontal pass: one lane per output pixel reads its taps from LDS (floats are dword aligned: no realignment) and
//     accumulates exactly like the reference's inner loop (step_two_dot_two/aa_interpolation_impl.h:60-87): tap 0 first,
//     then taps 1..xsize-1 in order, product and sum rounded separately (this file is built with -ffp-contract=off);
//     taps at or beyond a lane's xsize are not added at all, so non-finite neighbours cannot leak in;
//   * vertical pass in registers, scatter form: row r's result is multiplied by the weights it has in the outputs it
//     feeds (scatter record of the H table) and added to their accumulators.  Rows arrive in increasing order, which
//     IS the reference's tap order (:29-58), so the sums round identically;
//   * a finished output row is one coalesced 256-byte store per wave.
// Roofline: HBM (fp32 config A: 5 514 576 B/image, config 2: 13 185 024 B/image; ~1.8-2.8 flop/B).

#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "aa_common.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void;

struct FusedF32Params {
  int H, W, oH, oW;
  int ksize_w, ksize_h;
  int ybands, nstrips, strips_per_block, strip_w;
  int nseg, seg_bytes;
  int sc_off;
  int in_mis;
  unsigned long long plane_in_bytes, plane_out_bytes, total_in_bytes, total_out_bytes;
  long long n_groups;  // (plane, band) groups = planes * ybands
};

__device__ inline void wait_vmcnt_f(int n) {  // rounding n DOWN only waits longer
  if (n >= 12) { asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); return; }
  if (n >= 8) { asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); return; }
  if (n >= 6) { asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); return; }
  if (n >= 4) { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); return; }
  if (n >= 3) { asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); return; }
  if (n >= 2) { asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); return; }
  if (n >= 1) { asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); return; }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// TW: taps per lane (>= max xsize of the W table); G: staged rows (even); MAXC: outputs one input row can feed.
template <int TW, int G, bool TWO_DMA, int MAXC>
__global__ void __launch_bounds__(512)
fused_f32_nchw_kernel(const float *__restrict__ in, float *__restrict__ out, const char *__restrict__ tab_w,
                      const char *__restrict__ tab_h, const FusedF32Params p) {
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
  const int32_t *__restrict__ xsize_w = (const int32_t *)(tab_w + aa_table_xsize_off(p.oW));
  const float *__restrict__ kw = (const float *)(tab_w + aa_table_w_off(p.oW));
  const int32_t *__restrict__ ymin_h = (const int32_t *)(tab_h + aa_table_xmin_off());
  const int32_t *__restrict__ ysize_h = (const int32_t *)(tab_h + aa_table_xsize_off(p.oH));
  const int32_t *__restrict__ sc_rec = (const int32_t *)(tab_h + p.sc_off);

  const int r_begin = __builtin_amdgcn_readfirstlane(ymin_h[oy0]);
  const int ylm = __builtin_amdgcn_readfirstlane(ymin_h[oy1 - 1]);
  const int yls = __builtin_amdgcn_readfirstlane(ysize_h[oy1 - 1]);
  const int r_stop = ylm + (yls > 1 ? yls : 1);
  const int n_rows = r_stop - r_begin;
  const int n_groups = (n_rows + G - 1) / G;

  // ---- per-lane horizontal-pass state ------------------------------------------------------------------------
  const bool active = lane < bw;
  const int ox = ox0 + (active ? lane : 0);
  const int xm = xmin_w[ox];
  int xs = xsize_w[ox];
  xs = xs > 1 ? xs : 1;  // tap 0 is unconditional in the reference (s2.2:68-73)
  int lead = xm + TW - p.W;  // right-align windows whose unused tail would leave the row
  lead = lead > 0 ? lead : 0;
  const int start = xm - lead;
  float wreg[TW];
#pragma unroll
  for (int j = 0; j < TW; j++) {
    const int src = j - lead;
    wreg[j] = (src >= 0 && src < xs && src < p.ksize_w) ? kw[(size_t)ox * p.ksize_w + src] : 0.0f;
  }
  const int first_tap = lead;        // register index of the reference's tap 0
  const int last_tap = lead + xs;    // one past its last tap
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const int seg_first = __builtin_amdgcn_readfirstlane(start * 4);
  const int c_l = start * 4 - seg_first;

  const unsigned long long img_off = (unsigned long long)p.in_mis + (unsigned long long)plane * p.plane_in_bytes;
  const unsigned long long base_off = img_off & ~15ull;
  unsigned long long remaining = p.total_in_bytes - base_off;
  if (remaining > 0xFFFFFFFFull) remaining = 0xFFFFFFFFull;
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void *)((const uint8_t *)in + base_off), 0, (unsigned)remaining, 0x00020000);
  const unsigned row_bytes = (unsigned)p.W * 4u;
  const int lds_base = wv * G * p.seg_bytes;
  const unsigned lane_lds = (unsigned)(lds_base + c_l);
  const bool dma_lane0 = lane < p.nseg;
  const bool dma_lane1 = lane + 64 < p.nseg;
  constexpr int dma_per_row = TWO_DMA ? 2 : 1;
  const unsigned voff = (unsigned)lane * 16u;

  const unsigned long long out_off = (unsigned long long)plane * p.plane_out_bytes;
  unsigned long long out_rem = p.total_out_bytes - out_off;
  if (out_rem > 0xFFFFFFFFull) out_rem = 0xFFFFFFFFull;
  const __amdgpu_buffer_rsrc_t orsrc =
      __builtin_amdgcn_make_buffer_rsrc((void *)((uint8_t *)out + out_off), 0, (unsigned)out_rem, 0x00020000);
  const unsigned out_row_bytes = (unsigned)p.oW * 4u;
  const unsigned store_voff = (unsigned)(ox0 + lane) * 4u;

  unsigned a = (unsigned)(img_off - base_off) + (unsigned)seg_first + (unsigned)r_begin * row_bytes;

  float A[MAXC];
#pragma unroll
  for (int k = 0; k < MAXC; k++) A[k] = 0.0f;
  int o_base = oy0;
  int done_row;
  {
    const int m = __builtin_amdgcn_readfirstlane(ymin_h[oy0]);
    const int s = __builtin_amdgcn_readfirstlane(ysize_h[oy0]);
    done_row = m + (s > 1 ? s : 1) - 1;
  }

  auto dma = [&](unsigned a_row, int slot) {
    const unsigned soff = a_row & ~15u;
    const int dst = lds_base + slot * p.seg_bytes;
    if (dma_lane0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void *)(lds + dst), 16, voff, soff, 0, 0);
    if constexpr (TWO_DMA) {
      if (dma_lane1)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void *)(lds + dst + 1024), 16, voff + 1024, soff, 0, 0);
    }
  };
  struct Scatter { int first; int cnt; float w[MAXC]; };
  auto load_scatter = [&](int r) -> Scatter {
    Scatter s;
    const int rr = r < p.H ? r : p.H - 1;
    const int32_t *rec = sc_rec + (size_t)rr * 8;
    s.first = __builtin_amdgcn_readfirstlane(rec[0]);
    s.cnt = __builtin_amdgcn_readfirstlane(rec[1]) & 0xFFFF;  // (high half: outputs completing at this row)
#pragma unroll
    for (int k = 0; k < MAXC; k++) s.w[k] = __int_as_float(__builtin_amdgcn_readfirstlane(rec[2 + k]));
    return s;
  };
  auto emit = [&](int oy) {
    if (active) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(A[0]), orsrc, store_voff, (unsigned)oy * out_row_bytes, 0);
#pragma unroll
    for (int k = 0; k + 1 < MAXC; k++) A[k] = A[k + 1];
    A[MAXC - 1] = 0.0f;
  };
  // one input row: taps from LDS, reference-order accumulation, scatter into the open outputs
  auto row_step = [&](unsigned a_row, int slot, int r, const Scatter &sc) {
    const unsigned sa = lane_lds + (unsigned)(slot * p.seg_bytes) + (a_row & 15u);  // multiple of 4
    const __attribute__((address_space(3))) float *src = (const __attribute__((address_space(3))) float *)(uintptr_t)sa;
    float d[TW];
#pragma unroll
    for (int j = 0; j < TW; j++) d[j] = src[j];
    // acc = t0*w0; acc += tj*wj for the lane's own taps only (registers [first_tap, last_tap))
    float acc = 0.0f;
#pragma unroll
    for (int j = 0; j < TW; j++) {
      const float prod = d[j] * wreg[j];
      const float sum = acc + prod;
      acc = (j == first_tap) ? prod : ((j > first_tap && j < last_tap) ? sum : acc);
    }
    const int idx0 = sc.first - o_base;
#pragma unroll
    for (int k = 0; k < MAXC; k++) {
      if (k >= sc.cnt) break;  // wave-uniform: only the outputs this row really belongs to
      const int slot_k = idx0 + k;
#pragma unroll
      for (int s = 0; s < MAXC; s++) {
        if (slot_k == s) {
          // an output's first tap lands on the initial 0: 0 + x == x exactly (the reference assigns tap 0), so the
          // running sums round identically from there on
          A[s] = A[s] + acc * sc.w[k];
        }
      }
    }
    while (r == done_row && o_base < oy1) {
      emit(o_base);
      o_base++;
      if (o_base < oy1) {
        const int m = __builtin_amdgcn_readfirstlane(ymin_h[o_base]);
        const int s = __builtin_amdgcn_readfirstlane(ysize_h[o_base]);
        done_row = m + (s > 1 ? s : 1) - 1;
      }
    }
  };

  for (int i = 0; i < G; i++)
    if (i < n_rows) dma(a + (unsigned)i * row_bytes, i);
  int r = r_begin;
  for (int g = 0; g < n_groups; g++) {
    const int x0 = g * G;
    if (x0 + 2 * G <= n_rows) {
#pragma unroll
      for (int i = 0; i < G; i++) {
        // row x must have landed: rows x+1 .. x+G-1 were issued after it
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(dma_per_row * (G - 1)) : "memory");
        const Scatter sc = load_scatter(r);
        row_step(a, i, r, sc);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the taps are in registers: the slot may be refilled
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
        wait_vmcnt_f(younger * dma_per_row);
        const Scatter sc = load_scatter(r);
        row_step(a, i, r, sc);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (x + G < n_rows) dma(a + (unsigned)G * row_bytes, i);
        a += row_bytes;
        r++;
      }
    }
  }
}

int pick_ybands_f(int64_t items_per_band, double slots, int taps_h, int64_t H, int64_t oH) {
  const int64_t max_yb = oH / 8 > 1 ? oH / 8 : 1;
  int64_t ybands = 1;
  double best = 1e30;
  for (int64_t yb = 1; yb <= max_yb && yb <= 64; yb++) {
    const double rounds = (double)items_per_band * yb / slots;
    const double eff = rounds / ceil(rounds);
    const double halo = 1.0 + (double)(yb - 1) * taps_h / (double)H;
    const double cost = halo / eff;
    if (cost < best - 1e-9) {
      best = cost;
      ybands = yb;
    }
  }
  if (const char *e = getenv("AA_FUSED_YBANDS")) {
    const int64_t v = atoll(e);
    if (v >= 1 && v <= max_yb) ybands = v;
  }
  return (int)ybands;
}

template <int TW, int G, bool TWO, int MAXC>
int launch_k(FusedF32Params p, const AAProblem &q, size_t lds) {
  auto kern = fused_f32_nchw_kernel<TW, G, TWO, MAXC>;
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
  if (spb > 1 && resident(1) > 0 && (resident(spb) < 0 || resident(1) > resident(spb) * spb)) spb = 1;
  if (resident(spb) < 0) return 0;
  p.strips_per_block = spb;
  const int sgroups = (p.nstrips + spb - 1) / spb;
  const size_t lds_blk = lds * spb;
  const int taps_h = q.ah.max_taps > 0 ? q.ah.max_taps : q.ah.ksize;
  const int64_t planes = q.N * q.C;
  p.ybands = pick_ybands_f(planes * sgroups, (double)aa_device_cu_count() * resident(spb), taps_h, q.H, q.oH);
  p.n_groups = planes * (int64_t)p.ybands;
  const int64_t grid = (p.n_groups + 7) / 8 * 8 * sgroups;
  if (grid > 0x7FFFFFFF) return 0;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * spb), lds_blk, q.stream, (const float *)((const uint8_t *)q.in - p.in_mis),
                     (float *)q.out, (const char *)q.aw.table_dev, (const char *)q.ah.table_dev, p);
  AA_HIP_CHECK_LAUNCH();
  return 1;
}

template <int TW, int G>
int launch_m(int maxc, const FusedF32Params &p, const AAProblem &q, size_t lds) {
  const bool two = p.nseg > 64;
  if (maxc <= 2) return two ? launch_k<TW, G, true, 2>(p, q, lds) : launch_k<TW, G, false, 2>(p, q, lds);
  if (maxc <= 3) return two ? launch_k<TW, G, true, 3>(p, q, lds) : launch_k<TW, G, false, 3>(p, q, lds);
  if (maxc <= 4) return two ? launch_k<TW, G, true, 4>(p, q, lds) : launch_k<TW, G, false, 4>(p, q, lds);
  return two ? launch_k<TW, G, true, 6>(p, q, lds) : launch_k<TW, G, false, 6>(p, q, lds);
}

int round_tw_f(int taps) {
  const int opts[] = {2, 4, 8, 12, 16, 24};
  for (int o : opts)
    if (taps <= o) return o;
  return 0;
}

}  // namespace

bool aa_fused_float_nchw_applicable(int dtype, int layout, int64_t N, int64_t C, int64_t H, int64_t W, const aa_axis *ah,
                                    const aa_axis *aw) {
  if (dtype != AA_F32 || layout != AA_NCHW) return false;
  if (!ah || !aw || ah->kind != AA_TABLE_F32 || aw->kind != AA_TABLE_F32) return false;
  if (ah->scatter_off <= 0 || ah->scatter_max <= 0 || ah->scatter_max > 6) return false;
  if (H < ah->out_size) return false;
  const int taps_w = aw->max_taps > 0 ? aw->max_taps : aw->ksize;
  const int tw = round_tw_f(taps_w);
  if (tw == 0 || W < tw) return false;
  if ((uint64_t)H * W * 4 > 0xFFFFFFF0ull) return false;
  const int span_px = aa_strip_span_px(*aw, tw);
  if (span_px < 0 || (span_px * 4 + 15 + 15) / 16 > 128) return false;
  if ((uint64_t)ah->out_size * aw->out_size * 4 > 0xFFFFFFF0ull) return false;
  if (!aa_grid_fits(N * C * ((aw->out_size + 63) / 64 + 1))) return false;
  return true;
}

int aa_try_fused_float_nchw(const AAProblem &q, const char **variant) {
  if (!aa_fused_float_nchw_applicable(q.dtype, q.layout, q.N, q.C, q.H, q.W, &q.ah, &q.aw)) return 0;
  if (((uintptr_t)q.out & 3) != 0 || ((uintptr_t)q.in & 3) != 0) return 0;
  const int taps_w = q.aw.max_taps > 0 ? q.aw.max_taps : q.aw.ksize;
  const int tw = round_tw_f(taps_w);

  FusedF32Params p;
  p.H = (int)q.H; p.W = (int)q.W; p.oH = (int)q.oH; p.oW = (int)q.oW;
  p.ksize_w = q.aw.ksize; p.ksize_h = q.ah.ksize;
  p.plane_in_bytes = (unsigned long long)q.H * q.W * 4;
  p.plane_out_bytes = (unsigned long long)q.oH * q.oW * 4;
  p.in_mis = (int)((uintptr_t)q.in & 15);
  p.total_in_bytes = p.plane_in_bytes * (unsigned long long)(q.N * q.C) + (unsigned long long)p.in_mis;
  p.total_out_bytes = p.plane_out_bytes * (unsigned long long)(q.N * q.C);
  p.sc_off = q.ah.scatter_off;
  p.nstrips = (int)((q.oW + 63) / 64);
  p.strip_w = (int)((q.oW + p.nstrips - 1) / p.nstrips);
  p.nstrips = (int)((q.oW + p.strip_w - 1) / p.strip_w);
  p.strips_per_block = p.nstrips <= 8 ? p.nstrips : 4;
  const int span_px = aa_strip_span_px(q.aw, tw);
  p.nseg = (span_px * 4 + 15 + 15) / 16;
  p.seg_bytes = p.nseg * 16;
  p.ybands = 1;

  int rc = 0;
  const int mc = q.ah.scatter_max;
  // G (rows in flight) shrinks as segments grow so that a workgroup's stage rings stay within 64 KiB
  const size_t lds8 = (size_t)8 * p.seg_bytes, lds4 = (size_t)4 * p.seg_bytes;
  const bool g8 = lds8 * p.strips_per_block <= 32 * 1024;
  if (tw <= 2) rc = g8 ? launch_m<2, 8>(mc, p, q, lds8) : launch_m<2, 4>(mc, p, q, lds4);
  else if (tw <= 4) rc = g8 ? launch_m<4, 8>(mc, p, q, lds8) : launch_m<4, 4>(mc, p, q, lds4);
  else if (tw <= 8) rc = g8 ? launch_m<8, 8>(mc, p, q, lds8) : launch_m<8, 4>(mc, p, q, lds4);
  else if (tw <= 12) rc = g8 ? launch_m<12, 8>(mc, p, q, lds8) : launch_m<12, 4>(mc, p, q, lds4);
  else if (tw <= 16) rc = g8 ? launch_m<16, 8>(mc, p, q, lds8) : launch_m<16, 4>(mc, p, q, lds4);
  else rc = g8 ? launch_m<24, 8>(mc, p, q, lds8) : launch_m<24, 4>(mc, p, q, lds4);
  if (rc == 1) *variant = "fused_f32_nchw";
  return rc;
}
