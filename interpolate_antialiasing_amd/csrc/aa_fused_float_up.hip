// aa_fused_float_up.hip — fused single-launch resample for fp32 NCHW planes whose height does not shrink (H <= oH):
// the gather form of the true adjoint (aa_resample_bwd = forward resample of grad_out with the transposed tables,
// BASELINE config 5: [.,3,196,320] -> [.,3,438,906]) and forward up-scaling (test.py's 906x438 -> 1200x1200 sizes).
//
// aa_fused_float.hip keeps its vertical pass in registers in scatter form, which needs every output row to complete
// once and in order — true when the height shrinks.  When it grows, every input row feeds several output rows and an
// output row needs only the last few input rows, so the vertical pass runs in GATHER form over a small register ring:
//   * one wave = one strip of <=64 output columns of one band of one (n, c) plane, strips of a band share a workgroup
//     without barriers (as in aa_fused_float.hip);
//   * input-row segments are staged into a private G-slot LDS ring by LDS-DMA, G-1 rows ahead;
//   * horizontal pass of an input row: taps from LDS, accumulated in the reference's order (tap 0 first, product and
//     sum rounded separately, taps beyond a lane's xsize not added at all); the result is pushed into a ring of the KR
//     most recent rows, in registers;
//   * an output row is the weighted sum of the last ysize ring entries with the wave-uniform weights of its table row
//     (scalar loads, prefetched one output row ahead), again tap 0 first: the same arithmetic, in the same order, as the
//     generic vertical pass;
//   * one coalesced 256-byte store per wave per output row.  The path is write-bound: 4.76 MB out per 0.75 MB in for the
//     backward of config A.
// Output stores and staging DMAs share the in-order vmcnt counter, and there are ~2 stores per DMA here, so the wave
// keeps the issue index of every slot's DMA and waits for exactly the operations older than it.

#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "aa_common.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void;

struct FusedF32UpParams {
  int H, W, oH, oW;
  int ksize_w, ksize_h;
  int ybands, nstrips, strips_per_block, strip_w;
  int nseg, seg_bytes;
  int in_mis;
  unsigned long long plane_in_bytes, plane_out_bytes, total_in_bytes, total_out_bytes;
  long long n_groups;  // (plane, band) groups = planes * ybands
};

__device__ inline void wait_vmcnt_up(int n) {  // rounding n DOWN only waits longer
  if (n >= 32) { asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); return; }
  if (n >= 24) { asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); return; }
  if (n >= 16) { asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); return; }
  if (n >= 12) { asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); return; }
  if (n >= 8) { asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); return; }
  if (n >= 6) { asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); return; }
  if (n >= 4) { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); return; }
  if (n >= 3) { asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); return; }
  if (n >= 2) { asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); return; }
  if (n >= 1) { asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); return; }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// TW: horizontal taps per lane; G: staged rows; KR: vertical taps kept in registers (>= max ysize of the H table).
template <int TW, int G, int KR>
__global__ void __launch_bounds__(512)
fused_f32_nchw_up_kernel(const float *__restrict__ in, float *__restrict__ out, const char *__restrict__ tab_w,
                         const char *__restrict__ tab_h, const FusedF32UpParams p) {
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
  const float *__restrict__ kh = (const float *)(tab_h + aa_table_w_off(p.oH));

  const int r_begin = __builtin_amdgcn_readfirstlane(ymin_h[oy0]);
  const int ylm = __builtin_amdgcn_readfirstlane(ymin_h[oy1 - 1]);
  const int yls = __builtin_amdgcn_readfirstlane(ysize_h[oy1 - 1]);
  const int r_stop = ylm + (yls > 1 ? yls : 1);  // one past the last input row this band reads

  // ---- per-lane horizontal-pass state (as aa_fused_float.hip) ---------------------------------------------------
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
  const int first_tap = lead;      // register index of the reference's tap 0
  const int last_tap = lead + xs;  // one past its last tap
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // from here on vmcnt counts staging DMAs and output stores only
  const int seg_first = __builtin_amdgcn_readfirstlane(start * 4);  // lane 0 holds the strip's leftmost window
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
  const bool dma_lane = lane < p.nseg;  // nseg <= 64 (checked on the host)
  const unsigned voff = (unsigned)lane * 16u;

  const unsigned long long out_off = (unsigned long long)plane * p.plane_out_bytes;
  unsigned long long out_rem = p.total_out_bytes - out_off;
  if (out_rem > 0xFFFFFFFFull) out_rem = 0xFFFFFFFFull;
  const __amdgpu_buffer_rsrc_t orsrc =
      __builtin_amdgcn_make_buffer_rsrc((void *)((uint8_t *)out + out_off), 0, (unsigned)out_rem, 0x00020000);
  const unsigned out_row_bytes = (unsigned)p.oW * 4u;
  const unsigned store_voff = (unsigned)(ox0 + lane) * 4u;

  // byte offset (from the descriptor base) of this strip's segment in input row `row`
  const unsigned a_base = (unsigned)(img_off - base_off) + (unsigned)seg_first;

  // ---- staging ring bookkeeping ------------------------------------------------------------------------------------
  int vm_issued = 0;  // VMEM instructions (DMAs + stores) this wave has issued since the wait above
  int idxv = 0;       // lane s: value of vm_issued right after the DMA that filled slot s
  auto dma = [&](int row, int slot) {
    const unsigned a_row = a_base + (unsigned)row * row_bytes;
    const int dst = lds_base + slot * p.seg_bytes;
    if (dma_lane) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void *)(lds + dst), 16, voff, a_row & ~15u, 0, 0);
    vm_issued++;
    idxv = (lane == slot) ? vm_issued : idxv;
  };
  int dma_next = r_begin;  // next input row to stage
  int slot_next = 0;       // its slot
  for (int i = 0; i < G; i++) {
    if (dma_next < r_stop) {
      dma(dma_next, slot_next);
      dma_next++;
      slot_next = slot_next + 1 == G ? 0 : slot_next + 1;
    }
  }

  float ring[KR];  // horizontal-pass results of input rows top-KR .. top-1
#pragma unroll
  for (int k = 0; k < KR; k++) ring[k] = 0.0f;
  int top = r_begin;  // next input row to run the horizontal pass on
  int slot_top = 0;

  // one input row: wait for its DMA, taps from LDS, reference-order accumulation, refill the slot, push the ring
  auto hpass_row = [&]() {
    const int my_idx = __builtin_amdgcn_readlane(idxv, slot_top);
    wait_vmcnt_up(vm_issued - my_idx);  // everything issued up to and including that DMA has completed
    const unsigned a_row = a_base + (unsigned)top * row_bytes;
    const unsigned sa = lane_lds + (unsigned)(slot_top * p.seg_bytes) + (a_row & 15u);  // multiple of 4
    const __attribute__((address_space(3))) float *src = (const __attribute__((address_space(3))) float *)(uintptr_t)sa;
    float d[TW];
#pragma unroll
    for (int j = 0; j < TW; j++) d[j] = src[j];
    float acc = 0.0f;
#pragma unroll
    for (int j = 0; j < TW; j++) {
      const float prod = d[j] * wreg[j];
      const float sum = acc + prod;
      acc = (j == first_tap) ? prod : ((j > first_tap && j < last_tap) ? sum : acc);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the taps are in registers: the slot may be refilled
    if (dma_next < r_stop) {
      dma(dma_next, slot_top);
      dma_next++;
    }
#pragma unroll
    for (int k = 0; k + 1 < KR; k++) ring[k] = ring[k + 1];
    ring[KR - 1] = acc;
    top++;
    slot_top = slot_top + 1 == G ? 0 : slot_top + 1;
  };

  // table row of an output: window start, length, weights (wave-uniform, loaded one output row ahead)
  struct VRow { int m; int s; float w[KR]; };
  auto load_vrow = [&](int oy) -> VRow {
    VRow v;
    const int o = oy < p.oH ? oy : p.oH - 1;
    v.m = __builtin_amdgcn_readfirstlane(ymin_h[o]);
    v.s = __builtin_amdgcn_readfirstlane(ysize_h[o]);
    const float *wr = kh + (size_t)o * p.ksize_h;
#pragma unroll
    for (int k = 0; k < KR; k++)  // rows are zero padded to ksize_h; never read past the row
      v.w[k] = (k < p.ksize_h) ? __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(wr[k]))) : 0.0f;
    return v;
  };

  VRow cur = load_vrow(oy0);
  for (int oy = oy0; oy < oy1; oy++) {
    const VRow nxt = load_vrow(oy + 1);
    const int s = cur.s > 1 ? cur.s : 1;
    const int need = cur.m + s;
    while (top < need) hpass_row();
    // the window's rows m .. m+s-1 sit at ring[base .. base+s-1], base = KR - (top - m)
    const int base = KR - (top - cur.m);
    float acc = 0.0f;
#pragma unroll
    for (int bb = 0; bb < KR; bb++) {
      if (base == bb) {  // wave-uniform: one of the KR static unrollings runs
#pragma unroll
        for (int k = 0; k + bb < KR; k++) {
          if (k >= s) break;  // taps beyond the window are not added at all
          const float prod = ring[bb + k] * cur.w[k];
          acc = (k == 0) ? prod : acc + prod;
        }
      }
    }
    if (active) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc), orsrc, store_voff, (unsigned)oy * out_row_bytes, 0);
    vm_issued++;
    cur = nxt;
  }
}

int pick_ybands_up(int64_t items_per_band, double slots, int taps_h, int64_t H, int64_t oH) {
  // a band re-reads ~taps_h input rows; input rows are the cheap side here, so only the round efficiency matters much
  const int64_t max_yb = oH / 16 > 1 ? oH / 16 : 1;
  int64_t ybands = 1;
  double best = 1e30;
  for (int64_t yb = 1; yb <= max_yb && yb <= 64; yb++) {
    const double rounds = (double)items_per_band * yb / slots;
    const double eff = rounds / ceil(rounds);
    const double halo = 1.0 + (double)(yb - 1) * taps_h / (double)(H + oH);
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

template <int TW, int G, int KR>
int launch_k(FusedF32UpParams p, const AAProblem &q, size_t lds) {
  auto kern = fused_f32_nchw_up_kernel<TW, G, KR>;
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
  p.ybands = pick_ybands_up(planes * sgroups, (double)aa_device_cu_count() * resident(spb), taps_h, q.H, q.oH);
  p.n_groups = planes * (int64_t)p.ybands;
  const int64_t grid = (p.n_groups + 7) / 8 * 8 * sgroups;
  if (grid > 0x7FFFFFFF) return 0;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * spb), lds_blk, q.stream,
                     (const float *)((const uint8_t *)q.in - p.in_mis), (float *)q.out, (const char *)q.aw.table_dev,
                     (const char *)q.ah.table_dev, p);
  AA_HIP_CHECK_LAUNCH();
  return 1;
}

template <int TW>
int launch_kr(int kr, const FusedF32UpParams &p, const AAProblem &q, size_t lds) {
  if (kr <= 2) return launch_k<TW, 8, 2>(p, q, lds);
  if (kr <= 4) return launch_k<TW, 8, 4>(p, q, lds);
  return launch_k<TW, 8, 8>(p, q, lds);
}

}  // namespace

bool aa_fused_float_nchw_up_applicable(int dtype, int layout, int64_t N, int64_t C, int64_t H, int64_t W, const aa_axis *ah,
                                       const aa_axis *aw) {
  if (dtype != AA_F32 || layout != AA_NCHW) return false;
  if (!ah || !aw || ah->kind != AA_TABLE_F32 || aw->kind != AA_TABLE_F32) return false;
  if (H > ah->out_size) return false;  // shrinking heights: aa_fused_float.hip
  const int taps_h = ah->max_taps > 0 ? ah->max_taps : ah->ksize;
  const int taps_w = aw->max_taps > 0 ? aw->max_taps : aw->ksize;
  if (taps_h > 8 || taps_w > 8) return false;
  const int tw = taps_w <= 2 ? 2 : (taps_w <= 4 ? 4 : 8);
  if (W < tw) return false;
  if ((uint64_t)H * W * 4 > 0xFFFFFFF0ull || (uint64_t)ah->out_size * aw->out_size * 4 > 0xFFFFFFF0ull) return false;
  const int span_px = aa_strip_span_px(*aw, tw);
  if (span_px < 0 || (span_px * 4 + 15 + 15) / 16 > 64) return false;  // one DMA instruction per staged row
  if (!aa_grid_fits(N * C * ((aw->out_size + 63) / 64 + 1))) return false;
  return true;
}

int aa_try_fused_float_nchw_up(const AAProblem &q, const char **variant) {
  if (!aa_fused_float_nchw_up_applicable(q.dtype, q.layout, q.N, q.C, q.H, q.W, &q.ah, &q.aw)) return 0;
  const int taps_h = q.ah.max_taps > 0 ? q.ah.max_taps : q.ah.ksize;
  const int taps_w = q.aw.max_taps > 0 ? q.aw.max_taps : q.aw.ksize;
  const int tw = taps_w <= 2 ? 2 : (taps_w <= 4 ? 4 : 8);
  if (((uintptr_t)q.out & 3) != 0 || ((uintptr_t)q.in & 3) != 0) return 0;

  FusedF32UpParams p;
  p.H = (int)q.H; p.W = (int)q.W; p.oH = (int)q.oH; p.oW = (int)q.oW;
  p.ksize_w = q.aw.ksize; p.ksize_h = q.ah.ksize;
  p.plane_in_bytes = (unsigned long long)q.H * q.W * 4;
  p.plane_out_bytes = (unsigned long long)q.oH * q.oW * 4;
  p.in_mis = (int)((uintptr_t)q.in & 15);
  p.total_in_bytes = p.plane_in_bytes * (unsigned long long)(q.N * q.C) + (unsigned long long)p.in_mis;
  p.total_out_bytes = p.plane_out_bytes * (unsigned long long)(q.N * q.C);
  p.nstrips = (int)((q.oW + 63) / 64);
  p.strip_w = (int)((q.oW + p.nstrips - 1) / p.nstrips);
  p.nstrips = (int)((q.oW + p.strip_w - 1) / p.strip_w);
  p.strips_per_block = p.nstrips <= 8 ? p.nstrips : 4;
  const int span_px = aa_strip_span_px(q.aw, tw);
  p.nseg = (span_px * 4 + 15 + 15) / 16;
  p.seg_bytes = p.nseg * 16;
  p.ybands = 1;
  const size_t lds = (size_t)8 * p.seg_bytes;

  int rc;
  if (tw == 2) rc = launch_kr<2>(taps_h, p, q, lds);
  else if (tw == 4) rc = launch_kr<4>(taps_h, p, q, lds);
  else rc = launch_kr<8>(taps_h, p, q, lds);
  if (rc == 1) *variant = "fused_f32_nchw_up";
  return rc;
}
