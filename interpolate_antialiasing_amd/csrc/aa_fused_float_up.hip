// aa_fused_float_up.hip — fused single-launch resample for fp32 NCHW planes whose height does not shrink (H <= oH):
// the gather form of the true adjoint (aa_resample_bwd = forward resample of grad_out with the transposed tables,
// BASELINE config 5: [.,3,196,320] -> [.,3,438,906]) and forward up-scaling (test.py's 906x438 -> 1200x1200 sizes).
//
// aa_fused_float.hip keeps its vertical pass in registers in scatter form, which needs every output row to complete
// once and in order — true when the height shrinks.  When it grows, every input row feeds several output rows and an
// output row needs only the last few input rows, so the vertical pass runs in GATHER form over a small register ring.
// This path is write-bound (4.76 MB out per 0.75 MB in for the backward of config A) and its arithmetic is tiny (2-4 taps
// each way), so what it costs is per-row overhead: round 1's kernel, one output column per lane, spent 48 scalar and 13
// vector instructions per 244-byte row piece (profiles/r02_pmc_bwd_before.json: 261 M SALU vs 69 M VALU per launch).
// Second design (round 2):
//   * a lane computes CPL = 4 (or 2, or 1) NEIGHBOURING output columns, so a wave covers a strip of 256 columns and a
//     finished output row leaves as ONE 1-KiB store instruction (16 bytes per lane): 4x fewer wave-rows, 4x less
//     per-row scalar work per byte;
//   * the 4 windows of a lane overlap (heights and widths grow here), so the lane reads their UNION once per input row
//     (U = taps + spread of 4 neighbouring window starts, measured by the table kernel: header.span4p1) and every output
//     accumulates over the union with its own weights; positions outside an output's own taps are skipped with scalar
//     lane masks exactly as in aa_fused_float.hip (never added with a zero weight: no non-finite neighbour leaks in, sums
//     are the reference's bit for bit: tap 0 first, product and sum rounded separately, -ffp-contract=off);
//   * input-row segments are staged into a private G-slot LDS ring by LDS-DMA, the source rounded down to a multiple of
//     four floats OF THE ROW (constant phase of the LDS image, see aa_fused_float.hip);
//   * the horizontal-pass results of the KR most recent input rows live in a register ring; an output row is emitted as
//     soon as its last input row has been pushed, so its window is always the ring's LAST ysize entries: the gather is
//     one of KR static unrollings selected by ysize alone; weights come from the table row by scalar loads issued one
//     output row ahead;
//   * output stores and staging DMAs share the in-order vmcnt counter and there are ~2 stores per DMA, so the wave keeps
//     the issue index of every slot's DMA and waits for exactly the operations older than it (a 4-level decision tree).

#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "aa_common.h"

#ifndef AA_UP_G
#define AA_UP_G 8  // staged input rows per wave
#endif
#ifndef AA_UP_DMA_AUX
#define AA_UP_DMA_AUX 0    // cache-policy bits of the staging DMA (developer knob; nt loads are slower: neighbouring strips and bands re-read rows)
#endif
#ifndef AA_UP_PACE
#define AA_UP_PACE 4  // rows between two pacing barriers of a workgroup whose strips share partial sectors (0: none)
#endif
#ifndef AA_UP_ABL
#define AA_UP_ABL 0  // developer ablations (wrong results): 1 no output stores, 2 no input (no DMA, no horizontal pass), 3 stores
                     // of every output row land on row 0 (stay in cache)
#endif

namespace {

typedef __attribute__((address_space(3))) void lds_void;

struct FusedF32UpParams {
  int H, W, oH, oW;
  int ksize_w, ksize_h;
  int ybands, nstrips, strips_per_block, strip_w;
  int nseg, seg_bytes;
  int pace_all;    // experiment: pacing barrier whatever the store mode
  int store_nt;    // output far larger than the caches: wide stores with the streaming (nt) policy.  This path writes 6x what it
                   // reads; with the default policy the written lines push the input rows (re-read by neighbouring strips and
                   // bands) out of L2 / the Infinity Cache and reads queue behind writes: measured -33 % (row pitch a multiple of
                   // 64 B) to -8 % (oW = 906) with nt (profiles/r02_headline_experiments.txt).  2 = rows that are not whole
                   // 64-byte sectors: nt for the whole sectors of a piece only (see the store), a further -5..-25 % for oW = 906
  int gather_off;  // gather section of the H table: one 32-byte record {ymin, ysize, w[6]} per output row
  unsigned long long plane_in_bytes, plane_out_bytes, total_in_bytes, total_out_bytes;
  long long n_groups;  // (plane, band) groups = planes * ybands
};

__device__ inline void wait_vmcnt_up(int n) {  // rounding n DOWN only waits longer; a decision tree, not a chain
  if (n >= 6) {
    if (n >= 16) {
      if (n >= 32) { asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); return; }
      if (n >= 24) { asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); return; }
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      return;
    }
    if (n >= 12) { asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); return; }
    if (n >= 8) { asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); return; }
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    return;
  }
  if (n >= 3) {
    if (n >= 4) { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); return; }
    asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    return;
  }
  if (n >= 2) { asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); return; }
  if (n >= 1) { asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); return; }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__device__ inline float select_by_mask_up(float a, float b, unsigned long long mask) {  // mask[lane] ? b : a
  float d;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(mask));
  return d;
}

// 16-bit planes (round 3, template parameter DT): halves are staged and read as halves and converted on the way into the fp32 arithmetic;
// results are rounded once, at the store (= half(reference_fp32(float(x))), like aa_fused_float.hip); a lane's CPL results leave as one
// 8-byte store.  The streaming store forms that cut pieces at sector boundaries are fp32-only: 16-bit outputs take the plain forms.
template <int DT> __device__ inline float up_elem_to_f32(unsigned short bits);
template <> __device__ inline float up_elem_to_f32<AA_F16>(unsigned short bits) {
  union { unsigned short u; _Float16 h; } c;
  c.u = bits;
  return (float)c.h;
}
template <> __device__ inline float up_elem_to_f32<AA_BF16>(unsigned short bits) { return __uint_as_float((unsigned)bits << 16); }
template <int DT> __device__ inline unsigned up_f32_to_elem(float a);
template <> __device__ inline unsigned up_f32_to_elem<AA_F16>(float a) {
  union { unsigned short u; _Float16 h; } c;
  c.h = (_Float16)a;
  return c.u;
}
template <> __device__ inline unsigned up_f32_to_elem<AA_BF16>(float a) {  // round to nearest even, NaN stays NaN
  const unsigned u = __float_as_uint(a);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x0040u;
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

// U: floats a lane reads per input row (the union of its CPL windows); G: staged rows; KR: vertical taps kept in registers
// (>= max ysize of the H table); CPL: neighbouring output columns per lane (a strip is 64 * CPL columns).
template <int U, int G, int KR, int CPL, int DT = AA_F32>
__global__ void __launch_bounds__(512)
fused_f32_nchw_up_kernel(const void *__restrict__ in, void *__restrict__ out, const char *__restrict__ tab_w,
                         const char *__restrict__ tab_h, const FusedF32UpParams p) {
  constexpr int ES = DT == AA_F32 ? 4 : 2;  // element bytes
  constexpr int EPP = 16 / ES;              // elements per staged 16-byte piece
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
  // store_nt == 3 (sector-aligned pieces, see the store): strips advance by 240 columns and start 16 columns early, so a wave
  // holds every column its row-dependent 960-byte piece can need; columns outside the row are duplicates and never stored
  const bool aln = p.store_nt == 3;
  const int ox0 = aln ? strip * 240 - 16 : strip * p.strip_w;
  const int bw = aln ? 256 : min(p.strip_w, p.oW - ox0);
  const int oy0 = (int)((long long)yb * p.oH / p.ybands);
  const int oy1 = (int)((long long)(yb + 1) * p.oH / p.ybands);

  const int32_t *__restrict__ xmin_w = (const int32_t *)(tab_w + aa_table_xmin_off());
  const int32_t *__restrict__ xsize_w = (const int32_t *)(tab_w + aa_table_xsize_off(p.oW));
  const float *__restrict__ kw = (const float *)(tab_w + aa_table_w_off(p.oW));
  const int32_t *__restrict__ ymin_h = (const int32_t *)(tab_h + aa_table_xmin_off());
  const int32_t *__restrict__ ysize_h = (const int32_t *)(tab_h + aa_table_xsize_off(p.oH));

  const int r_begin = __builtin_amdgcn_readfirstlane(ymin_h[oy0]);
  const int ylm = __builtin_amdgcn_readfirstlane(ymin_h[oy1 - 1]);
  const int yls = __builtin_amdgcn_readfirstlane(ysize_h[oy1 - 1]);
  const int r_stop = ylm + (yls > 1 ? yls : 1);  // one past the last input row this band reads

  // ---- per-lane horizontal-pass state: CPL outputs sharing one union window of U floats ------------------------------
  const int col0 = lane * CPL;                 // first column of the lane inside the strip
  const bool any_active = col0 < bw;
  int oxb = ox0 + (any_active ? col0 : 0);  // (lanes beyond the strip duplicate lane 0 and never store)
  oxb = oxb < 0 ? 0 : (oxb < p.oW ? oxb : p.oW - 1);
  int ustart;  // row position of the union window's first float
  {
    const int xm0 = xmin_w[oxb];
    int hi = p.W - U;  // right-align a union that would leave the row
    hi = hi > 0 ? hi : 0;
    ustart = xm0 < hi ? xm0 : hi;
  }
  float wreg[CPL][U];
  unsigned long long inwin[CPL][U];  // lane masks (scalar registers): union position q is one of output e's own taps
#pragma unroll
  for (int e = 0; e < CPL; e++) {
    int oxe = ox0 + col0 + e;  // (columns outside the row compute a duplicate, never stored)
    oxe = (!any_active || oxe < 0) ? oxb : (oxe < p.oW ? oxe : p.oW - 1);
    const int xm = xmin_w[oxe];
    int xs = xsize_w[oxe];
    xs = xs > 1 ? xs : 1;  // tap 0 is unconditional in the reference (s2.2:68-73)
    const int t0 = xm - ustart;  // union position of the reference's tap 0 (>= 0: window starts do not decrease)
#pragma unroll
    for (int q = 0; q < U; q++) {
      const int j = q - t0;
      const bool mine = j >= 0 && j < xs;
      wreg[e][q] = (mine && j < p.ksize_w) ? kw[(size_t)oxe * p.ksize_w + j] : 0.0f;
      inwin[e][q] = __ballot(mine);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // from here on vmcnt counts staging DMAs and output stores only
  const int seg0 = __builtin_amdgcn_readfirstlane(ustart) & ~(EPP - 1);  // lane 0 holds the strip's leftmost window
  const unsigned lane_lds = (unsigned)(wv * G * p.seg_bytes + (ustart - seg0) * ES);

  const unsigned long long plane_off = (unsigned long long)plane * p.plane_in_bytes;
  unsigned long long remaining = p.total_in_bytes - plane_off;
  if (remaining > 0xFFFFFFFCull) remaining = 0xFFFFFFFCull;
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void *)((const uint8_t *)in + plane_off), 0, (unsigned)remaining, 0x00020000);
  const unsigned row_bytes = (unsigned)p.W * (unsigned)ES;
  const int lds_base = wv * G * p.seg_bytes;
  const bool dma_lane = lane < p.nseg;  // nseg <= 64 (checked on the host)
  const unsigned voff = (unsigned)lane * 16u;

  const unsigned long long out_off = (unsigned long long)plane * p.plane_out_bytes;
  unsigned long long out_rem = p.total_out_bytes - out_off;
  if (out_rem > 0xFFFFFFFFull) out_rem = 0xFFFFFFFFull;
  const __amdgpu_buffer_rsrc_t orsrc =
      __builtin_amdgcn_make_buffer_rsrc((void *)((uint8_t *)out + out_off), 0, (unsigned)out_rem, 0x00020000);
  const unsigned out_row_bytes = (unsigned)p.oW * (unsigned)ES;
  const unsigned store_voff = (unsigned)(ox0 + col0) * (unsigned)ES;
  const bool full_lane = col0 + CPL <= bw;  // all CPL columns of the lane exist: one wide store
  const unsigned phase0 = (unsigned)(((unsigned long long)(uintptr_t)out + out_off + (unsigned long long)ox0 * 4u) & 127u);

  const unsigned a_base = (unsigned)seg0 * (unsigned)ES;  // byte offset (from the plane) of the strip's segment in row 0
  // sector-aligned pieces: this wave's 1088-byte staging area behind the workgroup's stage rings; float phase of the plane in
  // the 64-byte sector grid
  const unsigned aln_lds = (unsigned)(p.strips_per_block * G * p.seg_bytes + wv * 1088);
  const unsigned phase0f = (unsigned)((((unsigned long long)(uintptr_t)out + out_off) >> 2) & 15u);

  // ---- staging ring bookkeeping ------------------------------------------------------------------------------------
  int vm_issued = 0;  // VMEM instructions (DMAs + stores) this wave has issued since the wait above
  int idxv = 0;       // lane s: value of vm_issued right after the DMA that filled slot s
  auto dma = [&](int row, int slot) {
    const int dst = lds_base + slot * p.seg_bytes;
    if (dma_lane)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void *)(lds + dst), 16, voff, a_base + (unsigned)row * row_bytes, 0, AA_UP_DMA_AUX);
    vm_issued++;
    idxv = (lane == slot) ? vm_issued : idxv;
  };
  int dma_next = r_begin;  // next input row to stage
  int slot_next = 0;       // its slot
  for (int i = 0; i < G; i++) {
    if (AA_UP_ABL != 2 && dma_next < r_stop) {
      dma(dma_next, slot_next);
      dma_next++;
      slot_next = slot_next + 1 == G ? 0 : slot_next + 1;
    }
  }

  float ring[KR][CPL];  // horizontal-pass results of input rows top-KR .. top-1
#pragma unroll
  for (int k2 = 0; k2 < KR; k2++)
#pragma unroll
    for (int e = 0; e < CPL; e++) ring[k2][e] = 0.0f;
  int top = r_begin;  // next input row to run the horizontal pass on
  int slot_top = 0;

  // one input row: wait for its DMA, union window from LDS, reference-order accumulation per output, refill, push
  // 16-bit elements, odd W: the dword holding the tensor's final element straddles the end of the tensor and is refused by the range
  // check (see aa_fused_float.hip: nothing may be read past a tensor); lane 0 fetches that element on its own into the staged row
  const int fix_row = (ES == 2 && (p.W & 1) && (long long)plane + 1 == p.n_groups / p.ybands) ? p.H - 1 : -1;
  auto hpass_row = [&]() {
    const int my_idx = __builtin_amdgcn_readlane(idxv, slot_top);
    wait_vmcnt_up(vm_issued - my_idx);  // everything issued up to and including that DMA has completed
    if (ES == 2 && top == fix_row) {
      const int pos = p.W - 1 - seg0;  // (even: seg0 is a multiple of 8, W is odd)
      if (pos >= 0 && pos < p.nseg * EPP) {
        if (lane == 0) {
          const unsigned short v = *(const unsigned short *)((const uint8_t *)in + plane_off + (unsigned long long)(p.H - 1) * row_bytes +
                                                             (unsigned long long)(p.W - 1) * 2u);
          *(unsigned *)(lds + lds_base + slot_top * p.seg_bytes + pos * 2) = (unsigned)v;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      }
    }
    float d[U];
    if constexpr (DT == AA_F32) {
      const __attribute__((address_space(3))) float *src =
          (const __attribute__((address_space(3))) float *)(uintptr_t)(lane_lds + (unsigned)(slot_top * p.seg_bytes));
#pragma unroll
      for (int q = 0; q < U; q++) d[q] = src[q];
    } else {
      const __attribute__((address_space(3))) unsigned short *src =
          (const __attribute__((address_space(3))) unsigned short *)(uintptr_t)(lane_lds + (unsigned)(slot_top * p.seg_bytes));
#pragma unroll
      for (int q = 0; q < U; q++) d[q] = up_elem_to_f32<DT>(src[q]);
    }
    float acc[CPL];
#pragma unroll
    for (int e = 0; e < CPL; e++) {
      acc[e] = -0.0f;  // (-0) + x == x exactly: the first own tap is an assignment
#pragma unroll
      for (int q = 0; q < U; q++) {
        const float prod = d[q] * wreg[e][q];
        const float sum = acc[e] + prod;
        acc[e] = select_by_mask_up(acc[e], sum, inwin[e][q]);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the window is in registers: the slot may be refilled
    if (dma_next < r_stop) {
      dma(dma_next, slot_top);
      dma_next++;
    }
#pragma unroll
    for (int k2 = 0; k2 + 1 < KR; k2++)
#pragma unroll
      for (int e = 0; e < CPL; e++) ring[k2][e] = ring[k2 + 1][e];
#pragma unroll
    for (int e = 0; e < CPL; e++) ring[KR - 1][e] = acc[e];
    top++;
    slot_top = slot_top + 1 == G ? 0 : slot_top + 1;
  };

  // table row of an output: window start, length, weights — ONE scalar load of the row's gather record (wave-uniform,
  // issued one output row ahead)
  struct VRow { int m; int s; float w[KR]; };
  const int32_t *__restrict__ grec = (const int32_t *)(tab_h + p.gather_off);
  auto load_vrow = [&](int oy) -> VRow {
    VRow v;
    const int o = oy < p.oH ? oy : p.oH - 1;
    const int32_t *rec = (const int32_t *)((const char *)grec + (unsigned)o * 32u);
    v.m = __builtin_amdgcn_readfirstlane(rec[0]);
    v.s = __builtin_amdgcn_readfirstlane(rec[1]);
#pragma unroll
    for (int k2 = 0; k2 < KR; k2++) v.w[k2] = __int_as_float(__builtin_amdgcn_readfirstlane(rec[2 + k2]));
    return v;
  };

  VRow cur = load_vrow(oy0);
  // Rows that are not whole sectors: the strips of a row (the waves of this workgroup) are kept within AA_UP_PACE rows of each
  // other, so that the two halves of a shared end sector reach the L2 close together and leave it as one full write
  // (906-wide gradients: 0.345 -> 0.312 ms).  Only workgroups all of whose waves are alive take the barrier.
  const bool pace = AA_UP_PACE != 0 && (p.store_nt == 2 || p.pace_all) && p.strips_per_block > 1 &&
                    ((k % sgroups) + 1) * p.strips_per_block <= p.nstrips;  // every wave of this workgroup is alive
  for (int oy = oy0; oy < oy1; oy++) {
    if (pace && ((oy - oy0) % AA_UP_PACE) == 0) __builtin_amdgcn_s_barrier();
    const VRow nxt = load_vrow(oy + 1);
    int s = cur.s > 1 ? cur.s : 1;
    s = s < KR ? s : KR;
    const int need = cur.m + s;
    if (AA_UP_ABL != 2) { while (top < need) hpass_row(); }
    // Windows end at non-decreasing rows and the ring was advanced exactly to this one's end (top == need), so the
    // window's rows m .. m+s-1 are the ring's LAST s entries: ring[KR-s .. KR-1].
    float res[CPL];
#pragma unroll
    for (int e = 0; e < CPL; e++) res[e] = 0.0f;
#pragma unroll
    for (int ss = 1; ss <= KR; ss++) {
      if (s == ss) {  // wave-uniform: one of the KR static unrollings runs; taps beyond the window are not added at all
#pragma unroll
        for (int e = 0; e < CPL; e++) {
          float acc = ring[KR - ss][e] * cur.w[0];
#pragma unroll
          for (int k2 = 1; k2 < ss; k2++) acc = acc + ring[KR - ss + k2][e] * cur.w[k2];
          res[e] = acc;
        }
      }
    }
    const unsigned soff = AA_UP_ABL == 3 ? 0u : (unsigned)oy * out_row_bytes;
    // vm_issued may only count instructions that are certainly issued (an all-lanes-off store is branched around): every
    // count below is guarded by a wave-uniform condition under which lane 0 or the ragged lane really stores
    if constexpr (DT != AA_F32) {  // 16-bit planes: one store of the lane's CPL halves (plain or streaming), ragged columns one by one
      unsigned hb[CPL];
#pragma unroll
      for (int e = 0; e < CPL; e++) hb[e] = up_f32_to_elem<DT>(res[e]);
      if constexpr (CPL == 4) {
        typedef unsigned u32x2h __attribute__((ext_vector_type(2)));
        const u32x2h t = {hb[0] | (hb[1] << 16), hb[2] | (hb[3] << 16)};
        if (bw >= CPL) {
          if (full_lane) {
            if (p.store_nt != 0) __builtin_amdgcn_raw_buffer_store_b64(t, orsrc, store_voff, soff, 2);
            else __builtin_amdgcn_raw_buffer_store_b64(t, orsrc, store_voff, soff, 0);
          }
          vm_issued++;
        }
      } else if constexpr (CPL == 2) {
        if (bw >= CPL) {
          if (full_lane) __builtin_amdgcn_raw_buffer_store_b32(hb[0] | (hb[1] << 16), orsrc, store_voff, soff, 0);
          vm_issued++;
        }
      } else {
        if (any_active) __builtin_amdgcn_raw_buffer_store_b16((unsigned short)hb[0], orsrc, store_voff, soff, 0);
        vm_issued++;
      }
      if constexpr (CPL > 1) {
        const int ragged = bw % CPL;
#pragma unroll
        for (int e = 0; e < CPL - 1; e++) {
          if (__builtin_expect(ragged != 0, 0) && e < ragged) {
            if (col0 == bw - ragged) __builtin_amdgcn_raw_buffer_store_b16((unsigned short)hb[e], orsrc, store_voff + 2u * e, soff, 0);
            vm_issued++;
          }
        }
      }
      cur = nxt;
      continue;
    }
    if constexpr (CPL == 4) {
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      const u32x4 t = {__float_as_uint(res[0]), __float_as_uint(res[1]), __float_as_uint(res[2]), __float_as_uint(res[3])};
      if (aln) {
        // Rows that are not whole 64-byte sectors, even widths: the piece a strip stores is cut at the sector boundaries of
        // THIS row — columns [240 k - d, 240 k + 240 - d), d = the row's phase in floats (even, 0 .. 14) — so every streamed
        // store covers whole sectors and no sector is shared between strips (only the row's two ends stay partial).  The wave
        // holds columns 240 k - 16 .. 240 k + 239; a pass through LDS moves each lane's four results to the lane that owns
        // their absolute 16-byte slot (8-byte aligned reads: d is even).
        __attribute__((address_space(3))) u32x4 *stw = (__attribute__((address_space(3))) u32x4 *)(uintptr_t)(aln_lds + (unsigned)lane * 16u);
        *stw = t;
        const unsigned d = (phase0f + (unsigned)oy * (unsigned)p.oW) & 15u;
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        const unsigned rd = aln_lds + (16u - d) * 4u + (lane < 60 ? (unsigned)lane * 16u : 0u);
        const __attribute__((address_space(3))) u32x2 *ldp = (const __attribute__((address_space(3))) u32x2 *)(uintptr_t)rd;
        const u32x2 v0 = ldp[0], v1 = ldp[1];
        const int colj = strip * 240 - (int)d + lane * 4;  // first column of this lane's slot
        const bool mine = lane < 60;
        const bool lo_ok = mine && colj >= 0 && colj + 1 < p.oW;
        const bool hi_ok = mine && colj + 2 >= 0 && colj + 3 < p.oW;
        const bool full = lo_ok && hi_ok;
        const unsigned cvoff = (unsigned)(colj * 4);
        if (__ballot(full) != 0ull) {  // (wave-uniform: the store is certainly issued and may be counted)
          const u32x4 tv = {v0.x, v0.y, v1.x, v1.y};
          if (full) __builtin_amdgcn_raw_buffer_store_b128(tv, orsrc, cvoff, soff, 2);
          vm_issued++;
        }
        // the row's first / last slot can hold two columns only (never counted: waiting for more is safe)
        if (lo_ok && !hi_ok) __builtin_amdgcn_raw_buffer_store_b64(v0, orsrc, cvoff, soff, 2);
        if (hi_ok && !lo_ok) __builtin_amdgcn_raw_buffer_store_b64(v1, orsrc, cvoff + 8u, soff, 2);
      } else if (p.store_nt == 2 && bw >= CPL) {
        // Output rows that are not whole 64-byte sectors (oW = 906: 3624 B): the sectors a piece shares with its
        // neighbouring strip are written half by each, and a streamed (nt) half-sector write costs a read-modify-write
        // at the memory.  So only the lanes whose 16 bytes lie in whole sectors of the piece stream; the lanes of its
        // two end sectors store with the default policy and the L2 merges the halves.  Measured on [256,3,438,906]
        // gradients: 0.395-0.44 ms (all nt) -> 0.32-0.376 ms depending on the box / buffer placement; aligned widths
        // keep the single store (the split costs them 3 %).
        const unsigned P = (phase0 + soff) & 63u;
        const unsigned nl_full = (unsigned)bw / CPL;
        const unsigned E = P + 16u * nl_full;
        const unsigned IA = (P + 63u) & ~63u, IB = E & ~63u;
        const unsigned lo = (IA - P + 15u) >> 4;
        const unsigned hi = IB > IA ? (IB - P) >> 4 : lo;
        const bool interior = (unsigned)lane >= lo && (unsigned)lane < hi;
        if (hi > lo) {  // (lane lo is a full, interior lane: the store is issued)
          if (full_lane && interior) __builtin_amdgcn_raw_buffer_store_b128(t, orsrc, store_voff, soff, 2);
          vm_issued++;
        }
        if (lo > 0 || hi < nl_full) {  // (lane 0 or lane hi is a full end lane)
          if (full_lane && !interior) __builtin_amdgcn_raw_buffer_store_b128(t, orsrc, store_voff, soff, 0);
          vm_issued++;
        }
      } else if (bw >= CPL) {
        if (full_lane && (AA_UP_ABL != 1 || t.x == 0x12345678u)) {
          if (p.store_nt != 0) __builtin_amdgcn_raw_buffer_store_b128(t, orsrc, store_voff, soff, 2);
          else __builtin_amdgcn_raw_buffer_store_b128(t, orsrc, store_voff, soff, 0);
        }
        if (AA_UP_ABL != 1) vm_issued++;
      }
    } else if constexpr (CPL == 2) {
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
      const u32x2 t = {__float_as_uint(res[0]), __float_as_uint(res[1])};
      if (bw >= CPL) {
        if (full_lane) __builtin_amdgcn_raw_buffer_store_b64(t, orsrc, store_voff, soff, 0);
        vm_issued++;
      }
    } else {
      if (any_active) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(res[0]), orsrc, store_voff, soff, 0);
      vm_issued++;
    }
    if constexpr (CPL > 1) {
      const int ragged = bw % CPL;  // (wave-uniform) columns of the strip's last, partial lane: stored one by one
#pragma unroll
      for (int e = 0; e < CPL - 1; e++) {
        if (__builtin_expect(ragged != 0, 0) && e < ragged) {
          if (col0 == bw - ragged) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(res[e]), orsrc, store_voff + 4u * e, soff, 0);
          vm_issued++;
        }
      }
    }
    cur = nxt;
  }
}

int pick_ybands_up(int64_t items_per_band, int waves_per_item, double slots, int taps_h, int64_t H, int64_t oH) {
  // a band re-reads ~taps_h input rows; input rows are the cheap side here, so only the round efficiency matters much
  const int64_t max_yb = oH / 16 > 1 ? oH / 16 : 1;
  int64_t ybands = 1;
  double best = 1e30;
  // Single-strip workgroups whose whole problem fits the chip in about one round: this write-bound kernel is fastest with
  // about 20 waves per CU in flight, not with every slot filled over several rounds (measured on one box,
  // [64,3,438,906] -> 1200x1200: 7.5 waves per CU 0.32 ms, 15 0.304-0.319, 19 0.297-0.313, 24 x 3 rounds 0.327-0.344).  Workgroups
  // of several strips (906-wide gradients: 4 strips with the pacing barrier) measured best with the round model below.
  const double target_waves = 20.0 * aa_device_cu_count();
  const double waves_per_band = (double)items_per_band * waves_per_item;
  if (waves_per_item == 1 && waves_per_band <= target_waves * 1.25 && !aa_knob("AA_FUSED_YBANDS")) {
    int64_t yb = (int64_t)(target_waves / (waves_per_band > 0 ? waves_per_band : 1) + 0.5);
    yb = yb < 1 ? 1 : (yb > max_yb ? max_yb : yb);
    if (yb > 64) yb = 64;
    return (int)yb;
  }
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
  if (const char *e = aa_knob("AA_FUSED_YBANDS")) {
    const int64_t v = atoll(e);
    if (v >= 1 && v <= max_yb) ybands = v;
  }
  return (int)ybands;
}

template <int U, int G, int KR, int CPL, int DT>
int launch_k(FusedF32UpParams p, const AAProblem &q, size_t lds) {
  auto kern = fused_f32_nchw_up_kernel<U, G, KR, CPL, DT>;
  auto resident = [&](int s) {  // workgroups of s strips a CU holds (-1: their rings do not fit a workgroup's LDS)
    if (lds * s > 64 * 1024) return -1;  // (never for s == 1: a strip's ring is at most 8 KiB)
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
  if (const char *e = aa_knob("AA_UP_SPB")) {  // experiment knob
    const int v = atoi(e);
    if (v >= 1 && v <= 8) spb = v;
  }
  if (spb > 1 && (resident(spb) < 0 || resident(1) > resident(spb) * spb)) spb = 1;
  p.strips_per_block = spb;
  const int sgroups = (p.nstrips + spb - 1) / spb;
  const size_t lds_blk = lds * spb;
  const int taps_h = q.ah.max_taps > 0 ? q.ah.max_taps : q.ah.ksize;
  const int64_t planes = q.N * q.C;
  p.ybands = pick_ybands_up(planes * sgroups, spb, (double)aa_device_cu_count() * resident(spb), taps_h, q.H, q.oH);
  p.n_groups = planes * (int64_t)p.ybands;
  const int64_t grid = (p.n_groups + 7) / 8 * 8 * sgroups;
  if (grid > 0x7FFFFFFF) return 0;
  if (aa_knob("AA_UP_DEBUG"))
    fprintf(stderr, "up: U=%d KR=%d CPL=%d nstrips=%d spb=%d resident=%d ybands=%d groups=%lld grid=%lld lds=%zu nt=%d\n", U, KR, CPL, p.nstrips, spb,
            resident(spb), p.ybands, (long long)p.n_groups, (long long)grid, lds_blk, p.store_nt);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * spb), lds_blk, q.stream, (const void *)q.in, (void *)q.out,
                     (const char *)q.aw.table_dev, (const char *)q.ah.table_dev, p);
  AA_HIP_CHECK_LAUNCH();
  return 1;
}

template <int U, int CPL, int DT>
int launch_kr(int kr, const FusedF32UpParams &p, const AAProblem &q, size_t lds) {
  if (kr <= 2) return launch_k<U, AA_UP_G, 2, CPL, DT>(p, q, lds);
  if (kr <= 4) return launch_k<U, AA_UP_G, 4, CPL, DT>(p, q, lds);
  return launch_k<U, AA_UP_G, 6, CPL, DT>(p, q, lds);
}

// Columns per lane and union width: the widest CPL whose CPL * U lane masks fit the scalar registers (<= 20) and whose
// strip segment is one DMA instruction (<= 64 pieces); U = taps + spread of CPL neighbouring window starts.
struct UpGeometry { int cpl, u, nstrips, strip_w, nseg; };

bool up_geometry(int64_t W, const aa_axis &aw, UpGeometry *g, int es = 4) {
  const int taps_w = aw.max_taps > 0 ? aw.max_taps : aw.ksize;
  if (taps_w > 8 || aw.span64p1 <= 0 || aw.span4p1 <= 0) return false;
  const int64_t oW = aw.out_size;
  const int cands[3] = {4, 2, 1};
  for (int ci = 0; ci < 3; ci++) {
    const int cpl = cands[ci];
    if (cpl > 1 && oW < 64 * cpl) continue;  // narrow outputs: keep the lanes busy
    // spread of cpl neighbouring window starts (span4p1 - 1 covers 4; 2 neighbours spread at most as much)
    const int spread = cpl == 1 ? 0 : aw.span4p1 - 1;
    int u = taps_w + spread;
    const int opts[6] = {2, 3, 4, 5, 6, 8};
    int uu = 0;
    for (int o : opts)
      if (u <= o) { uu = o; break; }
    if (uu == 0 || uu * cpl > 20 || W < uu) continue;
    // floats a strip of 64 * cpl outputs covers: cpl * spread of 64 starts (+3: rounded down to a multiple of 4) + union
    const int span = cpl * (aw.span64p1 - 1) + cpl + uu + (16 / es - 1);  // (+: the segment start rounded down to a 16-byte piece of the row)
    const int nseg = (span * es + 15) / 16 + 1;
    if (nseg > 64) continue;
    g->cpl = cpl; g->u = uu; g->nseg = nseg;
    g->strip_w = 64 * cpl;
    g->nstrips = (int)((oW + g->strip_w - 1) / g->strip_w);
    return true;
  }
  return false;
}

}  // namespace

bool aa_fused_float_nchw_up_applicable(int dtype, int layout, int64_t N, int64_t C, int64_t H, int64_t W, const aa_axis *ah,
                                       const aa_axis *aw) {
  if ((dtype != AA_F32 && dtype != AA_F16 && dtype != AA_BF16) || layout != AA_NCHW) return false;
  if (!ah || !aw || ah->kind != AA_TABLE_F32 || aw->kind != AA_TABLE_F32) return false;
  const int es = dtype == AA_F32 ? 4 : 2;
  if (H > ah->out_size) return false;  // shrinking heights: aa_fused_float.hip
  const int taps_h = ah->max_taps > 0 ? ah->max_taps : ah->ksize;
  if (taps_h > 6 || ah->gather_off <= 0) return false;  // (a gather record holds 6 weights)
  UpGeometry g;
  if (!up_geometry(W, *aw, &g, es)) return false;
  if ((uint64_t)H * W * 4 > 0xFFFFFFF0ull || (uint64_t)ah->out_size * aw->out_size * 4 > 0xFFFFFFF0ull) return false;
  if (!aa_grid_fits(N * C * g.nstrips)) return false;
  return true;
}

int aa_try_fused_float_nchw_up(const AAProblem &q, const char **variant) {
  if (!aa_fused_float_nchw_up_applicable(q.dtype, q.layout, q.N, q.C, q.H, q.W, &q.ah, &q.aw)) return 0;
  const int taps_h = q.ah.max_taps > 0 ? q.ah.max_taps : q.ah.ksize;
  const int es = q.dtype == AA_F32 ? 4 : 2;
  if (((uintptr_t)q.out & (es - 1)) != 0 || ((uintptr_t)q.in & (es - 1)) != 0) return 0;
  UpGeometry g;
  up_geometry(q.W, q.aw, &g, es);

  FusedF32UpParams p;
  p.H = (int)q.H; p.W = (int)q.W; p.oH = (int)q.oH; p.oW = (int)q.oW;
  p.ksize_w = q.aw.ksize; p.ksize_h = q.ah.ksize;
  p.plane_in_bytes = (unsigned long long)q.H * q.W * es;
  p.plane_out_bytes = (unsigned long long)q.oH * q.oW * es;
  p.total_in_bytes = p.plane_in_bytes * (unsigned long long)(q.N * q.C);
  p.total_out_bytes = p.plane_out_bytes * (unsigned long long)(q.N * q.C);
  p.nstrips = g.nstrips;
  p.strip_w = g.strip_w;
  p.strips_per_block = p.nstrips <= 8 ? p.nstrips : 4;
  p.nseg = g.nseg;
  p.seg_bytes = p.nseg * 16;
  p.gather_off = q.ah.gather_off;
  p.store_nt = g_aa_store_form < 0 ? (p.total_out_bytes > (64ull << 20) ? 1 : 0) : (g_aa_store_form ? 1 : 0);  // (aa_set_store_form: tests of the
                                                                                                               // streaming forms at small sizes)
  // rows or planes that are not whole 64-byte sectors: stream only the whole sectors of each piece (see the store)
  if (es == 4 && p.store_nt && g.cpl == 4 && ((((uintptr_t)q.out) | (uint64_t)q.oW * 4u | p.plane_out_bytes) & 63u) != 0 && !aa_knob("AA_UP_NO_SPLIT"))
    p.store_nt = 2;
  // ... and when the rows are 8-byte but not 16-byte aligned (oW = 906): strips cut at the sector boundaries of each row instead
  // (see the store).  Measured, [256,3,196,320] gradients -> 438 x W (ms, split + pacing | sector-aligned pieces): W = 898 0.347 | 0.312,
  // 906 0.321 | 0.301-0.311; rows that are 16-byte aligned are better off with the split: 900 0.257 | 0.303, 904 0.269 | 0.284
  if (p.store_nt == 2 && q.oW % 4 == 2 && ((uintptr_t)q.out & 15) == 0 && !aa_knob("AA_UP_NO_ALN")) {
    p.store_nt = 3;
    p.strip_w = 240;
    p.nstrips = (int)((q.oW + 14 + 239) / 240);
    p.strips_per_block = p.nstrips <= 8 ? p.nstrips : 4;
  }
  p.pace_all = aa_knob("AA_UP_PACE_ALL") ? 1 : 0;
  p.ybands = 1;
  p.n_groups = 0;
  const size_t lds = (size_t)AA_UP_G * p.seg_bytes + (p.store_nt == 3 ? 1088 : 0);  // per strip: stage ring (+ the aligned-store staging area)

  int rc = 0;
#define AA_UP_CASE(UU, CC)                                                                     \
  if (g.u == UU && g.cpl == CC)                                                                \
    rc = q.dtype == AA_F32 ? launch_kr<UU, CC, AA_F32>(taps_h, p, q, lds)                      \
                           : (q.dtype == AA_F16 ? launch_kr<UU, CC, AA_F16>(taps_h, p, q, lds) \
                                                : launch_kr<UU, CC, AA_BF16>(taps_h, p, q, lds))
  AA_UP_CASE(2, 4); else AA_UP_CASE(3, 4); else AA_UP_CASE(4, 4); else AA_UP_CASE(5, 4);
  else AA_UP_CASE(2, 2); else AA_UP_CASE(3, 2); else AA_UP_CASE(4, 2); else AA_UP_CASE(5, 2); else AA_UP_CASE(6, 2); else AA_UP_CASE(8, 2);
  else AA_UP_CASE(2, 1); else AA_UP_CASE(3, 1); else AA_UP_CASE(4, 1); else AA_UP_CASE(5, 1); else AA_UP_CASE(6, 1); else AA_UP_CASE(8, 1);
#undef AA_UP_CASE
  if (rc == 1) *variant = q.dtype == AA_F32 ? "fused_f32_nchw_up" : (q.dtype == AA_F16 ? "fused_f16_nchw_up" : "fused_bf16_nchw_up");
  return rc;
}
