// aa_fused_u8_v2.hip — fused resample for uint8 channels_last, Pillow arithmetic: LDS-DMA staged, wave-specialised.
//
// Why a second design (measured on MI355X, profiles/r01_*): the first-generation kernel (aa_fused_u8.hip) lets every
// lane read its own overlapping 24-byte window straight from global memory.  HBM traffic is ideal (FETCH+WRITE =
// 1.006x the algorithmic bytes) but the per-lane requests cannot be coalesced: 68 L1 accesses per load instruction
// keep the TCP 95 % busy (TCP_GATE_EN), TD 82 % busy, and the kernel stalls at 3.4 TB/s.  Here every input byte
// crosses the L1 once, in full 16-byte pieces, and lands in LDS without touching a VGPR:
//
//   * producer waves (one per 64 output columns) each own a private LDS ring of G staged row segments.  A segment
//     is exactly the bytes the wave's 64 windows cover in one input row (~64*scale*C + taps*C bytes), fetched by ONE
//     `buffer_load_dwordx4 ... lds` (LDS-DMA, range-checked, 16 B per lane, SGPR row offset, no address VALU).
//     G-2 rows stay in flight behind a counted `s_waitcnt vmcnt(N)`; no barrier guards the staging data because a
//     wave only ever reads what it fetched itself;
//   * horizontal pass: one lane per output pixel reads its window from LDS with dword-aligned reads + v_alignbyte
//     (byte-unaligned LDS reads are replayed and 2-3x slower — measured), after which the bytes sit at fixed
//     positions and the multiplies select them with SDWA: 1 v_mul_i32_i24_sdwa per tap-channel + 1 v_add3 per two.
//     The clipped uint8 intermediate (Pillow rounds between passes) goes into the ring shared with the consumers;
//   * consumer waves do the vertical pass and the global stores.  Splitting the roles keeps the producers' VM
//     counter free of stores, so their counted waits stay exact, and the vertical pass of one row group overlaps
//     the horizontal pass of the next (work split ~73 % / 27 % -> 5 producer : 2 consumer waves for 320 columns);
//   * one raw `s_barrier` per G input rows hands ring rows from producers to consumers; the ring depth
//     (2G + taps_h + 1) makes the write-after-read hazard impossible;
//   * the steady-state producer loop is unrolled over the G stage slots and carries no per-row branches: the first
//     version of this kernel spent more scalar than vector instructions (SQ_INSTS_SALU 156 M vs VALU 124 M per
//     launch) on bookkeeping.
// All integer ops except f32 FMA/MUL/ADD issue at 4 cycles per wave on gfx950 (scratch/ubench_valu.hip), so the
// kernel is VALU-bound once the memory path is clean; DESIGN.md has the budget.

#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "aa_common.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void;

struct FusedU8V2Params {
  int H, W, oH, oW;
  int ksize_w, ksize_h;
  int ybands, xbands;
  int bw;               // output columns per x band (multiple of 4)
  int ring_rows;        // intermediate ring depth (rows)
  unsigned ring_magic;  // floor(2^32/ring_rows)+1
  int pitch;            // ring row pitch in bytes (multiple of 16)
  int nseg;             // 16-byte pieces per staged row segment (<= 128)
  int seg_bytes;        // nseg * 16
  int n_prod, n_cons;   // producer / consumer waves
  int ring_off;         // byte offset of the ring inside the LDS array
  int in_mis;           // (input pointer & 15): the kernel gets the pointer rounded down to 16 B
  unsigned long long img_in_bytes, img_out_bytes, total_in_bytes;
};

// v_ashr_pk_u8_i32 D, S0, S1, sh: D[7:0] = sat_u8(S0 >> sh), D[15:8] = sat_u8(S1 >> sh), other half of D preserved;
// op_sel[3] targets D[31:16].  Inline asm on purpose (see aa_fused_u8.hip: hipcc mis-widens its own pattern match).
__device__ inline unsigned pack4_clip8(int a0, int a1, int a2, int a3) {
  unsigned d;
  asm("v_ashr_pk_u8_i32 %0, %1, %2, 22\n\tv_ashr_pk_u8_i32 %0, %3, %4, 22 op_sel:[0,0,0,1]"
      : "=&v"(d)
      : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
  return d;
}

// wait until at most n of this wave's vector-memory operations are outstanding (n wave-uniform; rounding n DOWN to
// an available immediate only waits longer, never shorter)
__device__ inline void wait_vmcnt(int n) {
  if (n >= 12) { asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); return; }
  if (n >= 8) { asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); return; }
  if (n >= 6) { asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); return; }
  if (n >= 4) { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); return; }
  if (n >= 3) { asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); return; }
  if (n >= 2) { asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); return; }
  if (n >= 1) { asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); return; }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// producers -> consumers hand-off: LDS ops drained, then the hardware barrier (no vmcnt(0): the DMA stays in flight)
__device__ inline void group_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ------------------------------------------------------------------------------------------------ vertical pass
// One output row for the consumer waves: lanes take dwords d = first, first+stride, ...; TAPS taps with wave-uniform
// weights w[] and ring row offsets soff[].
template <int TAPS, int ABL = 0>
__device__ inline void vpass_row(const uint8_t *ring, const int (&soff)[12], const int (&w)[12], unsigned *orow, int first,
                                 int stride, int nd) {
  for (int d = first; d < nd; d += stride) {
    const uint8_t *src = ring + 4 * d;
    int a0 = 1 << 21, a1 = 1 << 21, a2 = 1 << 21, a3 = 1 << 21;
    if constexpr (ABL == 3 || ABL == 4) {
      const unsigned dw = *(const unsigned *)(src + soff[0]);
      orow[d] = dw + (unsigned)w[0];
      continue;
    }
#pragma unroll
    for (int j = 0; j < TAPS; j++) {
      const unsigned dw = *(const unsigned *)(src + soff[j]);
      a0 += (int)(dw & 0xffu) * w[j];
      a1 += (int)((dw >> 8) & 0xffu) * w[j];
      a2 += (int)((dw >> 16) & 0xffu) * w[j];
      a3 += (int)(dw >> 24) * w[j];
    }
    orow[d] = pack4_clip8(a0, a1, a2, a3);
  }
}

// ------------------------------------------------------------------------------------------------ the kernel
// G = staged rows per producer wave = input rows per barrier (even).
// TWO_DMA: a staged segment needs two 1-KiB LDS-DMA pieces (nseg > 64)
// ABL (experiment builds only): 0 = real kernel; 1 = skip the horizontal-pass arithmetic; 2 = skip the DMA;
// 3 = skip the vertical-pass arithmetic; 4 = skip 1+3 (pure data movement)
template <int C, int TW, int G, bool TWO_DMA, int ABL = 0>
__global__ void __launch_bounds__(1024)
fused_u8_nhwc_v2_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, const char *__restrict__ tab_w,
                        const char *__restrict__ tab_h, const FusedU8V2Params p) {
  constexpr int NV = (C * TW + 3) / 4;  // dwords holding one window
  constexpr int ND = NV + 1;            // aligned dwords fetched per window
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];

  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int b = blockIdx.x;
  const int xb = b % p.xbands;
  b /= p.xbands;
  const int yb = b % p.ybands;
  const int n = b / p.ybands;
  const int ox0 = xb * p.bw;
  const int bw = min(p.bw, p.oW - ox0);
  const int oy0 = (int)((long long)yb * p.oH / p.ybands);
  const int oy1 = (int)((long long)(yb + 1) * p.oH / p.ybands);

  const int32_t *__restrict__ xmin_w = (const int32_t *)(tab_w + aa_table_xmin_off());
  const int32_t *__restrict__ xsize_w = (const int32_t *)(tab_w + aa_table_xsize_off(p.oW));
  const int32_t *__restrict__ kw = (const int32_t *)(tab_w + aa_table_w_off(p.oW));
  const int32_t *__restrict__ ymin_h = (const int32_t *)(tab_h + aa_table_xmin_off());
  const int32_t *__restrict__ ysize_h = (const int32_t *)(tab_h + aa_table_xsize_off(p.oH));
  const int32_t *__restrict__ kh = (const int32_t *)(tab_h + aa_table_w_off(p.oH));

  uint8_t *const ring = lds + p.ring_off;
  const int ring_bytes = p.ring_rows * p.pitch;
  auto slot_of = [&](int r) -> int { return r - p.ring_rows * (int)__umulhi((unsigned)r, p.ring_magic); };

  // input rows this band needs: [r_begin, r_stop), handed over in groups of G rows
  const int r_begin = __builtin_amdgcn_readfirstlane(ymin_h[oy0]);
  const int ylm = __builtin_amdgcn_readfirstlane(ymin_h[oy1 - 1]);
  const int yls = __builtin_amdgcn_readfirstlane(ysize_h[oy1 - 1]);
  const int r_stop = ylm + (yls > 1 ? yls : 1);
  const int n_rows = r_stop - r_begin;
  const int n_groups = (n_rows + G - 1) / G;

  if (wid < p.n_prod) {
    // =========================================== producer wave ===========================================
    const int col = wid * 64 + lane;
    const bool active = col < bw;
    const bool wave_full = wid * 64 + 64 <= bw;  // wave-uniform
    const int ox = ox0 + (active ? col : wid * 64);
    const int xm = xmin_w[ox];
    int xs = xsize_w[ox];
    xs = xs > 1 ? xs : 1;
    int lead = xm + TW - p.W;  // right-align windows whose zero-weight padding would leave the row
    lead = lead > 0 ? lead : 0;
    const int start = xm - lead;
    int wreg[TW];
#pragma unroll
    for (int j = 0; j < TW; j++) {
      const int src = j - lead;
      int w = (src >= 0 && src < xs && src < p.ksize_w) ? kw[(size_t)ox * p.ksize_w + src] : 0;
      wreg[j] = (w << 8) >> 8;  // 24-bit operand for v_mul_i32_i24
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // table loads done: from here on vmcnt counts only the DMA
    const int seg_first = __builtin_amdgcn_readfirstlane(start * C);  // lane 0 of a producer wave is always active
    const int c_l = start * C - seg_first;                            // window offset inside the segment (bytes)

    const unsigned long long img_off = (unsigned long long)p.in_mis + (unsigned long long)n * p.img_in_bytes;
    const unsigned long long base_off = img_off & ~15ull;
    unsigned long long remaining = p.total_in_bytes - base_off;
    if (remaining > 0xFFFFFFFFull) remaining = 0xFFFFFFFFull;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void *)(in + base_off), 0, (unsigned)remaining, 0x00020000);
    const unsigned row_bytes = (unsigned)p.W * C;
    const int stage_base = wid * G * p.seg_bytes;
    const unsigned lane_lds = (unsigned)(stage_base + c_l);  // per-lane constant part of the window address
    const unsigned lane_ring = (unsigned)(p.ring_off + col * C);
    const bool dma_lane0 = lane < p.nseg;
    const bool dma_lane1 = lane + 64 < p.nseg;
    constexpr bool two_dma = TWO_DMA;
    constexpr int dma_per_row = two_dma ? 2 : 1;
    const unsigned voff = (unsigned)lane * 16u;

    // a: byte offset (from the descriptor base) of the segment start of the CURRENT row
    unsigned a = (unsigned)(img_off - base_off) + (unsigned)seg_first + (unsigned)r_begin * row_bytes;
    int ring_off = slot_of(r_begin) * p.pitch;  // byte offset of the current row's ring slot

    auto dma = [&](unsigned a_row, int slot) {  // fetch the segment of the row whose offset is a_row into stage slot
      if constexpr (ABL == 2) return;
      const unsigned soff = a_row & ~15u;
      const int dst = stage_base + slot * p.seg_bytes;
      if (dma_lane0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void *)(lds + dst), 16, voff, soff, 0, 0);
      if constexpr (two_dma) {
        if (dma_lane1)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void *)(lds + dst + 1024), 16, voff + 1024, soff, 0, 0);
      }
    };
    auto fetch = [&](unsigned a_row, int slot, unsigned (&d)[ND]) -> unsigned {  // issue the window reads of a staged row
      const unsigned sa = lane_lds + (unsigned)(slot * p.seg_bytes) + (a_row & 15u);
      const unsigned *al = (const unsigned *)(lds + (sa & ~3u));
#pragma unroll
      for (int k = 0; k < ND; k++) d[k] = al[k];
      return sa;
    };
    auto compute = [&](const unsigned (&d)[ND], unsigned sa, bool store) {  // horizontal pass of one row -> ring
      unsigned v[NV];
#pragma unroll
      for (int k = 0; k < NV; k++) v[k] = __builtin_amdgcn_alignbyte(d[k + 1], d[k], sa);
      int acc[C];
#pragma unroll
      for (int c = 0; c < C; c++) acc[c] = 1 << 21;
      if constexpr (ABL == 1 || ABL == 4) {
#pragma unroll
        for (int c = 0; c < C; c++) acc[c] += (int)(v[c % NV] ^ v[NV - 1]) + wreg[c % TW];
      } else {
#pragma unroll
        for (int j = 0; j < TW; j++) {
#pragma unroll
          for (int c = 0; c < C; c++) {
            const int bi = j * C + c;
            const int px = (int)((v[bi >> 2] >> (8 * (bi & 3))) & 0xffu);
            acc[c] += px * wreg[j];
          }
        }
      }
      if (store) {
        uint8_t *dst = lds + lane_ring + ring_off;
        if constexpr (C == 4) {
          *(unsigned *)dst = pack4_clip8(acc[0], acc[1], acc[2], acc[3]);
        } else {  // C == 3: bytes c0,c1,c2 in one register; byte 1 needs one shift before its byte store
          const unsigned q = pack4_clip8(acc[0], acc[1], acc[2], acc[2]);
          dst[0] = (uint8_t)q;
          dst[1] = (uint8_t)(q >> 8);
          dst[2] = (uint8_t)(q >> 16);
        }
      }
    };
    auto advance = [&]() {
      a += row_bytes;
      ring_off += p.pitch;
      if (ring_off >= ring_bytes) ring_off -= ring_bytes;
    };

    // prologue: the first G rows in flight, window reads of row 0 issued
    for (int i = 0; i < G; i++)
      if (i < n_rows) dma(a + (unsigned)i * row_bytes, i);
    unsigned d0[ND], d1[ND];
    unsigned sa0 = 0, sa1 = 0;
    {
      const int younger = (n_rows < G ? n_rows : G) - 1;
      wait_vmcnt(younger * dma_per_row);
      sa0 = fetch(a, 0, d0);
    }
    // Invariant at the top of every row (index x, slot x % G): DMAs issued up to row x+G-1; row x's window reads
    // issued into d0 (x even) / d1 (x odd).
    for (int g = 0; g < n_groups; g++) {
      const int x0 = g * G;
      if (wave_full && x0 + 2 * G <= n_rows) {
        // ---- steady state: every lane active; all G rows, their prefetches and refills exist; no branches -----
#pragma unroll
        for (int i = 0; i < G; i++) {
          // rows issued after row x+1 = x+2 .. x+G-1  ->  G-2 may stay in flight
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(dma_per_row * (G - 2)) : "memory");
          if ((i & 1) == 0) {
            sa1 = fetch(a + row_bytes, (i + 1) % G, d1);
            compute(d0, sa0, true);
          } else {
            sa0 = fetch(a + row_bytes, (i + 1) % G, d0);
            compute(d1, sa1, true);
          }
          dma(a + (unsigned)G * row_bytes, i);  // the slot just consumed gets row x+G
          advance();
        }
      } else {
        // ---- tail groups: same order of operations, every step guarded ---------------------------------------
        for (int i = 0; i < G; i++) {
          const int x = x0 + i;
          if (x >= n_rows) break;
          if (x + 1 < n_rows) {
            int younger = n_rows - 1 - (x + 1);
            younger = younger < G - 2 ? younger : G - 2;
            wait_vmcnt(younger * dma_per_row);
            if ((i & 1) == 0) sa1 = fetch(a + row_bytes, (i + 1) % G, d1);
            else sa0 = fetch(a + row_bytes, (i + 1) % G, d0);
          }
          if ((i & 1) == 0) compute(d0, sa0, active);
          else compute(d1, sa1, active);
          if (x + G < n_rows) dma(a + (unsigned)G * row_bytes, i);
          advance();
        }
      }
      group_barrier();
    }
  } else {
    // =========================================== consumer wave ===========================================
    const int cw = wid - p.n_prod;
    const int nd = (bw * C) >> 2;  // dwords per output row of this band
    const int first = cw * 64 + lane;
    const int stride = p.n_cons * 64;
    unsigned *const out_img = (unsigned *)(out + (unsigned long long)n * p.img_out_bytes);
    int oy = oy0;
    int ym = __builtin_amdgcn_readfirstlane(ymin_h[oy]);
    int ys = __builtin_amdgcn_readfirstlane(ysize_h[oy]);
    ys = ys > 1 ? ys : 1;
    for (int g = 0; g < n_groups; g++) {
      group_barrier();  // rows [r_begin, avail) are in the ring
      int avail = r_begin + (g + 1) * G;
      avail = avail < r_stop ? avail : r_stop;
      while (oy < oy1 && ym + ys <= avail) {
        int w[12], soff[12];
        const int32_t *wrow = kh + (size_t)oy * p.ksize_h;
        const int s0 = slot_of(ym) * p.pitch;
#pragma unroll
        for (int j = 0; j < 12; j++) {
          const int wj = (j < ys) ? __builtin_amdgcn_readfirstlane(wrow[j < p.ksize_h ? j : 0]) : 0;
          w[j] = (wj << 8) >> 8;
          const int so = s0 + (j < ys ? j : 0) * p.pitch;
          soff[j] = so >= ring_bytes ? so - ring_bytes : so;
        }
        unsigned *orow = out_img + ((((size_t)oy * p.oW + ox0) * C) >> 2);
        switch (ys) {  // wave-uniform
          case 1: vpass_row<1, ABL>(ring, soff, w, orow, first, stride, nd); break;
          case 2: vpass_row<2, ABL>(ring, soff, w, orow, first, stride, nd); break;
          case 3: vpass_row<3, ABL>(ring, soff, w, orow, first, stride, nd); break;
          case 4: vpass_row<4, ABL>(ring, soff, w, orow, first, stride, nd); break;
          case 5: vpass_row<5, ABL>(ring, soff, w, orow, first, stride, nd); break;
          case 6: vpass_row<6, ABL>(ring, soff, w, orow, first, stride, nd); break;
          case 7: vpass_row<7, ABL>(ring, soff, w, orow, first, stride, nd); break;
          case 8: vpass_row<8, ABL>(ring, soff, w, orow, first, stride, nd); break;
          default: vpass_row<12, ABL>(ring, soff, w, orow, first, stride, nd); break;
        }
        oy++;
        if (oy < oy1) {
          ym = __builtin_amdgcn_readfirstlane(ymin_h[oy]);
          ys = __builtin_amdgcn_readfirstlane(ysize_h[oy]);
          ys = ys > 1 ? ys : 1;
        }
      }
    }
  }
}

int g_group = 8;  // G (experiment knob AA_V2_G in tuning builds)

int g_abl = 0;

template <int C, int TW, int G>
int launch_g(const FusedU8V2Params &p, const AAProblem &q, int block, size_t lds, int64_t grid) {
#ifdef AA_V2_TUNING
#define AA_ABL_CASE(N)                                                                                              \
  if (g_abl == N && p.nseg <= 64) {                                                                                 \
    hipLaunchKernelGGL((fused_u8_nhwc_v2_kernel<C, TW, G, false, N>), dim3((unsigned)grid), dim3(block), lds,        \
                       q.stream, (const uint8_t *)q.in - p.in_mis, (uint8_t *)q.out, (const char *)q.aw.table_dev,  \
                       (const char *)q.ah.table_dev, p);                                                            \
    AA_HIP_CHECK_LAUNCH();                                                                                          \
    return 1;                                                                                                       \
  }
  if (G == 8 && TW == 6) { AA_ABL_CASE(1) AA_ABL_CASE(2) AA_ABL_CASE(3) AA_ABL_CASE(4) }
#endif
  if (p.nseg > 64)
    hipLaunchKernelGGL((fused_u8_nhwc_v2_kernel<C, TW, G, true>), dim3((unsigned)grid), dim3(block), lds, q.stream,
                       (const uint8_t *)q.in - p.in_mis, (uint8_t *)q.out, (const char *)q.aw.table_dev,
                       (const char *)q.ah.table_dev, p);
  else
    hipLaunchKernelGGL((fused_u8_nhwc_v2_kernel<C, TW, G, false>), dim3((unsigned)grid), dim3(block), lds, q.stream,
                       (const uint8_t *)q.in - p.in_mis, (uint8_t *)q.out, (const char *)q.aw.table_dev,
                       (const char *)q.ah.table_dev, p);
  AA_HIP_CHECK_LAUNCH();
  return 1;
}

template <int C, int TW>
int launch(const FusedU8V2Params &p, const AAProblem &q, int block, size_t lds, int64_t grid) {
#ifdef AA_V2_TUNING
  if (g_group == 4) return launch_g<C, TW, 4>(p, q, block, lds, grid);
  if (g_group == 6) return launch_g<C, TW, 6>(p, q, block, lds, grid);
  if (g_group == 10) return launch_g<C, TW, 10>(p, q, block, lds, grid);
  if (g_group == 12) return launch_g<C, TW, 12>(p, q, block, lds, grid);
#endif
  return launch_g<C, TW, 8>(p, q, block, lds, grid);
}

template <int C>
int dispatch_tw(int tw, const FusedU8V2Params &p, const AAProblem &q, int block, size_t lds, int64_t grid) {
  if (tw <= 2) return launch<C, 2>(p, q, block, lds, grid);
  if (tw <= 4) return launch<C, 4>(p, q, block, lds, grid);
  if (tw <= 6) return launch<C, 6>(p, q, block, lds, grid);
  if (tw <= 8) return launch<C, 8>(p, q, block, lds, grid);
  if (tw <= 12) return launch<C, 12>(p, q, block, lds, grid);
  return 0;
}

int round_tw(int taps) {
  const int opts[] = {2, 4, 6, 8, 12};
  for (int o : opts)
    if (taps <= o) return o;
  return 0;
}

}  // namespace

int aa_try_fused_u8_nhwc_v2(const AAProblem &q, const char **variant) {
  if (q.dtype != AA_U8 || q.layout != AA_NHWC) return 0;
  if (q.ah.kind != AA_TABLE_PIL || q.aw.kind != AA_TABLE_PIL) return 0;
  const int C = (int)q.C;
  if (C != 3 && C != 4) return 0;
  const int taps_w = q.aw.max_taps > 0 ? q.aw.max_taps : q.aw.ksize;
  const int taps_h = q.ah.max_taps > 0 ? q.ah.max_taps : q.ah.ksize;
  const int tw = round_tw(taps_w);
  if (tw == 0 || q.W < tw || taps_h > 12) return 0;
  if ((q.oW * C) % 4 != 0) return 0;
  if ((uint64_t)q.H * q.W * C > 0xFFFFFFF0ull || q.H >= (1 << 20)) return 0;
  if (((uintptr_t)q.out & 3) != 0) return 0;
#ifdef AA_V2_TUNING
  if (const char *e = getenv("AA_V2_G")) g_group = atoi(e);
  if (const char *e = getenv("AA_V2_ABL")) g_abl = atoi(e);
#endif
  const int G = g_group;

  FusedU8V2Params p;
  p.H = (int)q.H; p.W = (int)q.W; p.oH = (int)q.oH; p.oW = (int)q.oW;
  p.ksize_w = q.aw.ksize; p.ksize_h = q.ah.ksize;
  p.img_in_bytes = (unsigned long long)q.H * q.W * C;
  p.img_out_bytes = (unsigned long long)q.oH * q.oW * C;
  p.in_mis = (int)((uintptr_t)q.in & 15);
  p.total_in_bytes = p.img_in_bytes * (unsigned long long)q.N + (unsigned long long)p.in_mis;

  // column bands of at most 512 columns (8 producer waves)
  int xbands = (int)((q.oW + 511) / 512);
  int bw = (int)((q.oW + xbands - 1) / xbands);
  bw = (bw + 3) & ~3;
  xbands = (int)((q.oW + bw - 1) / bw);
  p.bw = bw;
  p.xbands = xbands;
  p.n_prod = (bw + 63) / 64;
  // vertical pass ~ (oH*taps_h)/(H*taps_w) of the horizontal work per column: 2 consumers per 5 producers fits the
  // 2.2-2.8x down-scales this path is tuned for (clamped to 1..4)
  const double hwork = (double)q.H * tw * 1.0, vwork = (double)q.oH * taps_h * 1.1;
  int n_cons = (int)(p.n_prod * vwork / hwork + 0.7);
  if (n_cons < 1) n_cons = 1;
  if (n_cons > 4) n_cons = 4;
  if (const char *e = getenv("AA_V2_NCONS")) n_cons = atoi(e);  // experiment knob
  p.n_cons = n_cons;
  const int block = (p.n_prod + p.n_cons) * 64;
  if (block > 1024) return 0;

  // segment: bytes covered by 64 consecutive windows of one input row, + up to 15 bytes of 16-B alignment slack
  // (xmin[i+63] - xmin[i] <= floor(63*scale)+1 for the unclamped window starts; clamping only shrinks it)
  const double scale_w = (double)q.W / (double)q.oW;
  const int span_px = (int)floor(63.0 * (scale_w > 0 ? scale_w : 0)) + 1 + tw;
  p.nseg = (span_px * C + 3 + 15 + 15) / 16;  // +3: the aligned dword reads may run 3 bytes past the window
  if (p.nseg > 128) return 0;
  p.seg_bytes = p.nseg * 16;

  // ring depth: consumers of group g read rows > r_begin + G*g - taps_h while producers write up to r_begin+G*(g+2)
  p.ring_rows = 2 * G + taps_h + 1;
  p.ring_magic = (unsigned)(0x100000000ull / (unsigned)p.ring_rows) + 1u;
  p.pitch = ((bw * C + 15) / 16) * 16;
  p.ring_off = p.n_prod * G * p.seg_bytes;
  const size_t lds = (size_t)p.ring_off + (size_t)p.ring_rows * p.pitch;
  if (lds > 64 * 1024) return 0;

  const int cus = aa_device_cu_count();
  const int waves_per_block = block / 64;
  int blocks_per_cu = (int)((160 * 1024) / lds);
  if (blocks_per_cu > 32 / waves_per_block) blocks_per_cu = 32 / waves_per_block;
  if (blocks_per_cu < 1) blocks_per_cu = 1;
  const double slots = (double)cus * blocks_per_cu;
  const int64_t max_yb = q.oH / 8 > 1 ? q.oH / 8 : 1;
  int64_t ybands = 1;
  double best = 1e30;
  for (int64_t yb = 1; yb <= max_yb && yb <= 64; yb++) {
    const double items = (double)q.N * xbands * yb;
    const double rounds = items / slots;
    const double eff = rounds / ceil(rounds);
    const double halo = 1.0 + (double)(yb - 1) * taps_h / (double)q.H;
    const double cost = halo / eff;
    if (cost < best - 1e-9) {
      best = cost;
      ybands = yb;
    }
  }
  if (const char *e = getenv("AA_FUSED_YBANDS")) {  // experiment knob
    const int64_t v = atoll(e);
    if (v >= 1 && v <= max_yb) ybands = v;
  }
  p.ybands = (int)ybands;
  const int64_t grid = q.N * ybands * xbands;
  if (grid > 0x7FFFFFFF) return 0;

  const int rc = (C == 3) ? dispatch_tw<3>(tw, p, q, block, lds, grid) : dispatch_tw<4>(tw, p, q, block, lds, grid);
  if (rc == 1) *variant = "fused_u8_nhwc_pil_v2";
  return rc;
}
