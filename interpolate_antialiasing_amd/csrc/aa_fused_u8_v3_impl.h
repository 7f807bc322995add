// aa_fused_u8_v3_impl.h — fused resample for uint8 channels_last, Pillow arithmetic: wave-autonomous streaming.
//
// Third design, after measuring the first two on MI355X (DESIGN.md §Kernels has the numbers):
//   v1 (aa_fused_u8.hip)     per-lane window loads from global memory -> TCP 95 % busy, 3.4 TB/s;
//   v2 (aa_fused_u8_v2.hip)  LDS-DMA staging + producer/consumer waves + shared LDS ring -> the data-movement skeleton
//                            alone (all arithmetic removed) takes 0.33 ms per 1024 images: LDS bound (window reads with
//                            2-3-way bank conflicts + 3 byte-writes per pixel + ring reads), plus barrier coupling.
// v3 keeps v2's clean global path and removes everything else that touched LDS:
//   * ONE WAVE = ONE WORKGROUP = one 64-column strip of one band of one image.  No barriers, no shared ring, no
//     consumer waves: waves are completely independent, so 32 of them fit a CU and hide each other's latency;
//   * each wave streams its strip's input-row segments into a private G-slot LDS ring with one LDS-DMA instruction
//     per row (`buffer_load_dwordx4 ... lds`, 16 B per lane, range-checked, zero address VALU), G-2 rows in flight;
//   * horizontal pass: one lane per output pixel, dword-aligned LDS window reads + v_alignbyte, C*taps SDWA
//     multiplies + add3 (as v2);
//   * the VERTICAL PASS RUNS IN REGISTERS, in scatter form: the clipped uint8 result of input row r (C values per
//     lane, never packed, never stored) is multiplied by the wave-uniform weights that row r has in the <=4 output
//     rows still open (the table's scatter section, read with scalar loads one row ahead) and accumulated into
//     per-lane accumulators: one v_mad_i32_i24 per tap-channel, no LDS, no extraction;
//   * when an output row's last input row has been absorbed, its accumulators are clipped and packed
//     (v_ashr_pk_u8_i32), the 3-byte pixels of each lane quad are merged into 3 dwords with one DPP move + one
//     v_perm, and 48 of 64 lanes store one dword each: a 192-byte contiguous, fully coalesced row segment.
// Integer arithmetic is associative, so accumulating taps in scatter order gives bit-identical Pillow results.

//
// This header holds the kernel template and its launch chain; it is compiled once per channel count
// (aa_fused_u8_v3_c1.hip / _c3.hip / _c4.hip) so that the instantiations build in parallel, and the host-side
// dispatcher lives in aa_fused_u8_v3.hip.
#pragma once
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "aa_common.h"

#ifndef AA_V3_ABL
#define AA_V3_ABL 0  // developer ablations (wrong results!): 1 no stores, 2 no DMA waits, 4 no DMA at all,
                     // 5 no horizontal MACs, 6 DMA only (no LDS reads, no arithmetic, no stores),
                     // 7 DMAs that always hit cache, 9 every output row stored over row 0 (stores that stay in cache)
#endif

#ifndef AA_V3_AUX
#define AA_V3_AUX 0  // cache-policy bits of the staging DMA (developer knob)
#endif

#ifndef AA_V3_PRIO
#define AA_V3_PRIO 0  // developer knob: 1 = raise the wave's issue priority around its staging DMA
#endif
#ifndef AA_V3_DMA_EARLY
#define AA_V3_DMA_EARLY 0  // developer knob: 1 = refill a slot as soon as its window is in registers (before the row's arithmetic)
#endif

#ifndef AA_V3_BURST
#define AA_V3_BURST 1  // staging DMAs are issued in bursts of this many rows (divides G): a wave then meets a vector-memory
                       // instruction - where it blocks, in order, while the CU's address FIFO is full - every BURST rows
#endif

#ifndef AA_V3_F32OUT_AUX
#define AA_V3_F32OUT_AUX 0  // cache policy of the float32-output stores (decode-adjacent conversion).  nt measured: NCHW output 0.478 -> 0.505 ms, NHWC unchanged: default kept
#endif

#ifndef AA_V3_FLT_FAST
#define AA_V3_FLT_FAST 0  // 1 (aa_fused_u8_v3_c{1,3,4}ff.hip): the float-arithmetic kernels in the opt-in TOLERANCE mode (AA_FLAG_FAST): every tap of
                          // both passes is one fused multiply-add instead of a separately rounded product and sum.  Results differ from the
                          // reference harness's by rounding only (<= 1e-4 relative; after the truncating byte() at most one count)
#endif

#ifndef AA_V3_UNALIGNED
#define AA_V3_UNALIGNED 0  // 1: window reads straight from the window's BYTE address (gfx950's LDS does serve unaligned
                           // ds_read_b32/b64, and hipcc emits them for align-1 pointers), saving the 5 v_alignbyte per row.
                           // MEASURED on MI355X (profiles/r02_unaligned_lds_reads.txt): results identical, kernel 3.6x
                           // SLOWER (1.02 ms vs 0.283 ms per 1024 images): a misaligned LDS dword is not a one-pass access.
#endif

// (shared by the per-channel-count translation units and the host-side dispatcher)
struct FusedU8V3Params {
  int H, W, oH, oW;
  int ksize_w, ksize_h;
  int ybands, nstrips;
  int strips_per_block;  // waves per workgroup
  int strip_w;           // output columns per strip (<= 64, multiple of 4)
  int nseg;        // 16-byte pieces per staged row segment (<= 128)
  int seg_bytes;   // nseg * 16
  int sc_off;      // scatter section of the H table (bytes from table start): one 8-int record per input row
  int gather_off;  // gather section of the H table (growing heights): one 8-int record {ymin, ysize, w[6]} per output row
  int in_mis;      // (input pointer & 15): the kernel gets the pointer rounded down to 16 B
  unsigned long long img_in_bytes, img_out_bytes, total_in_bytes, total_out_bytes;
  long long n_images;  // = N for channels_last, N*C for planar input
  int spb_forced;      // strips_per_block comes from the AA_V3_SPB experiment knob: keep it
  // float-arithmetic kernels only: 0 = uint8 output in the input's layout (the harness's truncating byte()); 1 = float32
  // output, channel planes (NCHW); 2 = float32 output, interleaved channels (NHWC) — the fp32 result itself, before any
  // byte(): what np.asarray(pil) -> transpose -> .float() -> op gives (test.py:337-339,55), optionally (v - mean) / std
  int outm, normalize, cin;
  float mean[4], std[4];
  unsigned row_pitch;  // bytes between consecutive input rows (= W * C for a dense tensor; larger for a cropped view)
  int fast;        // AA_FLAG_FAST on a float-arithmetic problem: the FMA instantiations (aa_fused_u8_v3_c{1,3,4}ff.hip)
  int byte_store;  // output rows that are not whole dwords (oW*C % 4 != 0, or C == 3 with oW % 4 != 0) or an output pointer that
                   // is not dword aligned: every lane stores its own bytes instead of the quad-merged dword stores
  // plane-group kernels (template parameter PL): bytes between the channel planes of one image, input and output
  unsigned long long plane_in_bytes, plane_out_bytes;
  long long pl_planes;  // planes of the whole tensor (N * C): a group is PL CONSECUTIVE planes, whatever image they belong to — planes are
                        // independent and uniformly spaced, so grayscale batches and 2-, 4-, 5-channel planar images group the same way; the
                        // last group may hold fewer (its missing planes are refused by the range check and never stored)
};

namespace {

typedef __attribute__((address_space(3))) void lds_void;

// cache policy of the output stores of the interleaved-channel kernels: nt (streaming).  Their 192-byte pieces are whole
// 64-byte sectors, written once and never read here: -2 % against the default policy (sc0 alone changes nothing).  The
// planar kernel's 64-byte pieces and the fp32 kernels' 244-byte pieces get SLOWER with nt (0.55 -> 0.59 / 0.62 ms): default.
#ifndef AA_V3_STORE_AUX
#define AA_V3_STORE_AUX 2
#endif
constexpr int kStoreAuxInterleaved = AA_V3_STORE_AUX;



__device__ inline unsigned pack4_clip8(int a0, int a1, int a2, int a3) {  // semantics: see aa_fused_u8.hip
  unsigned d;
  asm("v_ashr_pk_u8_i32 %0, %1, %2, 22\n\tv_ashr_pk_u8_i32 %0, %3, %4, 22 op_sel:[0,0,0,1]"
      : "=&v"(d)
      : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
  return d;
}

__device__ inline int clip8_int(int acc) {  // Pillow clip8(ss >> 22) as a plain integer 0..255
  acc >>= 22;
  return acc < 0 ? 0 : (acc > 255 ? 255 : acc);
}

__device__ inline void wait_vmcnt(int n) {  // rounding n DOWN only waits longer
  if (n >= 12) { asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); return; }
  if (n >= 8) { asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); return; }
  if (n >= 6) { asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); return; }
  if (n >= 4) { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); return; }
  if (n >= 3) { asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); return; }
  if (n >= 2) { asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); return; }
  if (n >= 1) { asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); return; }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// G: staged rows per wave (even).  MAXC: outputs one input row can feed (= accumulator sets kept open).
// NONNEG: no negative weights (triangle / box): the horizontal-pass sum cannot leave [0, 255.5], so Pillow's clip8 of the
//         intermediate is a plain shift.  PERIODIC: G*row_bytes is a multiple of 16, so the 16-byte phase of a staged
//         row depends only on its stage slot and the per-slot LDS window addresses are loop invariants.
// FLT: the reference harness's uint8 semantics instead of Pillow's (AA_TABLE_F32 tables): bytes are converted to fp32,
//      both passes run in fp32 with separately rounded product and sum in tap order (the intermediate is never rounded),
//      the result is clamped to [0,255] and truncated (test.py:52-58,72,75).  Registers hold float bit patterns.
#ifndef AA_V3_WAVES_PER_EU
#define AA_V3_WAVES_PER_EU 0  // developer knob: force the register budget for this many waves per SIMD (0: compiler's choice)
#endif
#if AA_V3_WAVES_PER_EU
#define AA_V3_OCC __attribute__((amdgpu_waves_per_eu(AA_V3_WAVES_PER_EU, AA_V3_WAVES_PER_EU)))
#else
#define AA_V3_OCC
#endif
// UPK > 0: heights that GROW (oH > H).  An input row then feeds more output rows than a scatter record holds, so the vertical pass
//      GATHERS: the horizontal-pass results of the last UPK input rows stay in a register ring, and every output row whose window
//      ends at the row just filtered is the weighted sum of the ring's last `ysize` entries (weights: one scalar load of the H
//      table's gather record, taps in order).  The row pipeline is the generic-address one; DMA completion is tracked per stage
//      slot because the output stores (several per input row) share the in-order vmcnt counter with the DMAs.
// PL: plane groups — planar (NCHW) images of PL = C channel planes: one wave filters the same strip and band of ALL the image's planes,
//     row by row.  The single-plane form (C = 1, one wave per plane) spends as much on a row's fixed work — scatter record, waits, loop,
//     window addressing: the scalar unit is 0.81 busy beside a 0.85 busy vector ALU (r02_pmc_u8_planar.json) — as on its 6 multiplies;
//     here that work is shared by the planes.  What bounds the single-plane form, though, is the vector memory unit's ADDRESS path: a
//     staging DMA costs it the same whether 10 of its lanes fetch or 64, and planar rows need one per plane and strip
//     (SQ_VMEM_TA_ADDR_FIFO_FULL: 2 cycles for every cycle the unit works, r03_pmc_u8_planar_groups.json).  So ONE DMA per row stages
//     the segments of all PL planes (lane = plane * nseg + piece, PL * nseg <= 64).  The planes' segments start on different 16-byte
//     phases, so the DMA reads from each segment's exact first byte (LDS-DMA serves any source alignment): every staged image then
//     starts at LDS phase 0 and the window addresses are per-slot constants.  Price of unaligned dwords: the range check refuses a
//     dword that straddles the end of the tensor, so the last 3 bytes of the tensor's very last row are fetched on their own
//     (pl_patch_last; nothing is ever read past the tensor).  One single-channel window per plane, the same weights for all of them (either
//     arithmetic: Pillow's integers, or FLT = the harness's floats with uint8 or float32 planes out);
//     vertical pass and accumulators as for C interleaved channels; one 64-byte row piece stored per plane.  Pillow arithmetic,
//     shrinking heights.
// SP: split windows — SP = 4 LANES share one output pixel, each holding a quarter (TW taps) of its window; the four partial sums meet in two
//     DPP additions (quad_perm [1,0,3,2], then [2,3,0,1]), after which every lane of the quad holds the pixel's horizontal-pass result and
//     runs the vertical pass redundantly.  Integer sums are associative, so the result is Pillow's bit for bit whatever the split.  This is
//     the form for windows beyond what one lane's registers hold (35 .. 136 taps: down-scaling by 17 .. 68 bilinear, 9 .. 34 bicubic); a
//     strip is 16 columns, stored bytewise by each quad's first lane (outputs are tiny at such scales).
template <int C, int TW, int G, bool TWO_DMA, int MAXC, bool NONNEG, bool PERIODIC, bool FLT = false, int UPK = 0, int PL = 0, int SP = 1>
__global__ void __launch_bounds__(512) AA_V3_OCC
fused_u8_nhwc_v3_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, const char *__restrict__ tab_w,
                        const char *__restrict__ tab_h, const FusedU8V3Params p) {
  static_assert(PL == 0 || (PL == C && !PERIODIC && UPK == 0 && !TWO_DMA), "plane groups: shrinking heights, fixed stage layout");
  static_assert(SP == 1 || (SP == 4 && !FLT && !PERIODIC && UPK == 0 && PL == 0), "split windows: integer arithmetic, shrinking heights");
  constexpr int PLN = PL > 0 ? PL : 1;  // windows per lane and row
  constexpr int CB = PL > 0 ? 1 : C;    // bytes per pixel in a staged row
  constexpr int NV1 = (CB * TW + 3) / 4;  // dwords holding one window
  constexpr int ND1 = NV1 + 1;            // aligned dwords fetched per window
  constexpr int NV = PLN * NV1, ND = PLN * ND1;
  struct SA { unsigned s[PLN]; };  // LDS byte address of each window of a row (its low two bits: the realignment shift)
  // plane groups: FIXED stage layout — a slot is 1024 bytes (one DMA: 64 lanes x 16 B), plane c's segment starts at c * 336 (21 pieces
  // per plane, PL * 21 <= 64), so every window read is the lane's constant base + an immediate offset: no address arithmetic per row
  constexpr int PL_SLOT = 1024, PL_PIECES = 21, PL_PLANE = PL_PIECES * 16;
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];

  // a workgroup is the nstrips_blk independent waves (strips) of one band: no barrier, no shared LDS; they only
  // share a CU so that the 64-byte sectors two neighbouring segments have in common come from L1/L2, not HBM
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // XCD-aware index mapping: the dispatcher deals consecutive workgroup ids round-robin to the chip's 8 XCDs, each with
  // its own L2.  Workgroup id b = 8 * k + xcd; within an XCD, consecutive k walk the strips of one (image, band) group
  // first, so that neighbouring strips (which share input sectors and output lines) meet in the SAME L2 at about the
  // same time.  The grid is padded to whole rounds of 8 groups; the surplus workgroups exit here.
  const int sgroups = (p.nstrips + p.strips_per_block - 1) / p.strips_per_block;
  const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
  const int strip = (k % sgroups) * p.strips_per_block + wv;
  const long long grp = (long long)(k / sgroups) * 8 + xcd;  // (image, band) group
  if (grp >= p.n_images * p.ybands) return;
  if (strip >= p.nstrips) return;
  // (the pacing barrier below is only used by workgroups all of whose waves are alive: a short last group skips it)
  const bool full_group = (k % sgroups + 1) * p.strips_per_block <= p.nstrips;
  const int yb = (int)(grp % p.ybands);
  const int n = (int)(grp / p.ybands);
  const int ox0 = strip * p.strip_w;
  const int bw = min(p.strip_w, p.oW - ox0);
  const int oy0 = (int)((long long)yb * p.oH / p.ybands);
  const int oy1 = (int)((long long)(yb + 1) * p.oH / p.ybands);

  const int32_t *__restrict__ xmin_w = (const int32_t *)(tab_w + aa_table_xmin_off());
  const int32_t *__restrict__ xsize_w = (const int32_t *)(tab_w + aa_table_xsize_off(p.oW));
  const int32_t *__restrict__ kw = (const int32_t *)(tab_w + aa_table_w_off(p.oW));
  const int32_t *__restrict__ ymin_h = (const int32_t *)(tab_h + aa_table_xmin_off());
  const int32_t *__restrict__ ysize_h = (const int32_t *)(tab_h + aa_table_xsize_off(p.oH));
  const int32_t *__restrict__ sc_rec = (const int32_t *)(tab_h + p.sc_off);
  const int32_t *__restrict__ g_rec = (const int32_t *)(tab_h + p.gather_off);  // (UPK > 0 only)

  // input rows this band needs: [r_begin, r_stop)
  const int r_begin = __builtin_amdgcn_readfirstlane(ymin_h[oy0]);
  const int ylm = __builtin_amdgcn_readfirstlane(ymin_h[oy1 - 1]);
  const int yls = __builtin_amdgcn_readfirstlane(ysize_h[oy1 - 1]);
  const int r_stop = ylm + (yls > 1 ? yls : 1);
  const int n_rows = r_stop - r_begin;
  const int n_groups = (n_rows + G - 1) / G;

  // ---- per-lane horizontal-pass state ------------------------------------------------------------------------
  const int sp_px = SP > 1 ? lane / SP : lane;    // the lane's pixel of the strip
  const int sp_part = SP > 1 ? lane % SP : 0;     // ... and its quarter of that pixel's window
  const bool active = sp_px < bw;
  const int ox = ox0 + (active ? sp_px : 0);
  const int xm = xmin_w[ox];
  int xs = xsize_w[ox];
  xs = xs > 1 ? xs : 1;
  int lead = xm + TW * SP - p.W;  // right-align windows whose zero-weight padding would leave the row
  lead = lead > 0 ? lead : 0;
  const int start = xm - lead + sp_part * TW;
  const int acc_init = (SP > 1 && sp_part != 0) ? 0 : 1 << 21;  // (split windows: Pillow's rounding constant enters once per pixel)
  int wreg[TW];
#pragma unroll
  for (int j = 0; j < TW; j++) {
    const int src = sp_part * TW + j - lead;
    int w = (src >= 0 && src < xs && src < p.ksize_w) ? kw[(size_t)ox * p.ksize_w + src] : 0;
    wreg[j] = FLT ? w : (w << 8) >> 8;  // 24-bit operand for v_mul_i32_i24 (FLT: the float's bit pattern, 0 = +0.0f)
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // table loads done: from here on vmcnt counts DMAs and stores
  const int seg_first = __builtin_amdgcn_readfirstlane(start * CB);  // lane 0 is always active
  const int c_l = start * CB - seg_first;                            // window offset inside the segment (bytes)

  const unsigned long long img_off = (unsigned long long)p.in_mis + (unsigned long long)n * p.img_in_bytes;
  const unsigned long long base_off = img_off & ~15ull;
  unsigned long long remaining = p.total_in_bytes - base_off;
  // the range check works per dword: serve the last, partial one too — an ALIGNED dword, it cannot leave the tensor's last page.  (Plane
  // groups read unaligned dwords: exact range, see pl_patch_last.)
  if constexpr (PL == 0) remaining = (remaining + 3ull) & ~3ull;
  if (remaining > 0xFFFFFFFCull) remaining = 0xFFFFFFFCull;
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void *)(in + base_off), 0, (unsigned)remaining, 0x00020000);
  const unsigned row_bytes = p.row_pitch;
  const int lds_base = PL > 0 ? wv * G * PL_SLOT : wv * G * p.seg_bytes;  // this wave's private stage ring
  const unsigned lane_lds = (unsigned)(lds_base + c_l);
  unsigned pl_voff = 0;  // plane groups: staging lane = plane * 21 + piece
  bool pl_dma_lane = false;
  if constexpr (PL > 0) {
    const int pc = lane / PL_PIECES, piece = lane - pc * PL_PIECES;
    pl_voff = (unsigned)pc * (unsigned)p.plane_in_bytes + (unsigned)piece * 16u;
    pl_dma_lane = pc < PL && piece < p.nseg;
  }
  const bool dma_lane0 = PL > 0 ? pl_dma_lane : lane < p.nseg;
  const bool dma_lane1 = lane + 64 < p.nseg;
  constexpr int dma_per_row = TWO_DMA ? 2 : 1;
  const unsigned voff = (unsigned)lane * 16u;

  // ---- output: range-checked view of this image's output, per-lane dword slot inside the strip's row segment ----
  const unsigned long long out_off = (unsigned long long)n * p.img_out_bytes;
  unsigned long long out_rem = p.total_out_bytes - out_off;
  if (out_rem > 0xFFFFFFFFull) out_rem = 0xFFFFFFFFull;
  const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(out + out_off), 0, (unsigned)out_rem, 0x00020000);
  const unsigned out_row_bytes = (unsigned)p.oW * CB;
  const int pl_valid = PL > 0 ? (int)(p.pl_planes - (long long)n * PL < PL ? p.pl_planes - (long long)n * PL : PL) : C;  // (wave-uniform)
  float nm_mean[C], nm_std[C];  // (float output with normalisation only)
#pragma unroll
  for (int c = 0; c < C; c++) {
    const int ch = PL > 0 ? (int)(((long long)n * PL + c) % (p.cin > 0 ? p.cin : 1))   // plane groups: plane n * PL + c of the tensor
                          : (C == 1 ? (int)(n % (p.cin > 0 ? p.cin : 1)) : c);      // planar input: this wave's plane is channel n % Cin
    nm_mean[c] = FLT ? p.mean[ch & 3] : 0.f;
    nm_std[c] = FLT ? p.std[ch & 3] : 1.f;
  }
  unsigned store_voff;
  bool store_lane;
  unsigned perm_sel = 0;
  if constexpr (PL > 0) {
    // plane groups: per plane, a quad of lanes holds 4 consecutive bytes = 1 dword, stored by the quad's first lane
    store_voff = (unsigned)(ox0 + lane);
    store_lane = ((lane & 3) == 0) && ((lane | 3) < bw);
  } else if constexpr (C == 3) {
    // a quad of lanes holds 4 pixels = 12 bytes = 3 dwords; quad lanes 0..2 each assemble and store one of them
    const int q = lane & 3;
    store_voff = (unsigned)(ox0 * 3 + (lane >> 2) * 12 + q * 4);
    store_lane = (q != 3) && ((lane | 3) < bw);  // bw is a multiple of 4: the whole quad is in range or none of it
    perm_sel = q == 0 ? 0x04020100u : (q == 1 ? 0x05040201u : 0x06050402u);
  } else if constexpr (C == 1) {
    // planar bytes: a quad of lanes holds 4 consecutive bytes = 1 dword, stored by the quad's first lane
    store_voff = (unsigned)(ox0 + lane);
    store_lane = ((lane & 3) == 0) && ((lane | 3) < bw);
  } else {
    store_voff = (unsigned)((ox0 + lane) * 4);
    store_lane = active;
  }

  // a: byte offset (from the descriptor base) of the segment start of the CURRENT row
  unsigned a = (unsigned)(img_off - base_off) + (unsigned)seg_first + (unsigned)r_begin * row_bytes;

  // ---- vertical-pass state: MAXC accumulator sets, set k belongs to output row o_base + k --------------------------
  int A[MAXC][C];
#pragma unroll
  for (int k = 0; k < MAXC; k++)
#pragma unroll
    for (int c = 0; c < C; c++) A[k][c] = FLT ? 0 : 1 << 21;
  int o_base = oy0;

  // gather mode: ring of the last UPK horizontal-pass results; VMEM instructions issued so far and, in lane s, their count right
  // after the DMA that filled stage slot s
  constexpr int RK = UPK > 0 ? UPK : 1;
  int ring[RK][C];
#pragma unroll
  for (int k = 0; k < RK; k++)
#pragma unroll
    for (int c = 0; c < C; c++) ring[k][c] = 0;
  int vm_issued = 0, idxv = 0;

  auto dma = [&](unsigned a_row, int slot) {
    if (AA_V3_ABL == 4) return;
    if constexpr (UPK > 0) {  // (lane 0 always takes part in the DMA: the instruction is certainly issued)
      vm_issued += TWO_DMA ? 2 : 1;
      idxv = (lane == slot) ? vm_issued : idxv;
    }
    if (AA_V3_PRIO) __builtin_amdgcn_s_setprio(3);
    const unsigned soff = AA_V3_ABL == 7 ? (a_row & 0x3F0u) : (a_row & ~15u);  // 7: every DMA hits the same 1.6 KB
    const int dst = lds_base + slot * (PL > 0 ? PL_SLOT : p.seg_bytes);
    if constexpr (PL > 0) {  // ONE DMA for the row's PL segments, each from its exact first byte; side by side in the slot
      if (dma_lane0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void *)(lds + dst), 16, pl_voff, a_row, 0, AA_V3_AUX);
      if (AA_V3_PRIO) __builtin_amdgcn_s_setprio(0);
      return;
    }
    if (dma_lane0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void *)(lds + dst), 16, voff, soff, 0, AA_V3_AUX);
    if constexpr (TWO_DMA) {
      if (dma_lane1)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void *)(lds + dst + 1024), 16, voff + 1024, soff, 0, AA_V3_AUX);
    }
    if (AA_V3_PRIO) __builtin_amdgcn_s_setprio(0);
  };
  // per-slot loop invariants of the window reads (PERIODIC only): dword-aligned LDS address and 16-byte phase
  unsigned slot_ra[G];
  unsigned slot_ph[G];
#pragma unroll
  for (int j = 0; j < G; j++) {
    const unsigned ph = (a + (unsigned)j * row_bytes) & 15u;
    slot_ph[j] = (unsigned)(j * p.seg_bytes) + ph;  // uniform
    slot_ra[j] = (lane_lds + slot_ph[j]) & ~3u;      // per lane
  }
  auto fetch = [&](unsigned a_row, int slot, unsigned (&d)[ND], bool tables = true) -> SA {
    SA r;
    unsigned sa, ra;
    if (AA_V3_ABL == 6) { for (int c = 0; c < PLN; c++) r.s[c] = 0; return r; }
    if constexpr (PL > 0) {  // (every staged segment starts at phase 0 of its fixed place: constant base, immediate offsets)
      const __attribute__((address_space(3))) unsigned *al = (const __attribute__((address_space(3))) unsigned *)(uintptr_t)(lane_lds & ~3u);
#pragma unroll
      for (int c = 0; c < PL; c++) {
#pragma unroll
        for (int k = 0; k < ND1; k++) d[c * ND1 + k] = al[(slot * PL_SLOT + c * PL_PLANE) / 4 + k];
        r.s[c] = lane_lds;
      }
      return r;
    }
    if (PERIODIC && tables) {  // `slot` must be a compile-time constant here (register arrays)
      sa = lane_lds + slot_ph[slot];  // only its low two bits are used (v_alignbyte)
      ra = slot_ra[slot];
    } else {
      sa = lane_lds + (unsigned)(slot * p.seg_bytes) + (a_row & 15u);
      ra = sa & ~3u;
    }
    if constexpr (AA_V3_UNALIGNED) {
      typedef unsigned u32_any __attribute__((aligned(1)));
      const __attribute__((address_space(3))) u32_any *ul = (const __attribute__((address_space(3))) u32_any *)(uintptr_t)sa;
#pragma unroll
      for (int k = 0; k < NV; k++) d[k] = ul[k];
      d[ND - 1] = 0;
      r.s[0] = sa;
      return r;
    }
    const __attribute__((address_space(3))) unsigned *al = (const __attribute__((address_space(3))) unsigned *)(uintptr_t)ra;
#pragma unroll
    for (int k = 0; k < ND; k++) d[k] = al[k];
    r.s[0] = sa;
    return r;
  };
  // scatter record of an input row: first output it feeds, and its weight in that output and the next MAXC-1
  struct Scatter { int first; int cc; int w[MAXC]; };  // raw record words (nothing depends on them until they are used)
  auto load_scatter = [&](int r) -> Scatter {  // one 32-byte record: {first, count | completes << 16, w[6]}
    Scatter s;                                   // (the section has H + 1 records: r == H reads the all-zero sentinel)
    if constexpr (UPK > 0) {  // gather mode never reads the scatter section
      s.first = 0; s.cc = 0;
#pragma unroll
      for (int k = 0; k < MAXC; k++) s.w[k] = 0;
      return s;
    }
    const int32_t *rec = (const int32_t *)((const char *)sc_rec + (unsigned)r * 32u);
    s.first = __builtin_amdgcn_readfirstlane(rec[0]);
    s.cc = __builtin_amdgcn_readfirstlane(rec[1]);  // count | completes << 16
#pragma unroll
    for (int k = 0; k < MAXC; k++) s.w[k] = __builtin_amdgcn_readfirstlane(rec[2 + k]);
    return s;
  };
  auto trunc8 = [&](int bits) -> unsigned {  // harness: clamp to [0,255], truncating conversion (generic Store<uint8_t,float>)
    float a = __int_as_float(bits);
    a = a < 0.f ? 0.f : (a > 255.f ? 255.f : a);
    return (unsigned)(int)a;
  };
  // which store form the kernel uses is a launch constant: a scalar, so that the per-row choice is a scalar compare (as a bool
  // merged across the branches below the compiler kept it in a lane mask: a v_cndmask + v_cmp per output row)
  const int emit_path = __builtin_amdgcn_readfirstlane((FLT && p.outm != 0) ? 1 : (p.byte_store ? 2 : 0));
  auto emit = [&](int oy) {  // accumulator set 0 is complete: clip, pack, merge quads, store; then slide the sets down
    if constexpr (FLT) {
      if (emit_path == 1) {  // (wave-uniform) float32 output: the accumulators themselves, one 256-byte row piece per plane
        float v[C];
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));  // (address arithmetic of this path stays inside it, see the byte-store path)
#pragma unroll
        for (int c = 0; c < C; c++) {
          v[c] = __int_as_float(A[0][c]);
          if (p.normalize) v[c] = (v[c] - nm_mean[c]) / nm_std[c];
        }
        if (p.outm == 1) {
          if constexpr (UPK > 0) vm_issued += C;  // (lane 0 is active: each of the C stores is certainly issued)
#pragma unroll
          for (int c = 0; c < C; c++)
            if (active && (PL == 0 || c < pl_valid))
              __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[c]), orsrc, (unsigned)(ox0 + lane_o) * 4u,
                                                    ((unsigned)c * (unsigned)p.oH + (unsigned)oy) * (unsigned)p.oW * 4u, AA_V3_F32OUT_AUX);
        } else {
          if constexpr (UPK > 0) vm_issued += 1;
          const unsigned fv = (unsigned)(ox0 + lane_o) * (unsigned)(4 * C), fs = (unsigned)oy * (unsigned)p.oW * (unsigned)(4 * C);
          if constexpr (C == 3) {
            typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
            const u32x3 t = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2])};
            if (active) __builtin_amdgcn_raw_buffer_store_b96(t, orsrc, fv, fs, AA_V3_F32OUT_AUX);
          } else if constexpr (C == 4) {
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 t = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
            if (active) __builtin_amdgcn_raw_buffer_store_b128(t, orsrc, fv, fs, AA_V3_F32OUT_AUX);
          } else {
            if (active) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[0]), orsrc, fv, fs, AA_V3_F32OUT_AUX);
          }
        }
      }
    }
    if (emit_path == 2) {  // (wave-uniform) ragged rows: C byte stores per lane
      int lane_o = lane;
      asm volatile("" : "+v"(lane_o));  // keep this rare path's address arithmetic INSIDE it: hoisted out of the row loop it
                                        // costs the common path 5 VGPRs, i.e. a wave per SIMD (76 -> 81 registers)
      const int px_o = SP > 1 ? lane_o / SP : lane_o;  // (split windows: the quad's first lane stores the pixel)
      const unsigned bv = (unsigned)((ox0 + px_o) * CB);
      const bool act = px_o < bw && (SP == 1 || (lane_o % SP) == 0);
      if constexpr (UPK > 0) vm_issued += C;
#pragma unroll
      for (int c = 0; c < C; c++) {
        const unsigned b = FLT ? trunc8(A[0][c]) : (unsigned)clip8_int(A[0][c]);
        if constexpr (PL > 0) {  // (plane c of the image)
          if (act && c < pl_valid) __builtin_amdgcn_raw_buffer_store_b8((unsigned char)b, orsrc, bv, (unsigned)c * (unsigned)p.plane_out_bytes + (unsigned)oy * out_row_bytes, 0);
        } else {
          if (act) __builtin_amdgcn_raw_buffer_store_b8((unsigned char)b, orsrc, bv + c, (unsigned)oy * out_row_bytes, 0);
        }
      }
    }
    if constexpr (UPK > 0) {
      if (emit_path == 0) vm_issued += 1;  // the dword store below: lane 0 stores whenever the strip holds a whole quad (it does)
    }
    if (emit_path != 0) {
    } else if constexpr (PL > 0) {
      // the lane's PL bytes in one register; per plane, the quad's four bytes are merged into its first lane in two steps (pairs, then
      // the pair of pairs): one DPP move shared by the planes + per plane one byte permute, one DPP move, one byte permute
      const unsigned t = FLT ? (trunc8(A[0][0]) | (trunc8(A[0][1]) << 8) | (trunc8(A[0][PL > 2 ? 2 : 1]) << 16))
                             : pack4_clip8(A[0][0], A[0][1], A[0][PL > 2 ? 2 : 1], A[0][PL - 1]);
      const unsigned nb = (unsigned)__builtin_amdgcn_update_dpp(0, (int)t, 0xF5 /*quad_perm:[1,1,3,3]*/, 0xF, 0xF, false);
#pragma unroll
      for (int c = 0; c < PL; c++) {
        const unsigned pair = __builtin_amdgcn_perm(nb, t, 0x0c0c0000u | (unsigned)((4 + c) << 8) | (unsigned)c);  // [own c, neighbour c, 0, 0]
        const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)pair, 0xAA /*quad_perm:[2,2,2,2]*/, 0xF, 0xF, false);
        const unsigned dw = __builtin_amdgcn_perm(hi, pair, 0x05040100u);
        if (store_lane && c < pl_valid) __builtin_amdgcn_raw_buffer_store_b32(dw, orsrc, store_voff, (unsigned)c * (unsigned)p.plane_out_bytes + (unsigned)oy * out_row_bytes, 0);
      }
    } else if constexpr (C == 3) {
      const unsigned t = FLT ? (trunc8(A[0][0]) | (trunc8(A[0][1]) << 8) | (trunc8(A[0][2]) << 16))
                             : pack4_clip8(A[0][0], A[0][1], A[0][2], A[0][2]);
      const unsigned nb = (unsigned)__builtin_amdgcn_update_dpp(0, (int)t, 0xF9 /*quad_perm:[1,2,3,3]*/, 0xF, 0xF, false);
      const unsigned dw = __builtin_amdgcn_perm(nb, t, perm_sel);
      if (store_lane && (AA_V3_ABL != 1 || dw == 0x12345678u))
        __builtin_amdgcn_raw_buffer_store_b32(dw, orsrc, store_voff, AA_V3_ABL == 9 ? 0u : (unsigned)oy * out_row_bytes, C == 1 ? 0 : kStoreAuxInterleaved);
    } else if constexpr (C == 1) {
      const unsigned t = FLT ? trunc8(A[0][0]) * 0x01010101u : pack4_clip8(A[0][0], A[0][0], A[0][0], A[0][0]);  // the lane's byte, replicated
      const unsigned n1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)t, 0x55 /*quad_perm:[1,1,1,1]*/, 0xF, 0xF, false);
      const unsigned n2 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)t, 0xAA /*quad_perm:[2,2,2,2]*/, 0xF, 0xF, false);
      const unsigned n3 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)t, 0xFF /*quad_perm:[3,3,3,3]*/, 0xF, 0xF, false);
      const unsigned dw = (t & 0x000000ffu) | (n1 & 0x0000ff00u) | (n2 & 0x00ff0000u) | (n3 & 0xff000000u);
      if (store_lane) __builtin_amdgcn_raw_buffer_store_b32(dw, orsrc, store_voff, (unsigned)oy * out_row_bytes, C == 1 ? 0 : kStoreAuxInterleaved);
    } else {
      const unsigned dw = FLT ? (trunc8(A[0][0]) | (trunc8(A[0][1]) << 8) | (trunc8(A[0][2]) << 16) | (trunc8(A[0][C - 1]) << 24))
                              : pack4_clip8(A[0][0], A[0][1], A[0][2], A[0][C - 1]);
      if (store_lane) __builtin_amdgcn_raw_buffer_store_b32(dw, orsrc, store_voff, (unsigned)oy * out_row_bytes, C == 1 ? 0 : kStoreAuxInterleaved);
    }
#pragma unroll
    for (int k = 0; k + 1 < MAXC; k++)
#pragma unroll
      for (int c = 0; c < C; c++) A[k][c] = A[k + 1][c];
#pragma unroll
    for (int c = 0; c < C; c++) A[MAXC - 1][c] = FLT ? 0 : 1 << 21;
  };
  // gather mode: table row of an output = one scalar load of its gather record {ymin, ysize, w[6]}
  struct GRow { int m; int s; int w[RK]; };
  auto load_grow = [&](int oy) -> GRow {
    GRow gr;
    const int o = oy < p.oH ? oy : p.oH - 1;
    const int32_t *rec = (const int32_t *)((const char *)g_rec + (unsigned)o * 32u);
    gr.m = __builtin_amdgcn_readfirstlane(rec[0]);
    gr.s = __builtin_amdgcn_readfirstlane(rec[1]);
#pragma unroll
    for (int k = 0; k < RK; k++) gr.w[k] = __builtin_amdgcn_readfirstlane(rec[2 + (k < 6 ? k : 5)]);
    return gr;
  };
  GRow g_cur, g_nxt;
  if constexpr (UPK > 0) {
    g_cur = load_grow(oy0);
    g_nxt = load_grow(oy0 + 1);
  }
  int r = r_begin;  // the input row being filtered
  // one input row: horizontal pass from the fetched window, then scatter into the open output rows
  // the window of the CURRENT row, realigned; consumes the LDS reads issued one row earlier
  auto realign = [&](const unsigned (&d)[ND], const SA &sa, unsigned (&v)[NV]) {
#pragma unroll
    for (int c = 0; c < PLN; c++)
#pragma unroll
      for (int k = 0; k < NV1; k++)
        v[c * NV1 + k] = AA_V3_UNALIGNED ? d[c * ND1 + k] : __builtin_amdgcn_alignbyte(d[c * ND1 + k + 1], d[c * ND1 + k], sa.s[c]);
  };
  // `steady`: every band starts with a few rows that still feed outputs of the previous band (sc.first < o_base: their weights shift by
  // o_base - sc.first sets); once a row has sc.first == o_base that stays so to the band's end (window ends are non-decreasing).  The
  // unrolled row loop has two forms: the general one, and — taken once `steady` is set — one whose vertical pass has no branch on that
  // shift.  With the branch the compiler copies the untouched accumulator set where its two sides meet: three v_mov per input row
  // (4 % of the kernel's vector instructions) that the steady form does not have.
  bool steady = false;
  auto row_step = [&](const unsigned (&v)[NV], const Scatter &sc, auto steady_tag) {
    constexpr bool STEADY = decltype(steady_tag)::value;
    if (AA_V3_ABL == 6) return;
    int h[C];  // the horizontal-pass result of this row: Pillow's uint8 intermediate, or (FLT) the float's bits
    if constexpr (FLT) {
      float accf[C];
#pragma unroll
      for (int j = 0; j < TW; j++) {
#pragma unroll
        for (int c = 0; c < C; c++) {
          const int bi = PL > 0 ? c * (4 * NV1) + j : j * C + c;  // (plane groups: channel c's own window)
          const float px = (float)((v[bi >> 2] >> (8 * (bi & 3))) & 0xffu);
          if constexpr (AA_V3_FLT_FAST != 0) {
            accf[c] = j == 0 ? px * __int_as_float(wreg[j]) : __builtin_fmaf(px, __int_as_float(wreg[j]), accf[c]);
            continue;
          }
          const float prod = px * __int_as_float(wreg[j]);  // taps outside the window have weight +0.0: adding their
          accf[c] = j == 0 ? prod : accf[c] + prod;          // products never changes a value (bytes are finite)
        }
      }
#pragma unroll
      for (int c = 0; c < C; c++) h[c] = __float_as_int(accf[c]);
    } else {
      int acc[C];
#pragma unroll
      for (int c = 0; c < C; c++) acc[c] = AA_V3_ABL == 5 ? (int)v[c] : (SP > 1 ? acc_init : 1 << 21);
#pragma unroll
      for (int j = 0; j < (AA_V3_ABL == 5 ? 0 : TW); j++) {
#pragma unroll
        for (int c = 0; c < C; c++) {
          const int bi = PL > 0 ? c * (4 * NV1) + j : j * C + c;  // (plane groups: channel c's own window)
          const int px = (int)((v[bi >> 2] >> (8 * (bi & 3))) & 0xffu);
          acc[c] += px * wreg[j];
        }
      }
      if constexpr (SP > 1) {  // the quad's four partial sums: after two exchanges every lane holds their total
#pragma unroll
        for (int c = 0; c < C; c++) {
          acc[c] += __builtin_amdgcn_update_dpp(0, acc[c], 0xB1 /*quad_perm:[1,0,3,2]*/, 0xF, 0xF, false);
          acc[c] += __builtin_amdgcn_update_dpp(0, acc[c], 0x4E /*quad_perm:[2,3,0,1]*/, 0xF, 0xF, false);
        }
      }
#pragma unroll
      for (int c = 0; c < C; c++) h[c] = NONNEG ? (int)((unsigned)acc[c] >> 22) : clip8_int(acc[c]);
    }
    if constexpr (UPK > 0) {
      // push this row's result; then every output row whose window ends here is the sum over the ring's last `ysize` entries
#pragma unroll
      for (int k = 0; k + 1 < RK; k++)
#pragma unroll
        for (int c = 0; c < C; c++) ring[k][c] = ring[k + 1][c];
#pragma unroll
      for (int c = 0; c < C; c++) ring[RK - 1][c] = h[c];
      while (o_base < oy1) {
        int gs = g_cur.s > 1 ? g_cur.s : 1;
        if (g_cur.m + gs != r + 1) break;  // (window ends are non-decreasing: later outputs end later)
        gs = gs < RK ? gs : RK;
#pragma unroll
        for (int ss = 1; ss <= RK; ss++) {
          if (gs == ss) {  // wave-uniform: one static unrolling runs; taps beyond the window are not added at all
#pragma unroll
            for (int c = 0; c < C; c++) {
              int acc;
              if constexpr (FLT) {
                float f = __int_as_float(ring[RK - ss][c]) * __int_as_float(g_cur.w[0]);
#pragma unroll
                for (int k2 = 1; k2 < ss; k2++) f = f + __int_as_float(ring[RK - ss + k2][c]) * __int_as_float(g_cur.w[k2]);
                acc = __float_as_int(f);
              } else {
                acc = (1 << 21) + __mul24(ring[RK - ss][c], g_cur.w[0]);
#pragma unroll
                for (int k2 = 1; k2 < ss; k2++) acc += __mul24(ring[RK - ss + k2][c], g_cur.w[k2]);
              }
              A[0][c] = acc;
            }
          }
        }
        emit(o_base);
        o_base++;
        g_cur = g_nxt;
        g_nxt = load_grow(o_base + 1);
      }
      return;
    }
    auto vmac = [&](int a, int hv, int w) -> int {  // one vertical tap
      if constexpr (FLT && AA_V3_FLT_FAST != 0) return __float_as_int(__builtin_fmaf(__int_as_float(hv), __int_as_float(w), __int_as_float(a)));
      else if constexpr (FLT) return __float_as_int(__int_as_float(a) + __int_as_float(hv) * __int_as_float(w));
      else return a + __mul24(hv, w);
    };
    // accumulator set k is output o_base+k; this row feeds outputs sc.first .. sc.first+MAXC-1 (zero weights beyond)
    const int idx0 = sc.first - o_base;  // 0 in steady state; negative while the band's first rows still feed
                                         // outputs that belong to the previous band
    if constexpr (STEADY) {
#pragma unroll
      for (int k = 0; k < MAXC; k++) {
        if (k >= 2 && k >= (sc.cc & 0xFFFF)) break;
#pragma unroll
        for (int c = 0; c < C; c++) A[k][c] = vmac(A[k][c], h[c], sc.w[k]);
      }
    } else if (__builtin_expect(idx0 == 0, 1)) {
      steady = true;
#pragma unroll
      for (int k = 0; k < MAXC; k++) {
        if (k >= 2 && k >= (sc.cc & 0xFFFF)) break;  // wave-uniform: most rows feed two outputs only (a zero weight in
                                                     // the MIDDLE of the range must not end the loop: test the count)
#pragma unroll
        for (int c = 0; c < C; c++) A[k][c] = vmac(A[k][c], h[c], sc.w[k]);
      }
    } else if (idx0 < 0 && idx0 > -MAXC) {
#pragma unroll
      for (int s = 1; s < MAXC; s++) {
        if (idx0 == -s) {
#pragma unroll
          for (int k = s; k < MAXC; k++)
#pragma unroll
            for (int c = 0; c < C; c++) A[k - s][c] = vmac(A[k - s][c], h[c], sc.w[k]);
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

  // plane groups: the tensor's last 3 bytes (last image, last plane, last row), which the staging DMA's unaligned dwords may have been
  // refused; lanes 0 .. 2 fetch one byte each and put it into the staged row.  Called after the row's DMA has landed, before its
  // window reads.
  const int pl_fix_row = (PL > 0 && (long long)n + 1 == p.n_images) ? p.H - 1 : -1;
  auto pl_patch_last = [&](int slot) {
    if constexpr (PL > 0) {
      const int q = p.W - 3 + lane;          // row position of this lane's byte
      const int lpos = q - seg_first;         // ... inside the strip's segment
      if (lane < 3 && q >= 0 && lpos >= 0 && lpos < p.nseg * 16) {
        const uint8_t b = in[img_off + (unsigned long long)(pl_valid - 1) * p.plane_in_bytes + (unsigned long long)(p.H - 1) * row_bytes + (unsigned)q];
        lds[lds_base + slot * PL_SLOT + (pl_valid - 1) * PL_PLANE + lpos] = b;  // (the tensor's last plane: the last VALID one of the last group)
      }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
  };

  // prologue: the first G rows in flight, window reads of row 0 issued
  for (int i = 0; i < G; i++)
    if (i < n_rows) dma(a + (unsigned)i * row_bytes, i);
  unsigned d0[ND], d1[ND];
  SA sa0 = {}, sa1 = {};
  Scatter sc0 = load_scatter(r_begin), sc1 = sc0;
  {
    const int younger = (n_rows < G ? n_rows : G) - 1;
    wait_vmcnt(younger * dma_per_row);
    if (PL > 0 && r_begin == pl_fix_row) pl_patch_last(0);
    sa0 = fetch(a, 0, d0, false);
  }
  // Invariant at the top of row x (slot x % G): DMAs issued up to row x+G-1; row x's window reads issued into d0 (x
  // even) / d1 (x odd); its scatter record loaded into sc0 / sc1.  Output stores also count in vmcnt: they are
  // younger than every DMA the wait below must cover, so "at most G-2 outstanding" still implies row x+1 has landed
  // (it only waits for a few more rows than strictly necessary).
  for (int g = 0; g < n_groups; g++) {
    const int x0 = g * G;
    // the unrolled group of G rows, in its general and its steady form (see `steady` above)
    auto unrolled_group = [&](auto steady_tag) {
      // Not a data dependence: once per G rows the strips of a band line up, so that the 192-byte pieces they store into
      // the same output rows reach L2 within a few microseconds of each other and leave it as whole lines (measured
      // -2 %; every strip of the workgroup runs the same number of groups).
      if (full_group) __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int i = 0; i < G; i++) {
        // (rows up to x - i % BURST - 1 + G have been issued: G - 2 - i % BURST of them are younger than row x+1)
        if (AA_V3_ABL != 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(dma_per_row * (G - 2 - (i % AA_V3_BURST))) : "memory");
        // Order matters for the lgkm counter (LDS and scalar loads share it and scalar loads return out of order,
        // so every wait is lgkmcnt(0)): first consume the reads issued a whole row ago (no stall), THEN issue the
        // next row's window reads and scatter record, which land while this row's ~45 VALU instructions run.
        unsigned v[NV];
        if ((i & 1) == 0) {
          realign(d0, sa0, v);
          __builtin_amdgcn_sched_barrier(0);
          sc1 = load_scatter(r + 1);
          sa1 = fetch(a + row_bytes, (i + 1) % G, d1);
          __builtin_amdgcn_sched_barrier(0);
          if (AA_V3_DMA_EARLY) { dma(a + (unsigned)G * row_bytes, i); __builtin_amdgcn_sched_barrier(0); }
          row_step(v, sc0, steady_tag);
        } else {
          realign(d1, sa1, v);
          __builtin_amdgcn_sched_barrier(0);
          sc0 = load_scatter(r + 1);
          sa0 = fetch(a + row_bytes, (i + 1) % G, d0);
          __builtin_amdgcn_sched_barrier(0);
          if (AA_V3_DMA_EARLY) { dma(a + (unsigned)G * row_bytes, i); __builtin_amdgcn_sched_barrier(0); }
          row_step(v, sc1, steady_tag);
        }
        if (!AA_V3_DMA_EARLY) {
          if constexpr (AA_V3_BURST == 1) {
            dma(a + (unsigned)G * row_bytes, i);  // the slot just consumed gets row x+G
          } else if ((i % AA_V3_BURST) == AA_V3_BURST - 1) {  // the last BURST slots are free: rows x-BURST+1+G .. x+G
#pragma unroll
            for (int b = AA_V3_BURST - 1; b >= 0; b--) dma(a + (unsigned)(G - b) * row_bytes, i - b);
          }
        }
        a += row_bytes;
        r++;
        __builtin_amdgcn_sched_barrier(0);  // keep the unrolled rows from interleaving: it only costs registers
      }
    };
    if (UPK == 0 && x0 + 2 * G <= n_rows) {  // (lanes beyond the strip compute a duplicate of lane 0 and never store)
      if (steady) unrolled_group(std::true_type{});
      else unrolled_group(std::false_type{});
    } else {
      for (int i = 0; i < G; i++) {
        const int x = x0 + i;
        if (x >= n_rows) break;
        if (x + 1 < n_rows) {
          if constexpr (UPK > 0) {  // everything issued up to and including the DMA of row x+1's slot has completed
            const int my_idx = __builtin_amdgcn_readlane(idxv, __builtin_amdgcn_readfirstlane((i + 1) % G));
            wait_vmcnt(vm_issued - my_idx);
          } else {
            int younger = n_rows - 1 - (x + 1);
            younger = younger < G - 2 ? younger : G - 2;
            wait_vmcnt(younger * dma_per_row);
          }
          if (PL > 0 && r + 1 == pl_fix_row) pl_patch_last((i + 1) % G);  // (never inside the unrolled groups: they end G rows before the band does)
          if ((i & 1) == 0) { sc1 = load_scatter(r + 1); sa1 = fetch(a + row_bytes, (i + 1) % G, d1, false); }
          else { sc0 = load_scatter(r + 1); sa0 = fetch(a + row_bytes, (i + 1) % G, d0, false); }
        }
        unsigned v[NV];
        if ((i & 1) == 0) { realign(d0, sa0, v); row_step(v, sc0, std::false_type{}); }
        else { realign(d1, sa1, v); row_step(v, sc1, std::false_type{}); }
        if (x + G < n_rows) dma(a + (unsigned)G * row_bytes, i);
        a += row_bytes;
        r++;
      }
    }
  }
}

// staged rows per wave: 8; developer builds (-DAA_V2_TUNING) read AA_V3_G
inline int aa_v3_group() {
#ifdef AA_V2_TUNING
  if (const char *e = aa_knob("AA_V3_G")) return atoi(e);
#endif
  return 8;
}

// Row bands: every extra band re-reads and re-filters ~taps_h halo rows, but the grid must fill the chip's resident
// wave slots a near-integer number of times or the last partial round idles most CUs.  Pick the band count
// minimising (1 + halo fraction) / round efficiency.
int pick_ybands(int64_t items_per_band, double slots, int taps_h, int64_t H, int64_t oH) {
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
  if (const char *e = aa_knob("AA_FUSED_YBANDS")) {  // experiment knob
    const int64_t v = atoll(e);
    if (v >= 1 && v <= max_yb) ybands = v;
  }
  return (int)ybands;
}

template <int C, int TW, int G, int MAXC, bool TWO, bool NONNEG, bool PERIODIC, bool FLT = false, int UPK = 0, int PL = 0, int SP = 1>
int launch_k(FusedU8V3Params p, const AAProblem &q, size_t lds, int64_t) {
  auto kern = fused_u8_nhwc_v3_kernel<C, TW, G, TWO, MAXC, NONNEG, PERIODIC, FLT, UPK, PL, SP>;
  auto resident = [&](int s) {  // resident workgroups of s strips per CU for this instantiation and this problem's LDS
    int nb = aa_resident_blocks(kern, 64 * s, lds * s);
    if (nb <= 0) nb = 16 / s;
    return nb < 1 ? 1 : nb;
  };
  // Strips of a band share a workgroup (neighbouring segments share sectors, their stores meet in L2) unless that leaves
  // wave slots of the CU empty: 5 strips -> 4 workgroups = 20 of 24 waves, and 24 single-strip workgroups are 5 % faster;
  // 4 strips fill the CU either way and stay grouped (measured: grouped 0.396 ms vs 0.425 ms for the 906x438 shape).
  int spb = p.strips_per_block;
  if (spb > 1 && !p.spb_forced && resident(1) > resident(spb) * spb) spb = 1;
  p.strips_per_block = spb;
  const int sgroups = (p.nstrips + spb - 1) / spb;
  const size_t lds_blk = lds * spb;
  const int taps_h = q.ah.max_taps > 0 ? q.ah.max_taps : q.ah.ksize;
  p.ybands = pick_ybands(p.n_images * sgroups, (double)aa_device_cu_count() * resident(spb), taps_h, q.H, q.oH);
  if (aa_knob("AA_V3_DEBUG"))
    fprintf(stderr, "[aa v3] strips %d spb %d (resident(1) %d, resident(%d) %d) ybands %d lds/strip %zu images %lld\n", p.nstrips, spb,
            resident(1), p.nstrips <= 8 ? p.nstrips : 4, resident(p.nstrips <= 8 ? p.nstrips : 4), p.ybands, lds, (long long)p.n_images);
  const int64_t groups8 = (p.n_images * (int64_t)p.ybands + 7) / 8 * 8;  // whole rounds of the 8 XCDs (see the kernel)
  const int64_t grid = groups8 * sgroups;
  if (grid > 0x7FFFFFFF) return 0;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * spb), lds_blk,
                     q.stream, (const uint8_t *)q.in - p.in_mis, (uint8_t *)q.out, (const char *)q.aw.table_dev,
                     (const char *)q.ah.table_dev, p);
  AA_HIP_CHECK_LAUNCH();
  return 1;
}

template <int C, int TW, int G, int MAXC>
int launch_gm(const FusedU8V3Params &p, const AAProblem &q, size_t lds, int64_t grid) {
  const bool nonneg = q.aw.filter != AA_FILTER_CUBIC && q.ah.filter != AA_FILTER_CUBIC;
  const bool periodic = ((unsigned long long)G * (unsigned long long)p.row_pitch) % 16 == 0;
  if (p.nseg > 64) {  // wide segments (large down-scales): the generic-address variant only
    return nonneg ? launch_k<C, TW, G, MAXC, true, true, false>(p, q, lds, grid)
                  : launch_k<C, TW, G, MAXC, true, false, false>(p, q, lds, grid);
  }
  if (nonneg) return periodic ? launch_k<C, TW, G, MAXC, false, true, true>(p, q, lds, grid)
                              : launch_k<C, TW, G, MAXC, false, true, false>(p, q, lds, grid);
  return periodic ? launch_k<C, TW, G, MAXC, false, false, true>(p, q, lds, grid)
                  : launch_k<C, TW, G, MAXC, false, false, false>(p, q, lds, grid);
}

template <int C, int TW, int G>
int launch_g(int maxc, const FusedU8V3Params &p, const AAProblem &q, size_t lds, int64_t grid) {
  if (maxc <= 2) return launch_gm<C, TW, G, 2>(p, q, lds, grid);
  if (maxc <= 3) return launch_gm<C, TW, G, 3>(p, q, lds, grid);
  return launch_gm<C, TW, G, 4>(p, q, lds, grid);
}

template <int C, int TW>
int launch(int maxc, const FusedU8V3Params &p, const AAProblem &q, size_t lds, int64_t grid) {
#ifdef AA_V2_TUNING
  if (aa_v3_group() == 4) return launch_g<C, TW, 4>(maxc, p, q, lds, grid);
  if (aa_v3_group() == 6) return launch_g<C, TW, 6>(maxc, p, q, lds, grid);
  if (aa_v3_group() == 10) return launch_g<C, TW, 10>(maxc, p, q, lds, grid);
  if (aa_v3_group() == 12) return launch_g<C, TW, 12>(maxc, p, q, lds, grid);
  if (aa_v3_group() == 16) return launch_g<C, TW, 16>(maxc, p, q, lds, grid);
#endif
  return launch_g<C, TW, 8>(maxc, p, q, lds, grid);
}

// harness (float) arithmetic: the down-scaling window widths only, generic window addressing
template <int C, int TW>
int launch_flt(int maxc, const FusedU8V3Params &p, const AAProblem &q, size_t lds, int64_t grid) {
  const bool two = p.nseg > 64;
  if (maxc <= 2) return two ? launch_k<C, TW, 8, 2, true, false, false, true>(p, q, lds, grid)
                            : launch_k<C, TW, 8, 2, false, false, false, true>(p, q, lds, grid);
  if (maxc <= 3) return two ? launch_k<C, TW, 8, 3, true, false, false, true>(p, q, lds, grid)
                            : launch_k<C, TW, 8, 3, false, false, false, true>(p, q, lds, grid);
  return two ? launch_k<C, TW, 8, 4, true, false, false, true>(p, q, lds, grid)
             : launch_k<C, TW, 8, 4, false, false, false, true>(p, q, lds, grid);
}

template <int C>
int dispatch_tw_flt(int tw, int maxc, const FusedU8V3Params &p, const AAProblem &q, size_t lds, int64_t grid) {
  if (tw <= 6) return launch_flt<C, 6>(maxc, p, q, lds, grid);
  if (tw <= 8) return launch_flt<C, 8>(maxc, p, q, lds, grid);
  if (tw <= 12) return launch_flt<C, 12>(maxc, p, q, lds, grid);
  if (tw <= 16) return launch_flt<C, 16>(maxc, p, q, lds, grid);  // (test.py's 906 -> 120 thumbnails: 16 taps)
  return 0;
}

template <int C>
int dispatch_tw(int tw, int maxc, const FusedU8V3Params &p, const AAProblem &q, size_t lds, int64_t grid) {
#ifdef AA_V3_HEADLINE_ONLY  // developer builds: one window width, so the file compiles in seconds
  return tw == 6 ? launch<C, 6>(maxc, p, q, lds, grid) : 0;
#endif
  if (tw <= 2) return launch<C, 2>(maxc, p, q, lds, grid);
  if (tw <= 4) return launch<C, 4>(maxc, p, q, lds, grid);
  if (tw <= 6) return launch<C, 6>(maxc, p, q, lds, grid);
  if (tw <= 8) return launch<C, 8>(maxc, p, q, lds, grid);
  if (tw <= 12) return launch<C, 12>(maxc, p, q, lds, grid);
  if (tw <= 16) return launch<C, 16>(maxc, p, q, lds, grid);  // (bicubic thumbnails: 1750 -> 500 has 15 taps)
  return 0;
}

// Wide windows (17 .. 34 taps: test.py's 906 -> 120 thumbnails have 17 bilinear / 33 bicubic taps; the reference's loop takes any
// ids_size, s2.2:52-56,81-85): the same kernel with a longer window in registers — Pillow arithmetic, shrinking heights, generic
// window addressing, up to 6 open output rows (what a scatter record holds).  Instantiated in aa_fused_u8_v3_c{1,3,4}w.hip.
template <int C, int TW, int MAXC>
int launch_wide_m(const FusedU8V3Params &p, const AAProblem &q, size_t lds) {
  const bool nonneg = q.aw.filter != AA_FILTER_CUBIC && q.ah.filter != AA_FILTER_CUBIC;
  if (p.nseg > 64) return nonneg ? launch_k<C, TW, 8, MAXC, true, true, false>(p, q, lds, 0) : launch_k<C, TW, 8, MAXC, true, false, false>(p, q, lds, 0);
  return nonneg ? launch_k<C, TW, 8, MAXC, false, true, false>(p, q, lds, 0) : launch_k<C, TW, 8, MAXC, false, false, false>(p, q, lds, 0);
}
template <int C, int TW>
int launch_wide(int maxc, const FusedU8V3Params &p, const AAProblem &q, size_t lds) {
  if (maxc <= 2) return launch_wide_m<C, TW, 2>(p, q, lds);
  if (maxc <= 3) return launch_wide_m<C, TW, 3>(p, q, lds);
  if (maxc <= 4) return launch_wide_m<C, TW, 4>(p, q, lds);
  return launch_wide_m<C, TW, 6>(p, q, lds);
}
template <int C>
int dispatch_tw_wide(int tw, int maxc, const FusedU8V3Params &p, const AAProblem &q, size_t lds) {
  if (tw <= 24) return launch_wide<C, 24>(maxc, p, q, lds);
  if (tw <= 34) return launch_wide<C, 34>(maxc, p, q, lds);
  return 0;
}

// ... and in float arithmetic (the harness's semantics, float32 out): aa_fused_u8_v3_c{1,3,4}wf.hip
template <int C, int TW>
int launch_wide_flt(int maxc, const FusedU8V3Params &p, const AAProblem &q, size_t lds) {
  const bool two = p.nseg > 64;
  if (maxc <= 2) return two ? launch_k<C, TW, 8, 2, true, false, false, true>(p, q, lds, 0) : launch_k<C, TW, 8, 2, false, false, false, true>(p, q, lds, 0);
  if (maxc <= 3) return two ? launch_k<C, TW, 8, 3, true, false, false, true>(p, q, lds, 0) : launch_k<C, TW, 8, 3, false, false, false, true>(p, q, lds, 0);
  if (maxc <= 4) return two ? launch_k<C, TW, 8, 4, true, false, false, true>(p, q, lds, 0) : launch_k<C, TW, 8, 4, false, false, false, true>(p, q, lds, 0);
  return two ? launch_k<C, TW, 8, 6, true, false, false, true>(p, q, lds, 0) : launch_k<C, TW, 8, 6, false, false, false, true>(p, q, lds, 0);
}
template <int C>
int dispatch_tw_wide_flt(int tw, int maxc, const FusedU8V3Params &p, const AAProblem &q, size_t lds) {
  if (tw <= 24) return launch_wide_flt<C, 24>(maxc, p, q, lds);
  if (tw <= 34) return launch_wide_flt<C, 34>(maxc, p, q, lds);
  return 0;
}

// split windows (template parameter SP): 35 .. 136 taps, four lanes per output pixel; instantiated in aa_fused_u8_v3_c{1,3,4}s.hip
template <int C, int TW, int MAXC>
int launch_split_m(const FusedU8V3Params &p, const AAProblem &q, size_t lds) {
  const bool nonneg = q.aw.filter != AA_FILTER_CUBIC && q.ah.filter != AA_FILTER_CUBIC;
  if (p.nseg > 64)
    return nonneg ? launch_k<C, TW, 8, MAXC, true, true, false, false, 0, 0, 4>(p, q, lds, 0) : launch_k<C, TW, 8, MAXC, true, false, false, false, 0, 0, 4>(p, q, lds, 0);
  return nonneg ? launch_k<C, TW, 8, MAXC, false, true, false, false, 0, 0, 4>(p, q, lds, 0) : launch_k<C, TW, 8, MAXC, false, false, false, false, 0, 0, 4>(p, q, lds, 0);
}
template <int C, int TW>
int launch_split(int maxc, const FusedU8V3Params &p, const AAProblem &q, size_t lds) {
  if (maxc <= 2) return launch_split_m<C, TW, 2>(p, q, lds);
  if (maxc <= 3) return launch_split_m<C, TW, 3>(p, q, lds);
  if (maxc <= 4) return launch_split_m<C, TW, 4>(p, q, lds);
  return launch_split_m<C, TW, 6>(p, q, lds);
}
template <int C>
int dispatch_tw_split(int tws, int maxc, const FusedU8V3Params &p, const AAProblem &q, size_t lds) {  // tws: taps per LANE
  if (tws <= 16) return launch_split<C, 16>(maxc, p, q, lds);
  if (tws <= 24) return launch_split<C, 24>(maxc, p, q, lds);
  if (tws <= 34) return launch_split<C, 34>(maxc, p, q, lds);
  return 0;
}

// plane groups (template parameter PL): the three planes of a planar image in one wave; instantiated in aa_fused_u8_v3_c3g.hip
// (templates over the plane count so that only the translation units that name them instantiate the kernels)
template <int PLANES, int TW>
int launch_planes(int maxc, const FusedU8V3Params &p, const AAProblem &q, size_t lds) {
  static_assert(PLANES == 3, "three planes");
  const bool nonneg = q.aw.filter != AA_FILTER_CUBIC && q.ah.filter != AA_FILTER_CUBIC;
  if (maxc <= 2) return nonneg ? launch_k<3, TW, 8, 2, false, true, false, false, 0, 3>(p, q, lds, 0) : launch_k<3, TW, 8, 2, false, false, false, false, 0, 3>(p, q, lds, 0);
  if (maxc <= 3) return nonneg ? launch_k<3, TW, 8, 3, false, true, false, false, 0, 3>(p, q, lds, 0) : launch_k<3, TW, 8, 3, false, false, false, false, 0, 3>(p, q, lds, 0);
  return nonneg ? launch_k<3, TW, 8, 4, false, true, false, false, 0, 3>(p, q, lds, 0) : launch_k<3, TW, 8, 4, false, false, false, false, 0, 3>(p, q, lds, 0);
}
// ... in float arithmetic (harness semantics, float32 planes out): aa_fused_u8_v3_c3gf.hip, and c3gff.hip for the tolerance mode
template <int PLANES, int TW>
int launch_planes_flt(int maxc, const FusedU8V3Params &p, const AAProblem &q, size_t lds) {
  static_assert(PLANES == 3, "three planes");
  if (maxc <= 2) return launch_k<3, TW, 8, 2, false, false, false, true, 0, 3>(p, q, lds, 0);
  if (maxc <= 3) return launch_k<3, TW, 8, 3, false, false, false, true, 0, 3>(p, q, lds, 0);
  return launch_k<3, TW, 8, 4, false, false, false, true, 0, 3>(p, q, lds, 0);
}
template <int PLANES>
int dispatch_tw_planes_flt(int tw, int maxc, const FusedU8V3Params &p, const AAProblem &q, size_t lds) {
  if (maxc > 4) return 0;
  if (tw <= 6) return launch_planes_flt<PLANES, 6>(maxc, p, q, lds);
  if (tw <= 8) return launch_planes_flt<PLANES, 8>(maxc, p, q, lds);
  return 0;  // (12 taps in float arithmetic: 133-141 VGPRs, three waves per SIMD — the single-plane form keeps them)
}
template <int PLANES>
int dispatch_tw_planes(int tw, int maxc, const FusedU8V3Params &p, const AAProblem &q, size_t lds) {
  if (maxc > 4) return 0;
  if (tw <= 4) return launch_planes<PLANES, 4>(maxc, p, q, lds);
  if (tw <= 6) return launch_planes<PLANES, 6>(maxc, p, q, lds);
  if (tw <= 8) return launch_planes<PLANES, 8>(maxc, p, q, lds);
  if (tw <= 12) return launch_planes<PLANES, 12>(maxc, p, q, lds);
  return 0;  // (16 taps: 147 VGPRs, three waves per SIMD — the single-plane form is faster)
}

// growing heights (gather-form vertical pass): generic window addressing, one DMA per row, no scatter accumulators.  Ring of 2
// rows for the triangle / box filters (never negative: the intermediate needs no clamp), of 6 for everything else.
template <int C, int TW>
int launch_up(int upk, bool nonneg, bool flt, const FusedU8V3Params &p, const AAProblem &q, size_t lds) {
  if (flt) {
    if constexpr (TW >= 6 && TW <= 16)
      return upk <= 2 ? launch_k<C, TW, 8, 1, false, false, false, true, 2>(p, q, lds, 0)
                      : launch_k<C, TW, 8, 1, false, false, false, true, 6>(p, q, lds, 0);
    return 0;
  }
  return (nonneg && upk <= 2) ? launch_k<C, TW, 8, 1, false, true, false, false, 2>(p, q, lds, 0)
                              : launch_k<C, TW, 8, 1, false, false, false, false, 6>(p, q, lds, 0);
}

template <int C>
int dispatch_up(int tw, int upk, bool nonneg, bool flt, const FusedU8V3Params &p, const AAProblem &q, size_t lds) {
  if (flt && tw < 6) tw = 6;
  if (tw <= 2) return launch_up<C, 2>(upk, nonneg, flt, p, q, lds);
  if (tw <= 4) return launch_up<C, 4>(upk, nonneg, flt, p, q, lds);
  if (tw <= 6) return launch_up<C, 6>(upk, nonneg, flt, p, q, lds);
  if (tw <= 8) return launch_up<C, 8>(upk, nonneg, flt, p, q, lds);
  if (tw <= 12) return launch_up<C, 12>(upk, nonneg, flt, p, q, lds);
  if (tw <= 16) return launch_up<C, 16>(upk, nonneg, flt, p, q, lds);
  return 0;
}

int round_tw(int taps) {
  const int opts[] = {2, 4, 6, 8, 12, 16, 24, 34};
  for (int o : opts)
    if (taps <= o) return o;
  return 0;
}

}  // namespace

// per-channel-count entry points (one translation unit each)
int aa_v3_launch_c1(int tw, int maxc, bool flt, const FusedU8V3Params &p, const AAProblem &q, size_t lds);
int aa_v3_launch_c3(int tw, int maxc, bool flt, const FusedU8V3Params &p, const AAProblem &q, size_t lds);
int aa_v3_launch_c4(int tw, int maxc, bool flt, const FusedU8V3Params &p, const AAProblem &q, size_t lds);
// float arithmetic in the tolerance mode (aa_fused_u8_v3_c{1,3,4}ff.hip)
int aa_v3_launch_c1ff(int tw, int maxc, const FusedU8V3Params &p, const AAProblem &q, size_t lds);
int aa_v3_launch_c3ff(int tw, int maxc, const FusedU8V3Params &p, const AAProblem &q, size_t lds);
int aa_v3_launch_c4ff(int tw, int maxc, const FusedU8V3Params &p, const AAProblem &q, size_t lds);
// wide windows, 17 .. 34 taps (aa_fused_u8_v3_c{1,3,4}w.hip)
int aa_v3_launch_c1w(int tw, int maxc, const FusedU8V3Params &p, const AAProblem &q, size_t lds);
int aa_v3_launch_c3w(int tw, int maxc, const FusedU8V3Params &p, const AAProblem &q, size_t lds);
int aa_v3_launch_c4w(int tw, int maxc, const FusedU8V3Params &p, const AAProblem &q, size_t lds);
// plane groups: planar images of three channels (aa_fused_u8_v3_c3g.hip)
int aa_v3_launch_c3g(int tw, int maxc, const FusedU8V3Params &p, const AAProblem &q, size_t lds);
int aa_v3_launch_c3gf(int tw, int maxc, const FusedU8V3Params &p, const AAProblem &q, size_t lds);   // float arithmetic
int aa_v3_launch_c3gff(int tw, int maxc, const FusedU8V3Params &p, const AAProblem &q, size_t lds);  // ... in the tolerance mode
// wide windows in float arithmetic (aa_fused_u8_v3_c{1,3,4}wf.hip)
int aa_v3_launch_c1wf(int tw, int maxc, const FusedU8V3Params &p, const AAProblem &q, size_t lds);
int aa_v3_launch_c3wf(int tw, int maxc, const FusedU8V3Params &p, const AAProblem &q, size_t lds);
int aa_v3_launch_c4wf(int tw, int maxc, const FusedU8V3Params &p, const AAProblem &q, size_t lds);
// split windows, 35 .. 136 taps (aa_fused_u8_v3_c{1,3,4}s.hip)
int aa_v3_launch_c1s(int tws, int maxc, const FusedU8V3Params &p, const AAProblem &q, size_t lds);
int aa_v3_launch_c3s(int tws, int maxc, const FusedU8V3Params &p, const AAProblem &q, size_t lds);
int aa_v3_launch_c4s(int tws, int maxc, const FusedU8V3Params &p, const AAProblem &q, size_t lds);
// growing heights (aa_fused_u8_v3_c{1,3,4}u.hip)
int aa_v3_launch_up_c1(int tw, int upk, bool nonneg, bool flt, const FusedU8V3Params &p, const AAProblem &q, size_t lds);
int aa_v3_launch_up_c3(int tw, int upk, bool nonneg, bool flt, const FusedU8V3Params &p, const AAProblem &q, size_t lds);
int aa_v3_launch_up_c4(int tw, int upk, bool nonneg, bool flt, const FusedU8V3Params &p, const AAProblem &q, size_t lds);
