// aa_fused_u8.hip — fused single-launch resample for uint8 channels_last (NHWC) tensors, Pillow arithmetic.
//
// This is the kernel BASELINE.json's metric is quoted on (uint8 channels_last [B,3,438,906] -> [196,320]).  It
// replaces, in one launch and with the intermediate never leaving the CU, what the reference does in two
// TensorIterator passes with a full-size temp in between (step_two_dot_two/aa_interpolation_impl.h:628-683;
// inner loops :29-87; step_three/aa_separable_single_dim_loop2d_impl.h:79-128 is the same arithmetic).
//
// Design (DESIGN.md §Kernels has the roofline arithmetic):
//   * one workgroup = (image, band of output rows, band of output columns); it STREAMS DOWN its input rows once;
//   * horizontal pass: one lane per output pixel.  The lane's xmin/weights live in registers for the whole band
//     (they depend only on ox).  Per input row the lane reads its own ≤(C*taps+3)-byte window straight from
//     HBM/L2 with dword-aligned buffer loads (range-checked by the buffer descriptor, so zero-weight padding taps
//     can never fault), realigns it with v_alignbyte and does C*taps integer MACs (22-bit fixed-point weights,
//     v_mad_i32_i24).  A wave's 64 windows tile a contiguous ~64*scale*C-byte stretch of the row, so every 128-B
//     line is fetched once from HBM and re-served ~(window/stride)x from L1 — no LDS staging, no bank conflicts;
//   * the uint8 intermediate row (Pillow rounds/clips between passes) goes into an LDS ring of 2^k rows;
//   * vertical pass: as soon as the ring holds rows [ymin, ymin+ysize) of the next output row, lanes take one
//     dword (4 interleaved channel values) each, tap weights are wave-uniform scalars, and the finished row is
//     stored with fully coalesced dword stores.  One barrier per output row; the ring depth makes the
//     write-after-read hazard impossible (see ring_rows in the launcher).
// No MFMA: this is a gather-weighted-sum with ~2.3 MAC per input byte, HBM-bound by design.

#include <math.h>
#include <stdlib.h>

#include "aa_common.h"

namespace {

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct FusedU8Params {
  const uint8_t *in;
  uint8_t *out;
  const char *tab_w;
  const char *tab_h;
  int H, W, oH, oW;
  int ksize_w, ksize_h;
  int ybands, xbands;
  int bw;         // output columns per x band (multiple of 4)
  int ring_rows;        // LDS ring depth in rows (any value >= the hazard bound computed by the launcher)
  unsigned ring_magic;  // floor(2^32 / ring_rows) + 1: slot(r) = r - ring_rows * mulhi(r, magic), exact for r < 2^20
  int pitch;      // LDS bytes per ring row (multiple of 16)
  unsigned long long img_in_bytes, img_out_bytes, total_in_bytes;
  int byte_store;  // the output pointer is not dword aligned (a sliced view): four byte stores per lane instead of one dword
};

template <int NDW>
__device__ inline void load_window(__amdgpu_buffer_rsrc_t rsrc, unsigned off, unsigned (&d)[NDW]) {
  int k = 0;
#pragma unroll
  for (; k + 4 <= NDW; k += 4) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 4 * k, 0, 0);
    d[k] = v.x; d[k + 1] = v.y; d[k + 2] = v.z; d[k + 3] = v.w;
  }
  if constexpr (NDW % 4 == 3) {
    const u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(rsrc, off + 4 * k, 0, 0);
    d[k] = v.x; d[k + 1] = v.y; d[k + 2] = v.z;
  } else if constexpr (NDW % 4 == 2) {
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off + 4 * k, 0, 0);
    d[k] = v.x; d[k + 1] = v.y;
  } else if constexpr (NDW % 4 == 1) {
    d[k] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, off + 4 * k, 0, 0);
  }
}

// Pillow's clip8(ss >> PRECISION_BITS) is ONE gfx950 instruction for two values: v_ashr_pk_u8_i32 D, S0, S1, sh
// writes D[7:0] = sat_u8(S0 >> sh), D[15:8] = sat_u8(S1 >> sh) and PRESERVES the other half of D (op_sel[3]=1
// targets D[31:16] instead) — measured on MI355X (tools/microbench/test_pk.hip).  Written as inline asm on purpose:
// ROCm 7.2's hipcc pattern-matches `clip(a)|clip(b)<<8` to this instruction but then treats the preserved upper
// half as zero when the 16-bit result is widened, which corrupts bytes 2-3 of a packed dword (found by the parity
// tests).  VALU results are interlocked in hardware, so no manual wait states are needed around these.
__device__ inline unsigned pack4_clip8(int a0, int a1, int a2, int a3) {
  unsigned d;
  asm("v_ashr_pk_u8_i32 %0, %1, %2, 22\n\tv_ashr_pk_u8_i32 %0, %3, %4, 22 op_sel:[0,0,0,1]"
      : "=&v"(d)
      : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
  return d;
}

// K output rows per barrier; RB input rows of loads in flight per lane.
// UA: byte-unaligned window loads (NV dwords straight at the window's byte offset; correct and ~1.16x faster than
// aligned loads + v_alignbyte on gfx950 / ROCm 7.2, tools/microbench/test_unaligned.hip) vs dword-aligned loads + alignbyte.
template <int C, int TW, int K, int RB, bool UA>
__global__ void __launch_bounds__(1024, 8)
fused_u8_nhwc_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, const char *__restrict__ tab_w,
                     const char *__restrict__ tab_h, const FusedU8Params p) {
  constexpr int NV = (C * TW + 3) / 4;  // aligned dwords holding the window
  constexpr int NDW = UA ? NV : NV + 1;  // dwords fetched (aligned loads may start 1..3 bytes before the window)
  extern __shared__ __attribute__((aligned(16))) uint8_t ring[];

  const int tid = threadIdx.x;
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

  // range-checked view of this image onward (base rounded down to a dword; the remainder joins the lane offset)
  const unsigned long long img_off = (unsigned long long)n * p.img_in_bytes;
  const unsigned long long base_off = img_off & ~3ull;
  unsigned long long remaining = p.total_in_bytes - base_off;
  remaining = (remaining + 3ull) & ~3ull;  // the range check works per dword: serve the last, partial one too
  if (remaining > 0xFFFFFFFCull) remaining = 0xFFFFFFFCull;
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void *)(in + base_off), 0, (unsigned)remaining, 0x00020000);

  // ---- per-lane horizontal-pass state: window start and weights, fixed for the whole band -----------------
  const bool hlane = tid < bw;
  const int ox = ox0 + (hlane ? tid : 0);
  const int xm = xmin_w[ox];
  int xs = xsize_w[ox];
  xs = xs > 1 ? xs : 1;
  // zero-weight padding taps must not run past the end of the row: right-align such windows
  int lead = xm + TW - p.W;
  lead = lead > 0 ? lead : 0;
  const int start = xm - lead;
  int wreg[TW];
#pragma unroll
  for (int j = 0; j < TW; j++) {
    const int src = j - lead;
    int w = (src >= 0 && src < xs && src < p.ksize_w) ? kw[(size_t)ox * p.ksize_w + src] : 0;
    wreg[j] = (w << 8) >> 8;  // |w| <= 2^22: tell the compiler it is a 24-bit operand (v_mad_i32_i24)
  }
  const unsigned lane_off = (unsigned)(img_off - base_off) + (unsigned)start * C;
  const unsigned row_bytes = (unsigned)p.W * C;
  uint8_t *const ring_lane = ring + tid * C;
  // wave-uniform ring slot of input row r (scalar multiply-high; rows are < 2^20)
  auto slot_of = [&](int r) -> int {
    return r - p.ring_rows * (int)__umulhi((unsigned)r, p.ring_magic);
  };
  const int nd = (bw * C) >> 2;
  unsigned *const out_img = (unsigned *)(out + (unsigned long long)n * p.img_out_bytes) + tid;

  int r_done = __builtin_amdgcn_readfirstlane(ymin_h[oy0]);
  for (int oyc = oy0; oyc < oy1; oyc += K) {
    const int oye = min(oyc + K, oy1);
    const int ylast_min = __builtin_amdgcn_readfirstlane(ymin_h[oye - 1]);
    const int ylast_sz = __builtin_amdgcn_readfirstlane(ysize_h[oye - 1]);
    const int r_end = ylast_min + (ylast_sz > 1 ? ylast_sz : 1);
    const int yfirst = __builtin_amdgcn_readfirstlane(ymin_h[oyc]);
    if (r_done < yfirst) r_done = yfirst;

    // ---- horizontal pass over the input rows this chunk of output rows still needs ------------------------
    if (hlane) {
      for (int r = r_done; r < r_end; r += RB) {
        unsigned d[RB][NDW];
#pragma unroll
        for (int i = 0; i < RB; i++) {
          if (r + i < r_end) {
            const unsigned off = lane_off + (unsigned)(r + i) * row_bytes;
            load_window<NDW>(rsrc, UA ? off : (off & ~3u), d[i]);
          }
        }
#pragma unroll
        for (int i = 0; i < RB; i++) {
          if (r + i < r_end) {
            const unsigned sh = (lane_off + (unsigned)(r + i) * row_bytes) & 3u;
            unsigned v[NV];
            if constexpr (UA) {
#pragma unroll
              for (int k = 0; k < NV; k++) v[k] = d[i][k];
            } else {
#pragma unroll
              for (int k = 0; k < NV; k++) v[k] = __builtin_amdgcn_alignbyte(d[i][k + 1], d[i][k], sh);
            }
            int acc[C];
#pragma unroll
            for (int c = 0; c < C; c++) acc[c] = 1 << 21;
#pragma unroll
            for (int j = 0; j < TW; j++) {
#pragma unroll
              for (int c = 0; c < C; c++) {
                const int bi = j * C + c;
                const int px = (int)((v[bi >> 2] >> (8 * (bi & 3))) & 0xffu);
                acc[c] += px * wreg[j];
              }
            }
            uint8_t *dst = ring_lane + (size_t)slot_of(r + i) * p.pitch;
            if constexpr (C == 4) {
              *(unsigned *)dst = pack4_clip8(acc[0], acc[1], acc[2], acc[3]);
            } else if constexpr (C == 3) {
              // bytes 0 and 2 of one register + byte 0 of another: three byte stores, no shifts
              const unsigned d02 = pack4_clip8(acc[0], acc[0], acc[1], acc[1]);
              const unsigned d2 = pack4_clip8(acc[2], acc[2], acc[2], acc[2]);
              dst[0] = (uint8_t)d02;
              dst[1] = (uint8_t)(d02 >> 16);
              dst[2] = (uint8_t)d2;
            } else {
              dst[0] = (uint8_t)pack4_clip8(acc[0], acc[0], acc[0], acc[0]);
            }
          }
        }
      }
    }
    if (r_done < r_end) r_done = r_end;
    __syncthreads();

    // ---- vertical pass: one dword (4 interleaved channel values) per lane, tap weights wave-uniform --------
    if (tid < nd) {
      for (int oy = oyc; oy < oye; oy++) {
        const int ym = __builtin_amdgcn_readfirstlane(ymin_h[oy]);
        int ys = __builtin_amdgcn_readfirstlane(ysize_h[oy]);
        ys = ys > 1 ? ys : 1;
        int a0 = 1 << 21, a1 = 1 << 21, a2 = 1 << 21, a3 = 1 << 21;
        const int32_t *wrow = kh + (size_t)oy * p.ksize_h;
        for (int j = 0; j < ys; j++) {
          int w = __builtin_amdgcn_readfirstlane(wrow[j]);
          w = (w << 8) >> 8;
          const unsigned dw = *(const unsigned *)(ring + (size_t)slot_of(ym + j) * p.pitch + 4 * tid);
          a0 += (int)(dw & 0xffu) * w;
          a1 += (int)((dw >> 8) & 0xffu) * w;
          a2 += (int)((dw >> 16) & 0xffu) * w;
          a3 += (int)(dw >> 24) * w;
        }
        const unsigned o = pack4_clip8(a0, a1, a2, a3);
        unsigned *const dst = out_img + ((((size_t)oy * p.oW + ox0) * C) >> 2);
        if (p.byte_store) {  // (wave-uniform)
          uint8_t *const db = (uint8_t *)dst;
          db[0] = (uint8_t)o; db[1] = (uint8_t)(o >> 8); db[2] = (uint8_t)(o >> 16); db[3] = (uint8_t)(o >> 24);
        } else {
          *dst = o;
        }
      }
    }
    // no second barrier: the next chunk's horizontal pass writes rows >= r_end, whose ring slots cannot alias the
    // rows still being read here because ring_rows >= taps_h + 2*K*ceil(scale_h) + 2 (launcher).
  }
}

constexpr int kRowsPerBarrier = 4;  // K
// RB: input rows whose window loads are in flight per lane before the first is consumed
constexpr int rows_in_flight(int tw) { return tw <= 4 ? 8 : (tw <= 6 ? 6 : (tw <= 8 ? 4 : 2)); }

template <int C, int TW>
int launch(const FusedU8Params &p, int block, size_t lds, int64_t grid, hipStream_t stream) {
  // dword-aligned window loads + v_alignbyte.  Byte-unaligned loads (UA = true) measured SLOWER inside this kernel
  // (0.54 vs 0.43 ms) and lose the valid bytes of a dword that straddles the end of the tensor (the buffer range
  // check zeroes the whole dword) — kept only as a template parameter for experiments.
  hipLaunchKernelGGL((fused_u8_nhwc_kernel<C, TW, kRowsPerBarrier, rows_in_flight(TW), false>), dim3((unsigned)grid),
                     dim3(block), lds, stream, p.in, p.out, p.tab_w, p.tab_h, p);
  AA_HIP_CHECK_LAUNCH();
  return 1;
}

template <int C>
int dispatch_tw(int tw, const FusedU8Params &p, int block, size_t lds, int64_t grid, hipStream_t stream) {
  if (tw <= 2) return launch<C, 2>(p, block, lds, grid, stream);
  if (tw <= 4) return launch<C, 4>(p, block, lds, grid, stream);
  if (tw <= 6) return launch<C, 6>(p, block, lds, grid, stream);
  if (tw <= 8) return launch<C, 8>(p, block, lds, grid, stream);
  if (tw <= 12) return launch<C, 12>(p, block, lds, grid, stream);
  if (tw <= 16) return launch<C, 16>(p, block, lds, grid, stream);
  return 0;
}

int round_tw(int taps) {
  const int opts[] = {2, 4, 6, 8, 12, 16};
  for (int o : opts)
    if (taps <= o) return o;
  return 0;
}

int next_pow2(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

}  // namespace

static void v1_geometry(int64_t C, int64_t H, int64_t oW, const aa_axis &ah, int *xbands_o, int *bw_o, int *ring_rows_o, int *pitch_o) {
  // column bands: at most 1024 lanes, a multiple of 4 columns so every band starts dword-aligned
  int xbands = (int)((oW + 1023) / 1024);
  int bw = (int)((oW + xbands - 1) / xbands);
  bw = (bw + 3) & ~3;
  xbands = (int)((oW + bw - 1) / bw);
  // ring depth: while slow waves still read chunk i's rows [ymin(first oy of chunk i), r_end(i)), fast waves may
  // already write chunk i+1's rows [r_end(i), r_end(i+1)): span <= taps_h + 2*K*max(scale_h,1) (+ rounding slack)
  const int taps_h = ah.max_taps > 0 ? ah.max_taps : ah.ksize;
  const double scale_h = (double)H / (double)ah.out_size;
  *ring_rows_o = taps_h + (int)(2.0 * kRowsPerBarrier * (scale_h > 1.0 ? scale_h : 1.0) + 0.999) + 4;
  *pitch_o = ((bw * (int)C + 15) / 16) * 16;
  *xbands_o = xbands;
  *bw_o = bw;
}

bool aa_fused_u8_nhwc_applicable(int dtype, int layout, int64_t N, int64_t C, int64_t H, int64_t W, const aa_axis *ah,
                                 const aa_axis *aw) {
  if (dtype != AA_U8 || layout != AA_NHWC) return false;
  if (!ah || !aw || ah->kind != AA_TABLE_PIL || aw->kind != AA_TABLE_PIL) return false;
  if (C != 1 && C != 3 && C != 4) return false;
  const int taps_w = aw->max_taps > 0 ? aw->max_taps : aw->ksize;
  const int tw = round_tw(taps_w);
  if (tw == 0 || W < tw) return false;
  if ((aw->out_size * C) % 4 != 0) return false;        // rows stored as whole dwords
  if ((uint64_t)H * W * C > 0xFFFFFFF0ull) return false;  // 32-bit offsets inside one image
  const int taps_h = ah->max_taps > 0 ? ah->max_taps : ah->ksize;
  if (taps_h > 64) return false;
  if (H >= (1 << 20)) return false;
  int xbands, bw, ring_rows, pitch;
  v1_geometry(C, H, aw->out_size, *ah, &xbands, &bw, &ring_rows, &pitch);
  if ((size_t)ring_rows * pitch > 64 * 1024) return false;  // the intermediate ring must fit the workgroup's LDS
  if (!aa_grid_fits(N * xbands)) return false;
  return true;
}

int aa_try_fused_u8_nhwc(const AAProblem &q, const char **variant) {
  if (!aa_fused_u8_nhwc_applicable(q.dtype, q.layout, q.N, q.C, q.H, q.W, &q.ah, &q.aw)) return 0;
  const int C = (int)q.C;
  const int taps_w = q.aw.max_taps > 0 ? q.aw.max_taps : q.aw.ksize;
  const int taps_h = q.ah.max_taps > 0 ? q.ah.max_taps : q.ah.ksize;
  const int tw = round_tw(taps_w);

  FusedU8Params p;
  p.in = (const uint8_t *)q.in;
  p.out = (uint8_t *)q.out;
  p.tab_w = (const char *)q.aw.table_dev;
  p.tab_h = (const char *)q.ah.table_dev;
  p.H = (int)q.H; p.W = (int)q.W; p.oH = (int)q.oH; p.oW = (int)q.oW;
  p.ksize_w = q.aw.ksize; p.ksize_h = q.ah.ksize;
  p.img_in_bytes = (unsigned long long)q.H * q.W * C;
  p.img_out_bytes = (unsigned long long)q.oH * q.oW * C;
  p.total_in_bytes = p.img_in_bytes * (unsigned long long)q.N;
  p.byte_store = ((uintptr_t)q.out & 3) != 0 ? 1 : 0;  // (no pointer-dependent decline: aa_workspace_bytes answered 0 from the shape alone)

  int xbands, bw, ring_rows, pitch;
  v1_geometry(q.C, q.H, q.oW, q.ah, &xbands, &bw, &ring_rows, &pitch);
  const int block = ((bw + 63) / 64) * 64;
  p.ring_rows = ring_rows;
  p.ring_magic = (unsigned)(0x100000000ull / (unsigned)ring_rows) + 1u;
  p.pitch = pitch;
  const size_t lds = (size_t)ring_rows * p.pitch;

  // row bands.  Every extra band re-reads and re-filters ~taps_h halo rows, but the grid must fill the chip's
  // resident-workgroup slots a near-integer number of times or the last partial round idles most CUs
  // (2048 workgroups on 1280 slots ran 2 rounds for 1.6 rounds of work).  Pick the band count minimising
  // (1 + halo fraction) / round efficiency.
  const int cus = aa_device_cu_count();
  const int waves_per_block = block / 64;
  int blocks_per_cu = (int)((160 * 1024) / (lds > 0 ? lds : 1));
  if (blocks_per_cu > 32 / waves_per_block) blocks_per_cu = 32 / waves_per_block;
  if (blocks_per_cu > 8) blocks_per_cu = 8;
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
  if (const char *e = aa_knob("AA_FUSED_YBANDS")) {  // tuning knob for experiments; not used by tests or bench
    const int64_t v = atoll(e);
    if (v >= 1 && v <= max_yb) ybands = v;
  }
  p.ybands = (int)ybands;
  p.xbands = xbands;
  p.bw = bw;
  const int64_t grid = q.N * ybands * xbands;
  if (grid > 0x7FFFFFFF) return 0;

  int rc;
  if (C == 3) rc = dispatch_tw<3>(tw, p, block, lds, grid, q.stream);
  else if (C == 4) rc = dispatch_tw<4>(tw, p, block, lds, grid, q.stream);
  else rc = dispatch_tw<1>(tw, p, block, lds, grid, q.stream);
  if (rc == 1) *variant = "fused_u8_nhwc_pil";
  return rc;
}
