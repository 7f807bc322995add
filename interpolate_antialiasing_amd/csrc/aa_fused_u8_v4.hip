// aa_fused_u8_v4.hip — fused resample for uint8 channels_last, Pillow arithmetic: v3's arithmetic behind a DMA wave.
//
// What the counters said about v3: its waves never wait for their staging DMAs (removing the vmcnt
// waits changes nothing) and yet real HBM traffic costs 0.10 ms over the same kernel fed from cache.  The time goes
// into ISSUING the DMAs: the per-CU texture-address FIFO is full 28 % of the time (SQ_VMEM_TA_*_FIFO_FULL), a wave
// that reaches its `buffer_load ... lds` then stalls in order, and its ~50 VALU instructions per row stall with it.
// v4 takes the memory instructions out of the arithmetic waves:
//   * a workgroup = the S strips (waves) of one band of one image + ONE PRODUCER WAVE;
//   * the producer streams whole input rows (not per-strip segments: no duplicated sectors, 3 instead of 5 DMAs per
//     906-pixel row) into a ring of G row slots shared by the strips, up to G-1 rows ahead, and is the only wave that
//     ever blocks on the memory pipeline;
//   * flow control is two kinds of LDS words, no barriers in the row loop: `landed` (rows complete in LDS, written by
//     the producer after a counted vmcnt wait) and `consumed[s]` (rows whose window reads have returned, written by
//     strip s).  A strip polls `landed` one row ahead (the value it needs is normally already there), the producer
//     polls the `consumed` words before it reuses a slot;
//   * the arithmetic (horizontal pass from LDS with v_alignbyte + SDWA multiplies, vertical pass in registers in
//     scatter form, DPP+perm quad merge, dword stores) is v3's, bit for bit.
// Every spin loop is bounded: a wave that waits implausibly long gives up (wrong output, caught by the tests) rather
// than hanging the GPU.

#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "aa_common.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void;

#ifndef AA_V4_G
#define AA_V4_G 12
#endif
constexpr int kG = AA_V4_G;      // ring slots (rows)
constexpr int kSyncBytes = 64;   // landed @0, consumed[s] @16+4s
constexpr int kSpinLimit = 1 << 18;
#ifndef AA_V4_ABL
#define AA_V4_ABL 0  // developer ablations (wrong results): 1 strips ignore `landed`, 2 producer ignores `consumed`, 3 both
#endif
#ifndef AA_V4_LAG
#define AA_V4_LAG 4
#endif
constexpr int kLag = AA_V4_LAG;       // rows in flight before the producer blocks on the oldest one (< kG - 1)

struct FusedU8V4Params {
  int H, W, oH, oW;
  int ksize_w, ksize_h;
  int ybands, nstrips;
  int strip_w;     // output columns per strip (<= 64, multiple of 4)
  int pitch;       // bytes per ring slot (multiple of 16)
  int nseg;        // 16-byte pieces the producer copies per row
  int ndma;        // DMA instructions per row = ceil(nseg / 64)
  int sc_off;      // scatter section of the H table
  int in_mis;      // (input pointer & 15): the kernel gets the pointer rounded down to 16 B
  unsigned long long img_in_bytes, img_out_bytes, total_in_bytes, total_out_bytes;
};

__device__ inline unsigned pack4_clip8_v4(int a0, int a1, int a2, int a3) {  // semantics: see aa_fused_u8.hip
  unsigned d;
  asm("v_ashr_pk_u8_i32 %0, %1, %2, 22\n\tv_ashr_pk_u8_i32 %0, %3, %4, 22 op_sel:[0,0,0,1]"
      : "=&v"(d)
      : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
  return d;
}

__device__ inline int clip8_int_v4(int acc) {
  acc >>= 22;
  return acc < 0 ? 0 : (acc > 255 ? 255 : acc);
}

__device__ inline void wait_vmcnt_exact(int n) {  // n in [0, 63]
#define AA_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
  switch (n) {
    AA_W(0) AA_W(1) AA_W(2) AA_W(3) AA_W(4) AA_W(5) AA_W(6) AA_W(7) AA_W(8) AA_W(9) AA_W(10) AA_W(11) AA_W(12) AA_W(13)
    AA_W(14) AA_W(15) AA_W(16) AA_W(17) AA_W(18) AA_W(19) AA_W(20) AA_W(21) AA_W(22) AA_W(23) AA_W(24) AA_W(25) AA_W(26)
    AA_W(27) AA_W(28) AA_W(29) AA_W(30) AA_W(31) AA_W(32) AA_W(33) AA_W(34) AA_W(35) AA_W(36) AA_W(37) AA_W(38) AA_W(39)
    AA_W(40) AA_W(41) AA_W(42) AA_W(43) AA_W(44) AA_W(45) AA_W(46) AA_W(47) AA_W(48) AA_W(49) AA_W(50) AA_W(51) AA_W(52)
    AA_W(53) AA_W(54) AA_W(55) AA_W(56)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
#undef AA_W
}

typedef __attribute__((address_space(3))) int lds_vint;
// Flow-control words are read and written with relaxed workgroup-scope atomics: plain ds_read/ds_write instructions
// the compiler neither caches nor merges.  (`volatile` would do that too, but the AMDGPU backend follows every volatile
// LDS access with s_waitcnt lgkmcnt(0), which serialises the window reads issued next to it.)
__device__ inline int sync_load(lds_vint *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ inline void sync_store(lds_vint *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

template <int C, int TW, int MAXC, bool NONNEG>
__global__ void __launch_bounds__(512)
fused_u8_nhwc_v4_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, const char *__restrict__ tab_w,
                        const char *__restrict__ tab_h, const FusedU8V4Params p) {
  constexpr int NV = (C * TW + 3) / 4;  // dwords holding one window
  constexpr int ND = NV + 1;            // aligned dwords fetched per window
  constexpr int G = kG;
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];

  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int yb = blockIdx.x % p.ybands;
  const int n = blockIdx.x / p.ybands;
  const int oy0 = (int)((long long)yb * p.oH / p.ybands);
  const int oy1 = (int)((long long)(yb + 1) * p.oH / p.ybands);

  const int32_t *__restrict__ ymin_h = (const int32_t *)(tab_h + aa_table_xmin_off());
  const int32_t *__restrict__ ysize_h = (const int32_t *)(tab_h + aa_table_xsize_off(p.oH));
  const int r_begin = __builtin_amdgcn_readfirstlane(ymin_h[oy0]);
  const int ylm = __builtin_amdgcn_readfirstlane(ymin_h[oy1 - 1]);
  const int yls = __builtin_amdgcn_readfirstlane(ysize_h[oy1 - 1]);
  const int r_stop = ylm + (yls > 1 ? yls : 1);
  const int n_rows = r_stop - r_begin;

  // flow-control words start at zero
  lds_vint *sync_w = (lds_vint *)(uintptr_t)0;
  if (threadIdx.x < kSyncBytes / 4) sync_store(sync_w + threadIdx.x, 0);
  __syncthreads();

  const unsigned long long img_off = (unsigned long long)p.in_mis + (unsigned long long)n * p.img_in_bytes;
  const unsigned long long base_off = img_off & ~15ull;
  const unsigned row_bytes = (unsigned)p.W * C;
  // byte offset (from the descriptor base) of the first byte of input row r_begin
  const unsigned a0 = (unsigned)(img_off - base_off) + (unsigned)r_begin * row_bytes;

  if (wv == p.nstrips) {
    // ===================================== producer wave ==========================================================
#if AA_V4_ABL == 5
    return;  // strips alone, on whatever the LDS holds
#endif
    unsigned long long remaining = p.total_in_bytes - base_off;
    remaining = (remaining + 3ull) & ~3ull;  // the range check works per dword: serve the last, partial one too
    if (remaining > 0xFFFFFFFCull) remaining = 0xFFFFFFFCull;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void *)(in + base_off), 0, (unsigned)remaining, 0x00020000);
    const unsigned voff = (unsigned)lane * 16u;
    int issued = 0, landed = 0, slot_issue = 0;
    unsigned a_issue = a0;
    int max_flight = 56 / p.ndma + 1;  // vmcnt is a 6-bit counter
    max_flight = max_flight < G - 1 ? max_flight : G - 1;
    int guard = 0;
    while (landed < n_rows) {
      // issue rows while the ring has a free slot and fewer than G-1 rows are in flight
      while (issued < n_rows && issued - landed < max_flight) {
        if (issued >= G && AA_V4_ABL != 2 && AA_V4_ABL != 3 && AA_V4_ABL != 4) {
          // slot (issued % G) still holds row issued-G: every strip must have finished reading it
          const int need = issued - G + 1;
          const int c = (lane < p.nstrips) ? sync_load(sync_w + 4 + lane) : 0x7fffffff;
          const bool ok = __builtin_amdgcn_ballot_w64(c < need) == 0ull;
          if (!ok) break;
        }
        const unsigned soff = a_issue & ~15u;
        const int dst = kSyncBytes + slot_issue * p.pitch;
        for (int i = 0; i < p.ndma; i++) {
          if (i * 64 + lane < p.nseg)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void *)(lds + dst + i * 1024), 16, voff + (unsigned)i * 1024u, soff,
                                                     0, 0);
        }
        a_issue += row_bytes;
        issued++;
        slot_issue = slot_issue + 1 == G ? 0 : slot_issue + 1;
      }
      // Waiting for the oldest row in flight BLOCKS this wave, and while it is blocked it cannot refill slots the strips
      // free.  So: block only when that row was issued long ago (kLag rows in flight: it has almost certainly landed) or
      // when there is nothing left to issue; otherwise poll the `consumed` words.  `landed` then trails `issued` by
      // fewer than kLag rows, and the strips read G - 1 - kLag rows behind the newest issued row at the closest.
      const int in_flight = issued - landed;
      if (in_flight >= kLag || (in_flight > 0 && issued == n_rows)) {
        // the oldest row in flight is complete once at most ndma * (rows younger than it) DMAs are outstanding
        wait_vmcnt_exact(p.ndma * (in_flight - 1));
        landed++;
        if (lane == 0) sync_store(sync_w, landed);
      } else {
        __builtin_amdgcn_s_sleep(1);
        if (++guard > kSpinLimit) break;
      }
    }
    return;
  }

  // ======================================= strip (consumer) waves ===================================================
#if AA_V4_ABL == 4
  return;  // producer alone (it ignores `consumed`)
#endif
  const int strip = wv;
  const int ox0 = strip * p.strip_w;
  const int bw = min(p.strip_w, p.oW - ox0);
  const int32_t *__restrict__ xmin_w = (const int32_t *)(tab_w + aa_table_xmin_off());
  const int32_t *__restrict__ xsize_w = (const int32_t *)(tab_w + aa_table_xsize_off(p.oW));
  const int32_t *__restrict__ kw = (const int32_t *)(tab_w + aa_table_w_off(p.oW));
  const int32_t *__restrict__ sc_rec = (const int32_t *)(tab_h + p.sc_off);

  const bool active = lane < bw;
  const int ox = ox0 + (active ? lane : 0);
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
  const unsigned lane_lds = (unsigned)(kSyncBytes + start * C);  // window offset inside a ring slot, before the row phase

  const unsigned long long out_off = (unsigned long long)n * p.img_out_bytes;
  unsigned long long out_rem = p.total_out_bytes - out_off;
  if (out_rem > 0xFFFFFFFFull) out_rem = 0xFFFFFFFFull;
  const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(out + out_off), 0, (unsigned)out_rem, 0x00020000);
  const unsigned out_row_bytes = (unsigned)p.oW * C;
  unsigned store_voff;
  bool store_lane;
  unsigned perm_sel = 0;
  if constexpr (C == 3) {
    const int q = lane & 3;
    store_voff = (unsigned)(ox0 * 3 + (lane >> 2) * 12 + q * 4);
    store_lane = (q != 3) && ((lane | 3) < bw);
    perm_sel = q == 0 ? 0x04020100u : (q == 1 ? 0x05040201u : 0x06050402u);
  } else {
    store_voff = (unsigned)((ox0 + lane) * 4);
    store_lane = active;
  }

  int A[MAXC][C];
#pragma unroll
  for (int k = 0; k < MAXC; k++)
#pragma unroll
    for (int c = 0; c < C; c++) A[k][c] = 1 << 21;
  int o_base = oy0;

  struct Scatter { int first; int cc; int w[MAXC]; };  // raw record words: nothing may depend on them until they are used
  auto load_scatter = [&](int r) -> Scatter {  // one 32-byte record: {first, count | completes << 16, w[6]}
    Scatter s;
    const int32_t *rec = (const int32_t *)((const char *)sc_rec + (unsigned)r * 32u);
    s.first = __builtin_amdgcn_readfirstlane(rec[0]);
    s.cc = __builtin_amdgcn_readfirstlane(rec[1]);  // count | completes << 16
#pragma unroll
    for (int k = 0; k < MAXC; k++) s.w[k] = __builtin_amdgcn_readfirstlane(rec[2 + k]);
    return s;
  };
  auto emit = [&](int oy) {
    if constexpr (C == 3) {
      const unsigned t = pack4_clip8_v4(A[0][0], A[0][1], A[0][2], A[0][2]);
      const unsigned nb = (unsigned)__builtin_amdgcn_update_dpp(0, (int)t, 0xF9 /*quad_perm:[1,2,3,3]*/, 0xF, 0xF, false);
      const unsigned dw = __builtin_amdgcn_perm(nb, t, perm_sel);
      if (store_lane) __builtin_amdgcn_raw_buffer_store_b32(dw, orsrc, store_voff, (unsigned)oy * out_row_bytes, 0);
    } else {
      const unsigned dw = pack4_clip8_v4(A[0][0], A[0][1], A[0][2], A[0][C - 1]);
      if (store_lane) __builtin_amdgcn_raw_buffer_store_b32(dw, orsrc, store_voff, (unsigned)oy * out_row_bytes, 0);
    }
#pragma unroll
    for (int k = 0; k + 1 < MAXC; k++)
#pragma unroll
      for (int c = 0; c < C; c++) A[k][c] = A[k + 1][c];
#pragma unroll
    for (int c = 0; c < C; c++) A[MAXC - 1][c] = 1 << 21;
  };
  auto realign = [&](const unsigned (&d)[ND], unsigned sa, unsigned (&v)[NV]) {
#pragma unroll
    for (int k = 0; k < NV; k++) v[k] = __builtin_amdgcn_alignbyte(d[k + 1], d[k], sa);
  };
  auto row_step = [&](const unsigned (&v)[NV], const Scatter &sc) {
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
    int h[C];
#pragma unroll
    for (int c = 0; c < C; c++) h[c] = NONNEG ? (int)((unsigned)acc[c] >> 22) : clip8_int_v4(acc[c]);
    const int idx0 = sc.first - o_base;
    if (__builtin_expect(idx0 == 0, 1)) {
#pragma unroll
      for (int k = 0; k < MAXC; k++) {
        if (k >= 2 && sc.w[k] == 0) break;
#pragma unroll
        for (int c = 0; c < C; c++) A[k][c] += __mul24(h[c], sc.w[k]);
      }
    } else if (idx0 < 0 && idx0 > -MAXC) {
#pragma unroll
      for (int s = 1; s < MAXC; s++) {
        if (idx0 == -s) {
#pragma unroll
          for (int k = s; k < MAXC; k++)
#pragma unroll
            for (int c = 0; c < C; c++) A[k - s][c] += __mul24(h[c], sc.w[k]);
        }
      }
    }
    const int sc_end = sc.first + (sc.cc >> 16);  // outputs [first, sc_end) take their last row here
    const int e_end = sc_end < oy1 ? sc_end : oy1;
    while (o_base < e_end) {
      emit(o_base);
      o_base++;
    }
  };

  // window reads of ring row x (global row r_begin + x): slot x % G, 16-byte phase of the row's first byte
  unsigned a_row = a0;
  auto fetch = [&](int slot, unsigned a_r, unsigned (&d)[ND]) -> unsigned {
    const unsigned s_off = (unsigned)(slot * p.pitch) + (a_r & 15u);  // uniform
    const unsigned sa = lane_lds + s_off;
    const unsigned ra = sa & ~3u;
    const __attribute__((address_space(3))) unsigned *al = (const __attribute__((address_space(3))) unsigned *)(uintptr_t)ra;
#pragma unroll
    for (int k = 0; k < ND; k++) d[k] = al[k];
    return sa;
  };
  int landed_seen = 0;
  auto need_landed = [&](int cnt) {  // rows [0, cnt) must be in the ring
#if AA_V4_ABL == 1 || AA_V4_ABL == 3 || AA_V4_ABL == 5
    return;
#endif
    int guard = 0;
    while (landed_seen < cnt) {
      landed_seen = __builtin_amdgcn_readfirstlane(sync_load(sync_w));
      if (landed_seen >= cnt) break;
      __builtin_amdgcn_s_sleep(1);
      if (++guard > kSpinLimit) break;
    }
    asm volatile("" ::: "memory");  // the window reads below must not move above the poll
  };
  lds_vint *my_consumed = sync_w + 4 + strip;

  unsigned d0[ND], d1[ND];
  unsigned sa0 = 0, sa1 = 0;
  Scatter sc0 = load_scatter(r_begin), sc1 = sc0;
  need_landed(1);
  sa0 = fetch(0, a_row, d0);
  int slot_next = G > 1 ? 1 : 0;  // ring slot of row x + 1
  int lv = 0;  // refreshed copy of `landed`, read one row ahead of its use
  int r = r_begin;
  // one row: consume the window reads issued a row ago, issue the next row's, publish progress, do the arithmetic
  auto half = [&](int x, const unsigned (&dc)[ND], unsigned sac, const Scatter &scc, unsigned (&dn)[ND], unsigned &san,
                  Scatter &scn) {
    unsigned v[NV];
    realign(dc, sac, v);  // (waits for the reads of row x: its ring slot may be recycled from here on)
#pragma unroll
    for (int k = 0; k < NV; k++) asm volatile("" : "+v"(v[k]));  // pin it here: the compiler would sink it below the
                                                                 // next row's reads and then wait for those as well
    __builtin_amdgcn_sched_barrier(0);
    if (x > 0) landed_seen = __builtin_amdgcn_readfirstlane(lv);
    if (x + 1 < n_rows) {
      need_landed(x + 2);
      scn = load_scatter(r + 1);
      san = fetch(slot_next, a_row + row_bytes, dn);
      slot_next = slot_next + 1 == G ? 0 : slot_next + 1;
      lv = sync_load(sync_w);
    }
    if (lane == 0) sync_store(my_consumed, x + 1);
    __builtin_amdgcn_sched_barrier(0);
    row_step(v, scc);
    a_row += row_bytes;
    r++;
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int x = 0; x < n_rows; x += 2) {
    half(x, d0, sa0, sc0, d1, sa1, sc1);
    if (x + 1 < n_rows) half(x + 1, d1, sa1, sc1, d0, sa0, sc0);
  }
}

int pick_ybands_v4(int64_t items_per_band, double slots, int taps_h, int64_t H, int64_t oH) {
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
  if (const char *e = getenv("AA_FUSED_YBANDS")) {  // experiment knob
    const int64_t v = atoll(e);
    if (v >= 1 && v <= max_yb) ybands = v;
  }
  return (int)ybands;
}

template <int C, int TW, int MAXC, bool NONNEG>
int launch_k(FusedU8V4Params p, const AAProblem &q, size_t lds) {
  auto kern = fused_u8_nhwc_v4_kernel<C, TW, MAXC, NONNEG>;
  const int waves = p.nstrips + 1;
  static int blocks_per_cu[9] = {0};  // resident workgroups per CU for this instantiation, by waves per workgroup
  if (blocks_per_cu[waves] == 0) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 64 * waves, lds) != hipSuccess || nb <= 0) nb = 1;
    blocks_per_cu[waves] = nb;
  }
  const int taps_h = q.ah.max_taps > 0 ? q.ah.max_taps : q.ah.ksize;
  p.ybands = pick_ybands_v4(q.N, (double)aa_device_cu_count() * blocks_per_cu[waves], taps_h, q.H, q.oH);
  const int64_t grid = q.N * (int64_t)p.ybands;
  if (grid > 0x7FFFFFFF) return 0;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * waves), lds, q.stream, (const uint8_t *)q.in - p.in_mis,
                     (uint8_t *)q.out, (const char *)q.aw.table_dev, (const char *)q.ah.table_dev, p);
  AA_HIP_CHECK_LAUNCH();
  return 1;
}

template <int C, int TW>
int launch_m(int maxc, bool nonneg, const FusedU8V4Params &p, const AAProblem &q, size_t lds) {
  if (maxc <= 2) return nonneg ? launch_k<C, TW, 2, true>(p, q, lds) : launch_k<C, TW, 2, false>(p, q, lds);
  if (maxc <= 3) return nonneg ? launch_k<C, TW, 3, true>(p, q, lds) : launch_k<C, TW, 3, false>(p, q, lds);
  return nonneg ? launch_k<C, TW, 4, true>(p, q, lds) : launch_k<C, TW, 4, false>(p, q, lds);
}

template <int C>
int dispatch_tw(int tw, int maxc, bool nonneg, const FusedU8V4Params &p, const AAProblem &q, size_t lds) {
#ifdef AA_V3_HEADLINE_ONLY  // developer builds: one window width, so the file compiles in seconds
  return tw == 6 ? launch_m<C, 6>(maxc, nonneg, p, q, lds) : 0;
#endif
  if (tw <= 2) return launch_m<C, 2>(maxc, nonneg, p, q, lds);
  if (tw <= 4) return launch_m<C, 4>(maxc, nonneg, p, q, lds);
  if (tw <= 6) return launch_m<C, 6>(maxc, nonneg, p, q, lds);
  if (tw <= 8) return launch_m<C, 8>(maxc, nonneg, p, q, lds);
  if (tw <= 12) return launch_m<C, 12>(maxc, nonneg, p, q, lds);
  return 0;
}

int round_tw_v4(int taps) {
  const int opts[] = {2, 4, 6, 8, 12};
  for (int o : opts)
    if (taps <= o) return o;
  return 0;
}

}  // namespace

int aa_try_fused_u8_nhwc_v4(const AAProblem &q, const char **variant) {
  if (q.dtype != AA_U8 || q.layout != AA_NHWC) return 0;
  if (q.ah.kind != AA_TABLE_PIL || q.aw.kind != AA_TABLE_PIL) return 0;
  const int C = (int)q.C;
  if (C != 3 && C != 4) return 0;
  if (q.ah.scatter_off <= 0 || q.ah.scatter_max <= 0 || q.ah.scatter_max > 4) return 0;
  if (q.H < q.oH) return 0;
  const int taps_w = q.aw.max_taps > 0 ? q.aw.max_taps : q.aw.ksize;
  const int tw = round_tw_v4(taps_w);
  if (tw == 0 || q.W < tw) return 0;
  if ((q.oW * C) % 4 != 0 || (C == 3 && q.oW % 4 != 0)) return 0;
  if ((uint64_t)q.H * q.W * C > 0xFFFFFFF0ull || (uint64_t)q.oH * q.oW * C > 0xFFFFFFF0ull) return 0;
  if (((uintptr_t)q.out & 3) != 0) return 0;

  FusedU8V4Params p;
  p.H = (int)q.H; p.W = (int)q.W; p.oH = (int)q.oH; p.oW = (int)q.oW;
  p.ksize_w = q.aw.ksize; p.ksize_h = q.ah.ksize;
  p.img_in_bytes = (unsigned long long)q.H * q.W * C;
  p.img_out_bytes = (unsigned long long)q.oH * q.oW * C;
  p.in_mis = (int)((uintptr_t)q.in & 15);
  p.total_in_bytes = p.img_in_bytes * (unsigned long long)q.N + (unsigned long long)p.in_mis;
  p.total_out_bytes = p.img_out_bytes * (unsigned long long)q.N;
  p.sc_off = q.ah.scatter_off;
  p.nstrips = (int)((q.oW + 63) / 64);
  if (p.nstrips > 7) return 0;  // strips + the producer must fit one 8-wave workgroup (wider outputs: v3)
  p.strip_w = (int)(((q.oW + p.nstrips - 1) / p.nstrips + 3) & ~3);
  p.nstrips = (int)((q.oW + p.strip_w - 1) / p.strip_w);
  // a ring slot holds one whole input row from the 16-byte boundary below its first byte; the last window's aligned
  // reads may run up to 4 + 3 bytes past the row
  const int64_t row_bytes = q.W * C;
  p.nseg = (int)((row_bytes + 15 + 8 + 15) / 16);
  p.ndma = (p.nseg + 63) / 64;
  if (p.ndma > 8) return 0;  // vmcnt is 6 bits: ndma * (G-2) outstanding DMAs must stay countable
  p.pitch = p.ndma * 1024 > p.nseg * 16 ? p.nseg * 16 : p.ndma * 1024;
  const size_t lds = (size_t)kSyncBytes + (size_t)kG * p.pitch;
  if (lds > 64 * 1024) return 0;
  p.ybands = 1;

  const bool nonneg = q.aw.filter != AA_FILTER_CUBIC && q.ah.filter != AA_FILTER_CUBIC;
  const int rc = (C == 3) ? dispatch_tw<3>(tw, q.ah.scatter_max, nonneg, p, q, lds) : dispatch_tw<4>(tw, q.ah.scatter_max, nonneg, p, q, lds);
  if (rc == 1) *variant = "fused_u8_nhwc_pil_v4";
  return rc;
}
