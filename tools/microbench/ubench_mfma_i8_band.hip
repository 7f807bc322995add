// ubench_mfma_i8_band.hip — can the matrix pipe take the uint8 (Pillow, 22-bit fixed point) horizontal pass off the vector ALUs?
//
// The horizontal pass is out[row][e] = clip8((sum_k in[row][k] * Wband[k][e] + 2^21) >> 22) with a banded Wband.  As an
// i8 MFMA (v_mfma_i32_16x16x64_i8): M = 16 input rows, K = 64 contiguous interleaved input bytes of each row (one
// ds_read_b128 per lane from an LDS image laid out [16-byte chunk][row]), N = 12 output elements (4 pixels x 3 channels);
// Pillow's weights split into three signed byte digits (3 MFMAs), pixels biased by -128 (v_xor 0x80808080) with the
// constant 128 * sum(w) + 2^21 entering through the C operand: pure integer arithmetic, hence bit-exact.
//
// Parts (all run by default):
//   1  operand / result lane maps of v_mfma_i32_16x16x64_i8, checked with random data against the host;
//   2  LDS-DMA (buffer_load_dwordx4 ... lds) source alignment: which byte offsets of the global source are served correctly;
//   3  the horizontal-pass tile pipeline on LDS-resident rows: results vs the scalar formula, cycles per tile (16 rows x 12
//      output elements) at 1-4 waves per SIMD;
//   4  staging-pattern streaming rates on the headline geometry (1024 images of 438 rows x 2718 bytes), DMA only.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_mfma_i8_band ubench_mfma_i8_band.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// ------------------------------------------------------------------------------------------------------------- part 1
__global__ void k_layout(const v4i *a, const v4i *b, v4i *d) {
  v4i c = {0, 0, 0, 0};
  d[threadIdx.x] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[threadIdx.x], b[threadIdx.x], c, 0, 0, 0);
}

static int part1() {
  // hypothesis: lane l = (g = l >> 4, i = l & 15) holds A[row i][slot 16 g + j], B[slot 16 g + j][col i] in byte j of its 16;
  // D[row 4 g + r][col i] in register r
  std::vector<int8_t> A(64 * 16), B(64 * 16);
  srand(1);
  for (auto &x : A) x = (int8_t)(rand() & 255);
  for (auto &x : B) x = (int8_t)(rand() & 255);
  v4i *da, *db, *dd;
  CK(hipMalloc(&da, 1024)); CK(hipMalloc(&db, 1024)); CK(hipMalloc(&dd, 1024));
  CK(hipMemcpy(da, A.data(), 1024, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, B.data(), 1024, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_layout, 1, 64, 0, 0, da, db, dd);
  std::vector<int> D(256);
  CK(hipMemcpy(D.data(), dd, 1024, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int l = 0; l < 64; l++)
    for (int r = 0; r < 4; r++) {
      const int row = 4 * (l >> 4) + r, col = l & 15;
      int s = 0;
      for (int g = 0; g < 4; g++)
        for (int j = 0; j < 16; j++) s += (int)A[(g * 16 + row) * 16 + j] * (int)B[(g * 16 + col) * 16 + j];
      if (s != D[l * 4 + r]) bad++;
    }
  printf("part1 mfma_i32_16x16x64_i8 lane maps (A row = lane&15, slots (lane>>4, byte); D row = 4*(lane>>4)+reg, col = lane&15): %s (%d mismatches of 256)\n",
         bad ? "MISMATCH" : "confirmed", bad);
  return bad;
}

// ------------------------------------------------------------------------------------------------------------- part 2
__global__ void k_dma_align(const uint8_t *src, unsigned bytes, int delta, int stride, uint8_t *out) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[1024];
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, bytes, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void *)lds, 16, (unsigned)threadIdx.x * (unsigned)stride + (unsigned)delta, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = 0; i < 16; i++) out[threadIdx.x * 16 + i] = lds[threadIdx.x * 16 + i];
}

static void part2() {
  const unsigned bytes = 1 << 20;
  std::vector<uint8_t> h(bytes);
  for (unsigned i = 0; i < bytes; i++) h[i] = (uint8_t)((i * 2654435761u) >> 13);
  uint8_t *src, *out;
  CK(hipMalloc(&src, bytes)); CK(hipMalloc(&out, 1024));
  CK(hipMemcpy(src, h.data(), bytes, hipMemcpyHostToDevice));
  std::vector<uint8_t> o(1024);
  const int strides[] = {16, 5436};  // contiguous pieces; pieces two 2718-byte rows apart (the [chunk][row] staging)
  for (int stride : strides)
    for (int delta = 0; delta < 16; delta++) {
      CK(hipMemset(out, 0xEE, 1024));
      hipLaunchKernelGGL(k_dma_align, 1, 64, 0, 0, src, bytes, delta, stride, out);
      CK(hipMemcpy(o.data(), out, 1024, hipMemcpyDeviceToHost));
      int bad = 0;
      for (int l = 0; l < 64; l++)
        for (int i = 0; i < 16; i++) bad += o[l * 16 + i] != h[l * stride + delta + i];
      // what did it fetch instead?  try "source rounded down to 4 bytes"
      int bad4 = 0;
      for (int l = 0; l < 64; l++)
        for (int i = 0; i < 16; i++) bad4 += o[l * 16 + i] != h[((l * stride + delta) & ~3) + i];
      printf("part2 LDS-DMA dwordx4, lane stride %4d, source byte offset %2d: %s (%d wrong bytes; vs source rounded down to 4: %d wrong)\n", stride, delta,
             bad ? "WRONG" : "exact", bad, bad4);
    }
}

// ------------------------------------------------------------------------------------------------------------- part 3
// Pillow coefficients (precompute_coeffs + normalize_coeffs_8bpc), triangle filter
struct PilTable { int in, out, ksize; std::vector<int> xmin, xsize, w; };
static PilTable pil_table(int in, int out) {
  PilTable t; t.in = in; t.out = out;
  const double scale = (double)in / out, fs = scale < 1.0 ? 1.0 : scale, support = 1.0 * fs;
  t.ksize = (int)ceil(support) * 2 + 1;
  t.xmin.resize(out); t.xsize.resize(out); t.w.assign((size_t)out * t.ksize, 0);
  for (int i = 0; i < out; i++) {
    const double center = (i + 0.5) * scale, ss = 1.0 / fs;
    int xmin = (int)(center - support + 0.5); if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5); if (xmax > in) xmax = in;
    xmax -= xmin;
    std::vector<double> k(xmax);
    double ww = 0;
    for (int x = 0; x < xmax; x++) { double a = fabs((x + xmin - center + 0.5) * ss); k[x] = a < 1.0 ? 1.0 - a : 0.0; ww += k[x]; }
    for (int x = 0; x < xmax; x++) { double v = ww != 0 ? k[x] / ww : k[x]; t.w[(size_t)i * t.ksize + x] = v < 0 ? (int)(-0.5 + v * (1 << 22)) : (int)(0.5 + v * (1 << 22)); }
    t.xmin[i] = xmin; t.xsize[i] = xmax;
  }
  return t;
}

__device__ inline unsigned pack4_clip8(int a0, int a1, int a2, int a3) {
  unsigned d;
  asm("v_ashr_pk_u8_i32 %0, %1, %2, 22\n\tv_ashr_pk_u8_i32 %0, %3, %4, 22 op_sel:[0,0,0,1]" : "=&v"(d) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
  return d;
}

// One wave: T tiles, NST staged 16-row blocks resident in LDS ([chunk][row] images of NCH chunks), `iters` sweeps over them.
// btab: [tile][plane][lane] 16 bytes; ctab: [tile][lane] int; ch0: [tile] first chunk of the tile's window.
// MODE 0: full pipeline; 1: no MFMA (VALU + LDS only); 2: MFMA + LDS only (no combine / pack); 3: chained accumulators
// (plane 2 -> shift -> C of plane 1 -> shift-add -> C of plane 0); 4: as 0 with the planes combined Horner style
template <int T, int MODE>
__global__ void __launch_bounds__(1024) k_hpass(const uint8_t *rows_img, int nch, int nst, const v4i *btab, const int *ctab, const int *ch0,
                                                int iters, unsigned *out, unsigned long long *cyc) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int lane = threadIdx.x & 63;
  const int stage_bytes = nch * 256;
  for (int i = threadIdx.x; i < nst * stage_bytes / 16; i += blockDim.x) ((uint4 *)lds)[i] = ((const uint4 *)rows_img)[i];
  __syncthreads();
  v4i B[T][3], C0[T];
  unsigned aoff[T];
#pragma unroll
  for (int t = 0; t < T; t++) {
#pragma unroll
    for (int p = 0; p < 3; p++) B[t][p] = btab[(t * 3 + p) * 64 + lane];
    const int c = ctab[t * 64 + lane];
    C0[t] = v4i{c, c, c, c};
    aoff[t] = (unsigned)(ch0[t] * 256 + lane * 16);
  }
  unsigned chk = 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int it = 0; it < iters; it++) {
#pragma unroll 1
    for (int s = 0; s < nst; s++) {
      const unsigned sb = (unsigned)(s * stage_bytes);
      unsigned hres[T];
#pragma unroll
      for (int t = 0; t < T; t++) {
        v4i a = *(const v4i *)(lds + sb + aoff[t]);
        a ^= (int)0x80808080;
        const v4i z = {0, 0, 0, 0};
        v4i val;
        if (MODE == 3) {
          v4i d2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, B[t][2], z, 0, 0, 0);
          d2 <<= 8;
          v4i d1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, B[t][1], d2, 0, 0, 0);
          d1 = (d1 << 8) + C0[t];
          val = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, B[t][0], d1, 0, 0, 0);
        } else if (MODE == 1) {
          const v4i d0 = a + C0[t], d1 = a ^ B[t][1], d2 = a + B[t][2];
          val = d0 + (d1 << 8) + (d2 << 16);
        } else {
          const v4i d0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, B[t][0], C0[t], 0, 0, 0);
          const v4i d1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, B[t][1], z, 0, 0, 0);
          const v4i d2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, B[t][2], z, 0, 0, 0);
          if (MODE == 2) val = d0 ^ d1 ^ d2;
          else if (MODE == 4) val = (((d2 << 8) + d1) << 8) + d0;  // two v_lshl_add_u32 per register
          else val = d0 + (d1 << 8) + (d2 << 16);
        }
        hres[t] = MODE == 2 ? (unsigned)(val.x ^ val.y ^ val.z ^ val.w) : pack4_clip8(val.x, val.y, val.z, val.w);
      }
      if (it == 0 && out && blockIdx.x == 0 && threadIdx.x < 64) {
#pragma unroll
        for (int t = 0; t < T; t++) out[(s * T + t) * 64 + lane] = hres[t];
      }
#pragma unroll
      for (int t = 0; t < T; t++) chk ^= hres[t];
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (chk == 0x12345678u && out) out[0] = chk;
  if (lane == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

static void split_digits(int w, int8_t d[3]) {
  int d0 = ((w + 128) & 255) - 128;
  int w1 = (w - d0) >> 8;
  int d1 = ((w1 + 128) & 255) - 128;
  int d2 = (w1 - d1) >> 8;
  if (d2 < -128 || d2 > 127 || d0 + 256 * d1 + 65536 * d2 != w) { printf("digit split failed for %d\n", w); exit(1); }
  d[0] = (int8_t)d0; d[1] = (int8_t)d1; d[2] = (int8_t)d2;
}

template <int T, int MODE>
static double run_hpass(int wps, int nch, int nst, const uint8_t *d_rows, const v4i *d_b, const int *d_c, const int *d_ch0, unsigned *d_out, int iters,
                        float *ms_out) {
  const int waves = 4 * wps;  // one workgroup per CU: 120 KB of LDS
  const int blocks = 256;
  unsigned long long *d_cyc;
  CK(hipMalloc(&d_cyc, (size_t)blocks * waves * 8));
  const size_t lds = 120 * 1024;
  CK(hipFuncSetAttribute((const void *)k_hpass<T, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_hpass<T, MODE>), dim3(blocks), dim3(64 * waves), lds, 0, d_rows, nch, nst, d_b, d_c, d_ch0, iters, rep == 0 ? d_out : nullptr, d_cyc);
    hipEventRecord(e1); CK(hipEventSynchronize(e1));
    hipEventElapsedTime(&ms, e0, e1);
  }
  std::vector<unsigned long long> c((size_t)blocks * waves);
  CK(hipMemcpy(c.data(), d_cyc, c.size() * 8, hipMemcpyDeviceToHost));
  double s = 0; for (auto v : c) s += (double)v;
  hipFree(d_cyc);
  *ms_out = ms;
  return s / c.size() / ((double)iters * nst * T);  // counter ticks per tile per wave
}

static void part3() {
  const int W = 906, oW = 320, C = 3, T = 4, NST = 4;
  PilTable tb = pil_table(W, oW);
  const int row_bytes = W * C;
  // a strip in the middle of the row: tiles tile0 .. tile0+T-1 (4 pixels each)
  const int tile0 = 37;
  const int seg_first = tb.xmin[tile0 * 4] * C;
  const int seg_origin = seg_first & ~3;  // LDS byte 0 of a staged row = this byte of the row (DMA source rounded down to 4)
  int need_end = 0;
  for (int t = 0; t < T; t++) { const int px = (tile0 + t) * 4 + 3; need_end = (tb.xmin[px] + tb.xsize[px]) * C; }
  const int nch = ((need_end - seg_origin + 15) / 16 + 3) & ~3;
  printf("part3 strip: tiles %d..%d, segment bytes [%d,%d) -> %d chunks of 16 B\n", tile0, tile0 + T - 1, seg_origin, need_end, nch);
  // image rows: random bytes; NST blocks of 16 rows
  std::vector<uint8_t> img((size_t)NST * 16 * row_bytes);
  srand(7);
  for (auto &x : img) x = (uint8_t)(rand() >> 7);
  for (int i = 0; i < row_bytes; i++) { img[i] = 255; img[row_bytes + i] = 0; }  // extreme rows
  std::vector<uint8_t> ldsimg((size_t)NST * nch * 256);
  for (int s = 0; s < NST; s++)
    for (int ch = 0; ch < nch; ch++)
      for (int m = 0; m < 16; m++)
        for (int j = 0; j < 16; j++) {
          const int rb = seg_origin + ch * 16 + j;
          ldsimg[(size_t)s * nch * 256 + (ch * 16 + m) * 16 + j] = rb < row_bytes ? img[(size_t)(s * 16 + m) * row_bytes + rb] : 0;
        }
  std::vector<int8_t> btab((size_t)T * 3 * 64 * 16, 0);
  std::vector<int> ctab(T * 64, 0), ch0(T);
  for (int t = 0; t < T; t++) {
    const int px0 = (tile0 + t) * 4;
    const int f = tb.xmin[px0] * C - seg_origin;
    ch0[t] = f >> 4;
    const int last = (tb.xmin[px0 + 3] + tb.xsize[px0 + 3]) * C - seg_origin;
    if (last > ch0[t] * 16 + 64) { printf("tile %d window does not fit 64 slots (%d)\n", t, last - ch0[t] * 16); exit(1); }
    for (int n = 0; n < 12; n++) {
      const int px = px0 + n / 3, c = n % 3;
      long long sumw = 0;
      for (int k = 0; k < tb.xsize[px]; k++) {
        const int w = tb.w[(size_t)px * tb.ksize + k];
        sumw += w;
        const int rb = (tb.xmin[px] + k) * C + c;           // byte of the row
        const int slot = rb - seg_origin - ch0[t] * 16;      // 0..63
        int8_t d[3]; split_digits(w, d);
        for (int p = 0; p < 3; p++) btab[(((size_t)t * 3 + p) * 64 + (slot >> 4) * 16 + n) * 16 + (slot & 15)] = d[p];
      }
      for (int g = 0; g < 4; g++) ctab[t * 64 + g * 16 + n] = (int)(128 * sumw + (1 << 21));
    }
  }
  uint8_t *d_rows; v4i *d_b; int *d_c, *d_ch0; unsigned *d_out;
  CK(hipMalloc(&d_rows, ldsimg.size())); CK(hipMalloc(&d_b, btab.size())); CK(hipMalloc(&d_c, ctab.size() * 4)); CK(hipMalloc(&d_ch0, T * 4));
  CK(hipMalloc(&d_out, (size_t)NST * T * 64 * 4));
  CK(hipMemcpy(d_rows, ldsimg.data(), ldsimg.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(d_b, btab.data(), btab.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(d_c, ctab.data(), ctab.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_ch0, ch0.data(), T * 4, hipMemcpyHostToDevice));
  const int iters = 2000;
  for (int mode : {0, 3, 4}) {
    float ms;
    CK(hipMemset(d_out, 0, (size_t)NST * T * 64 * 4));
    if (mode == 0) run_hpass<T, 0>(1, nch, NST, d_rows, d_b, d_c, d_ch0, d_out, 4, &ms);
    else if (mode == 3) run_hpass<T, 3>(1, nch, NST, d_rows, d_b, d_c, d_ch0, d_out, 4, &ms);
    else run_hpass<T, 4>(1, nch, NST, d_rows, d_b, d_c, d_ch0, d_out, 4, &ms);
    std::vector<unsigned> o((size_t)NST * T * 64);
    CK(hipMemcpy(o.data(), d_out, o.size() * 4, hipMemcpyDeviceToHost));
    long long bad = 0, total = 0;
    for (int s = 0; s < NST; s++)
      for (int t = 0; t < T; t++)
        for (int l = 0; l < 64; l++) {
          const int n = l & 15, g = l >> 4;
          if (n >= 12) continue;
          const int px = (tile0 + t) * 4 + n / 3, c = n % 3;
          for (int r = 0; r < 4; r++) {
            const int row = s * 16 + 4 * g + r;
            long long acc = 1 << 21;
            for (int k = 0; k < tb.xsize[px]; k++) acc += (long long)img[(size_t)row * row_bytes + (tb.xmin[px] + k) * C + c] * tb.w[(size_t)px * tb.ksize + k];
            long long v = acc >> 22; v = v < 0 ? 0 : (v > 255 ? 255 : v);
            const unsigned got = (o[((size_t)s * T + t) * 64 + l] >> (8 * r)) & 255u;
            total++;
            if ((unsigned)v != got) { if (bad < 5) printf("  mismatch stage %d tile %d lane %d row %d: want %lld got %u\n", s, t, l, row, v, got); bad++; }
          }
        }
    printf("part3 mode %d (%s): horizontal-pass results vs the scalar formula: %lld of %lld differ -> %s\n", mode, mode == 0 ? "three independent planes" : (mode == 3 ? "chained C operands" : "Horner combine"),
           bad, total, bad ? "WRONG" : "bit-exact");
  }
  for (int wps = 1; wps <= 4; wps++) {
    float ms0, ms1, ms2, ms3, ms4;
    const double c0 = run_hpass<T, 0>(wps, nch, NST, d_rows, d_b, d_c, d_ch0, d_out, iters, &ms0);
    const double c1 = run_hpass<T, 1>(wps, nch, NST, d_rows, d_b, d_c, d_ch0, d_out, iters, &ms1);
    const double c2 = run_hpass<T, 2>(wps, nch, NST, d_rows, d_b, d_c, d_ch0, d_out, iters, &ms2);
    const double c3 = run_hpass<T, 3>(wps, nch, NST, d_rows, d_b, d_c, d_ch0, d_out, iters, &ms3);
    run_hpass<T, 4>(wps, nch, NST, d_rows, d_b, d_c, d_ch0, d_out, iters, &ms4);
    const double tiles_per_simd = (double)iters * NST * T * wps;
    auto ns = [&](float ms) { return ms * 1e6 / tiles_per_simd; };
    printf("part3 waves/SIMD %d: per tile (16 rows x 12 elements) and SIMD: full %.1f ns (%.0f ticks/wave)  VALU+LDS only %.1f ns  MFMA+LDS only %.1f ns  chained %.1f ns (%.0f ticks/wave)  Horner combine %.1f ns"
           "   [one 64-column strip row = 1 tile: compare 150 issue cycles = 71 ns at 2.1 GHz for the VALU kernel's horizontal part]\n",
           wps, ns(ms0), c0, ns(ms1), ns(ms2), ns(ms3), c3, ns(ms4));
  }
}

// ------------------------------------------------------------------------------------------------------------- part 4
// DMA-only streaming of [n_img][H][row_bytes] uint8.  One wave = one workgroup = (image, strip), all rows.
// pattern 0: the VALU kernel's staging: one 1-KiB contiguous piece per row (source rounded down to 16 B), `depth` rows in flight.
// pattern 1: [chunk][row] staging for the MFMA kernel: a DMA instruction = 4 chunks (64 B) of 16 same-parity rows
//            (source rounded down to 4 B); a block = 16 rows x nch chunks = nch/4 instructions; `depth` blocks in flight.
// pattern 2: 16 chunks (256 B) of 4 same-parity rows per instruction.
__global__ void __launch_bounds__(64) k_stream(const uint8_t *in, unsigned long long total, int H, int row_bytes, int nstrips, int strip_stride, int nch,
                                               int pattern, int depth, long long n_img, unsigned *out) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int lane = threadIdx.x;
  const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
  const int strip = k % nstrips;
  const long long n = (long long)(k / nstrips) * 8 + xcd;
  if (n >= n_img) return;
  const unsigned long long img_off = (unsigned long long)n * H * row_bytes;
  const unsigned long long base_off = img_off & ~15ull;
  unsigned long long rem = total - base_off; if (rem > 0xFFFFFFFCull) rem = 0xFFFFFFFCull;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(in + base_off), 0, (unsigned)rem, 0x00020000);
  const unsigned a0 = (unsigned)(img_off - base_off) + (unsigned)(strip * strip_stride);
  unsigned acc = 0;
  if (pattern == 0) {
    const int slot_bytes = 1024;
    int issued = 0;
    for (; issued < depth && issued < H; issued++)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void *)(lds + (issued % depth) * slot_bytes), 16, lane * 16u, (a0 + issued * row_bytes) & ~15u, 0, 0);
    for (int r = 0; r < H; r++) {
      if (H - 1 - r >= depth - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(7) : "memory");  // depth 8
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      acc ^= *(const unsigned *)(lds + (r % depth) * slot_bytes + lane * 16);
      if (issued < H) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void *)(lds + (issued % depth) * slot_bytes), 16, lane * 16u, (a0 + issued * row_bytes) & ~15u, 0, 0);
        issued++;
      }
    }
  } else {
    const int ipb = nch / 4;  // instructions per block (pattern 1) ; pattern 2: 4 row-groups x ceil(nch/16)
    const int blk_bytes = nch * 256;
    const int nblk = (H + 31) / 32 * 2;  // even block i, odd block i, ...
    unsigned voff;
    if (pattern == 1) voff = (unsigned)((lane & 15) * 2 * row_bytes + (lane >> 4) * 16);
    else voff = (unsigned)((lane & 3) * 2 * row_bytes + (lane >> 2) * 16);
    auto issue = [&](int b) {
      const int par = b & 1, i = b >> 1;
      const unsigned arow = (a0 + (unsigned)(32 * i + par) * (unsigned)row_bytes) & ~3u;
      uint8_t *dst = lds + (b % depth) * blk_bytes;
      if (pattern == 1) {
        for (int q = 0; q < ipb; q++) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void *)(dst + q * 1024), 16, voff, arow + q * 64u, 0, 0);
      } else {
        for (int rg = 0; rg < 4; rg++)
          for (int q = 0; q < (nch + 15) / 16; q++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void *)(dst + (rg * ((nch + 15) / 16) + q) * 1024), 16, voff, arow + (unsigned)(rg * 8 * row_bytes) + q * 256u, 0, 0);
      }
    };
    int issued = 0;
    for (; issued < depth - 1 && issued < nblk; issued++) issue(issued);
    for (int b = 0; b < nblk; b++) {
      if (issued < nblk) { issue(issued); issued++; }
      // wait until block b has landed: everything but the (issued - 1 - b) younger blocks
      const int younger = issued - 1 - b;
      if (younger <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else {
        const int per = pattern == 1 ? ipb : 4 * ((nch + 15) / 16);
        const int cnt = younger * per;
        if (cnt >= 30) asm volatile("s_waitcnt vmcnt(30)" ::: "memory");
        else if (cnt >= 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
        else if (cnt >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (cnt >= 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if (cnt >= 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        else if (cnt >= 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (cnt >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (cnt >= 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else if (cnt >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
      }
      acc ^= *(const unsigned *)(lds + (b % depth) * blk_bytes + lane * 16);
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
}

static void part4() {
  const int H = 438, W = 906, C = 3, row_bytes = W * C;
  const long long n_img = 1024;
  const unsigned long long total = (unsigned long long)n_img * H * row_bytes;
  uint8_t *in; unsigned *out;
  CK(hipMalloc(&in, total + 4096)); CK(hipMalloc(&out, 64));
  CK(hipMemset(in, 0x5A, total + 4096));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  struct Cfg { const char *name; int pattern, nstrips, strip_stride, nch, depth; };
  // strip strides: 64 output columns = 543.6 bytes of input (5 strips per row); 16 columns = 135.9 (20 strips); 32 columns = 271.8 (10 strips)
  const Cfg cfgs[] = {
      {"VALU kernel staging: 5 strips x 1 KiB per row, 8 rows in flight", 0, 5, 544, 64, 8},
      {"[chunk][16 rows]: 20 strips x 12 chunks (3 instr / block), 4 blocks in flight", 1, 20, 136, 12, 4},
      {"[chunk][16 rows]: 20 strips x 12 chunks, 6 blocks in flight", 1, 20, 136, 12, 6},
      {"[chunk][16 rows]: 10 strips x 20 chunks (5 instr / block), 4 blocks in flight", 1, 10, 272, 20, 4},
      {"[chunk][16 rows]: 5 strips x 40 chunks (10 instr / block), 3 blocks in flight", 1, 5, 544, 40, 3},
      {"[16 chunks][4 rows]: 20 strips x 16 chunks (4 instr / block), 4 blocks in flight", 2, 20, 136, 16, 4},
      {"[16 chunks][4 rows]: 10 strips x 32 chunks (8 instr / block), 3 blocks in flight", 2, 10, 272, 32, 3},
  };
  for (const Cfg &c : cfgs) {
    const size_t lds = c.pattern == 0 ? (size_t)c.depth * 1024 : (size_t)c.depth * c.nch * 256;  // (pattern 2: nch is a multiple of 16)
    CK(hipFuncSetAttribute((const void *)k_stream, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const long long groups8 = (n_img + 7) / 8 * 8;
    const unsigned grid = (unsigned)(groups8 * c.nstrips);
    float best = 1e9;
    for (int rep = 0; rep < 4; rep++) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_stream, dim3(grid), dim3(64), lds, 0, in, total, H, row_bytes, c.nstrips, c.strip_stride, c.nch, c.pattern, c.depth, n_img, out);
      hipEventRecord(e1); CK(hipEventSynchronize(e1));
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf("part4 %-86s LDS/wave %6zu B: %.3f ms per 1024 images = %.2f TB/s of input\n", c.name, lds, best, total / (best * 1e9));
  }
}


// ------------------------------------------------------------------------------------------------------------- part 5
// issue interval of the matrix instructions themselves: 8 independent accumulators, operands in registers, inline asm (hipcc's own
// code for an accumulator array in a loop shuffles everything through AGPRs)
template <int KIND>
__global__ void __launch_bounds__(512) k_mfma_rate(int iters, unsigned *out, unsigned long long *cyc) {
  const int lane = threadIdx.x & 63;
  v4i a = {lane, lane * 3, lane * 5, lane * 7}, b = {lane * 11, lane * 13, lane * 17, lane * 19};
  v4i c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int it = 0; it < iters; it++) {
    if (KIND == 0)
      asm volatile(
          "v_mfma_i32_16x16x64_i8 %0, %8, %9, %0\n\tv_mfma_i32_16x16x64_i8 %1, %8, %9, %1\n\tv_mfma_i32_16x16x64_i8 %2, %8, %9, %2\n\t"
          "v_mfma_i32_16x16x64_i8 %3, %8, %9, %3\n\tv_mfma_i32_16x16x64_i8 %4, %8, %9, %4\n\tv_mfma_i32_16x16x64_i8 %5, %8, %9, %5\n\t"
          "v_mfma_i32_16x16x64_i8 %6, %8, %9, %6\n\tv_mfma_i32_16x16x64_i8 %7, %8, %9, %7"
          : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
          : "v"(a), "v"(b));
    else
      asm volatile(
          "v_mfma_f32_16x16x32_bf16 %0, %8, %9, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %8, %9, %1\n\tv_mfma_f32_16x16x32_bf16 %2, %8, %9, %2\n\t"
          "v_mfma_f32_16x16x32_bf16 %3, %8, %9, %3\n\tv_mfma_f32_16x16x32_bf16 %4, %8, %9, %4\n\tv_mfma_f32_16x16x32_bf16 %5, %8, %9, %5\n\t"
          "v_mfma_f32_16x16x32_bf16 %6, %8, %9, %6\n\tv_mfma_f32_16x16x32_bf16 %7, %8, %9, %7"
          : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
          : "v"(a), "v"(b));
  }
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  const v4i s = c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7;
  if ((unsigned)(s.x ^ s.y ^ s.z ^ s.w) == 0x12345678u) out[0] = 1;
  if (lane == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

static void part5() {
  unsigned *out; unsigned long long *cyc;
  CK(hipMalloc(&out, 64)); CK(hipMalloc(&cyc, 256 * 16 * 8));
  const int iters = 20000;
  for (int kind = 0; kind < 2; kind++)
    for (int wps = 1; wps <= 2; wps++) {
      const int waves = 4 * wps;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      float ms = 0;
      for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        if (kind == 0) hipLaunchKernelGGL(k_mfma_rate<0>, dim3(256), dim3(64 * waves), 0, 0, iters, out, cyc);
        else hipLaunchKernelGGL(k_mfma_rate<1>, dim3(256), dim3(64 * waves), 0, 0, iters, out, cyc);
        hipEventRecord(e1); CK(hipEventSynchronize(e1));
        hipEventElapsedTime(&ms, e0, e1);
      }
      std::vector<unsigned long long> c((size_t)256 * waves);
      CK(hipMemcpy(c.data(), cyc, c.size() * 8, hipMemcpyDeviceToHost));
      double s = 0; for (auto v : c) s += (double)v;
      const double ticks = s / c.size() / ((double)iters * 8);
      const double ns = ms * 1e6 / ((double)iters * 8 * wps);
      printf("part5 %-26s waves/SIMD %d: %.1f ticks per instruction and wave = %.1f per SIMD; %.2f ns per instruction and SIMD (clock %.2f GHz)\n",
             kind == 0 ? "v_mfma_i32_16x16x64_i8" : "v_mfma_f32_16x16x32_bf16", wps, ticks, ticks / wps, ns, ticks / wps / ns);
    }
}

int main(int argc, char **argv) {
  const int only = argc > 1 ? atoi(argv[1]) : 0;
  if (only == 0 || only == 1) part1();
  if (only == 0 || only == 2) part2();
  if (only == 0 || only == 3) part3();
  if (only == 0 || only == 4) part4();
  if (only == 0 || only == 5) part5();
  return 0;
}
