// proto_mfma.hip — stand-alone driver of the MFMA uint8 kernel (aa_fused_u8_mfma_impl.h): builds Pillow tables on the host, the
// plan on the device, runs the kernel, checks a few images bit for bit against a scalar two-pass Pillow restatement, and times it.
// Developer tool (the library's own tests go through the C-ABI); usage: proto_mfma [N=1024] [H W oH oW] [check_images=3]
// Build: hipcc --offload-arch=gfx950 -O3 -I../../interpolate_antialiasing_amd/csrc -o proto_mfma proto_mfma.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "aa_fused_u8_mfma_impl.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct PilTable { int in, out, ksize; std::vector<int> xmin, xsize, w; };
static PilTable pil_table(int in, int out, int cubic) {
  PilTable t; t.in = in; t.out = out;
  const double scale = (double)in / out, fs = scale < 1.0 ? 1.0 : scale, support = (cubic ? 2.0 : 1.0) * fs;
  t.ksize = (int)ceil(support) * 2 + 1;
  t.xmin.resize(out); t.xsize.resize(out); t.w.assign((size_t)out * t.ksize, 0);
  for (int i = 0; i < out; i++) {
    const double center = (i + 0.5) * scale, ss = 1.0 / fs;
    int xmin = (int)(center - support + 0.5); if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5); if (xmax > in) xmax = in;
    xmax -= xmin;
    std::vector<double> k(xmax);
    double ww = 0;
    for (int x = 0; x < xmax; x++) {
      double a = fabs((x + xmin - center + 0.5) * ss);
      if (cubic) { const double A = -0.5; k[x] = a < 1.0 ? ((A + 2.0) * a - (A + 3.0)) * a * a + 1 : (a < 2.0 ? (((a - 5) * a + 8) * a - 4) * A : 0.0); }
      else k[x] = a < 1.0 ? 1.0 - a : 0.0;
      ww += k[x];
    }
    for (int x = 0; x < xmax; x++) { double v = ww != 0 ? k[x] / ww : k[x]; t.w[(size_t)i * t.ksize + x] = v < 0 ? (int)(-0.5 + v * (1 << 22)) : (int)(0.5 + v * (1 << 22)); }
    t.xmin[i] = xmin; t.xsize[i] = xmax;
  }
  return t;
}
static int clip8(long long v) { v >>= 22; return v < 0 ? 0 : (v > 255 ? 255 : (int)v); }

struct DevAxis { int32_t *xmin, *xsize, *w; };
static AAPilAxis upload(const PilTable &t, DevAxis &d) {
  CK(hipMalloc(&d.xmin, t.out * 4)); CK(hipMalloc(&d.xsize, t.out * 4)); CK(hipMalloc(&d.w, t.w.size() * 4));
  CK(hipMemcpy(d.xmin, t.xmin.data(), t.out * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d.xsize, t.xsize.data(), t.out * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d.w, t.w.data(), t.w.size() * 4, hipMemcpyHostToDevice));
  return AAPilAxis{d.xmin, d.xsize, d.w, t.ksize, t.in, t.out};
}

__global__ void fill_random(uint8_t *p, size_t n, unsigned seed) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    unsigned x = (unsigned)i * 2654435761u + seed;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    p[i] = (uint8_t)(x >> 11);
  }
}

#ifndef PROTO_T
#define PROTO_T 4
#endif
#ifndef PROTO_R
#define PROTO_R 6
#endif

int main(int argc, char **argv) {
  const long long N = argc > 1 ? atoll(argv[1]) : 1024;
  const int H = argc > 5 ? atoi(argv[2]) : 438, W = argc > 5 ? atoi(argv[3]) : 906;
  const int oH = argc > 5 ? atoi(argv[4]) : 196, oW = argc > 5 ? atoi(argv[5]) : 320;
  const int ncheck = argc > 6 ? atoi(argv[6]) : 3;
  const int cubic = argc > 7 ? atoi(argv[7]) : 0;
  const int C = 3;
  PilTable tw = pil_table(W, oW, cubic), th = pil_table(H, oH, cubic);
  // strip span: window starts of 64 consecutive outputs + taps
  int span = 0, taps = 0;
  const int strip_px = 16 * PROTO_T;
  for (int i = 0; i < oW; i++) { const int j = i + strip_px - 1 < oW ? i + strip_px - 1 : oW - 1; span = std::max(span, tw.xmin[j] - tw.xmin[i] + 1); taps = std::max(taps, tw.xsize[i]); }
  const AAPlanGeom g = aa_plan_geometry(C, H, W, oH, oW, span - 1 + taps, 4 * PROTO_T);
  printf("geometry: C %d npx %d tiles %d strips %d nch %d (%d B per staged row) njt %d nsb %d plan %zu B\n", g.C, g.npx, g.ntiles, g.nstrips, g.nch, g.nch * 16, g.njt,
         g.nsb, g.total);
  DevAxis dw, dh;
  const AAPilAxis axw = upload(tw, dw), axh = upload(th, dh);
  char *plan;
  CK(hipMalloc(&plan, g.total));
  CK(hipMemset(plan, 0, g.total));
  aa_plan_header hd;
  memset(&hd, 0, sizeof(hd));
  hd.magic = AA_PLAN_MAGIC; hd.C = C; hd.npx = g.npx; hd.ntiles = g.ntiles; hd.tiles_per_strip = g.tps; hd.nstrips = g.nstrips; hd.nch = g.nch; hd.njt = g.njt; hd.nsb = g.nsb;
  hd.H = H; hd.W = W; hd.oH = oH; hd.oW = oW;
  hd.off_strip = (int)g.off_strip; hd.off_tile = (int)g.off_tile; hd.off_bh = (int)g.off_bh; hd.off_ch = (int)g.off_ch; hd.off_jt = (int)g.off_jt; hd.off_wv = (int)g.off_wv;
  hd.off_cv = (int)g.off_cv; hd.fits = 1; hd.total_bytes = (int)g.total;
  CK(hipMemcpy(plan, &hd, sizeof(hd), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(aa_plan_build_h, dim3(g.ntiles), dim3(64), 0, 0, plan, axw);
  hipLaunchKernelGGL(aa_plan_build_v, dim3(g.njt * 2), dim3(64), 0, 0, plan, axh);
  hipLaunchKernelGGL(aa_plan_finish, dim3(1), dim3(1), 0, 0, plan);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(&hd, plan, sizeof(hd), hipMemcpyDeviceToHost));
  printf("plan: fits %d max_jt_per_blk %d\n", hd.fits, hd.max_jt_per_blk);
  if (!hd.fits) { printf("shape does not fit the MFMA kernel\n"); return 2; }

  const size_t in_bytes = (size_t)N * H * W * C, out_bytes = (size_t)N * oH * oW * C;
  uint8_t *in, *out;
  CK(hipMalloc(&in, in_bytes + 256)); CK(hipMalloc(&out, out_bytes + 256));
  hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, in, in_bytes, 12345u);
  CK(hipMemset(out, 0xCD, out_bytes));

  FusedU8MfmaParams p;
  memset(&p, 0, sizeof(p));
  p.in = in; p.out = out; p.plan = plan; p.H = H; p.W = W; p.oH = oH; p.oW = oW;
  p.nstrips = g.nstrips; p.tps = g.tps; p.nch = g.nch; p.njt = g.njt; p.nsb = g.nsb; p.ntiles = g.ntiles;
  p.off_strip = hd.off_strip; p.off_tile = hd.off_tile; p.off_bh = hd.off_bh; p.off_ch = hd.off_ch; p.off_jt = hd.off_jt; p.off_wv = hd.off_wv; p.off_cv = hd.off_cv;
  p.plan_bytes = (unsigned)g.total;
  p.in_mis = 0;
  p.img_in_bytes = (unsigned long long)H * W * C; p.img_out_bytes = (unsigned long long)oH * oW * C;
  p.total_in_bytes = in_bytes + 3;  // (unaligned dwords: the one holding the last byte may end 3 bytes later.  PROTOTYPE ONLY: the buffer below is
                                  //  allocated 256 bytes longer; a shipped kernel must not read past a tensor - see patch_last in aa_fused_float.hip)
  p.total_out_bytes = out_bytes; p.n_images = N;
  p.x4 = (((uintptr_t)out & 15) == 0 && (oW * C) % 16 == 0 && (oH * oW * C) % 16 == 0) ? 1 : 0;
  constexpr int R = PROTO_R;
  const size_t lds = aa_mfma_lds_bytes(R, g.nch, oH, g.njt, 4 * PROTO_T * 12);
  auto kern = p.x4 ? aa_mfma::fused_u8_nhwc_mfma_kernel<3, 4, R, true, PROTO_T> : aa_mfma::fused_u8_nhwc_mfma_kernel<3, 4, R, false, PROTO_T>;
  CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int nb = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 256, lds));
  const long long groups8 = (N + 7) / 8 * 8;
  const unsigned grid = (unsigned)(groups8 * g.nstrips);
  printf("launch: grid %u x 256 threads, LDS %zu B per workgroup, %d resident workgroups per CU, ring R = %d, T = %d, x4 stores %d\n", grid, lds, nb, R, PROTO_T, p.x4);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, p);
  CK(hipDeviceSynchronize());

  if (ncheck > 0 && AA_MFMA_ABL == 0) {
    // images 0, N/2, N-1 (as many as asked)
    std::vector<long long> which;
    which.push_back(0);
    if (ncheck > 1 && N > 1) which.push_back(N - 1);
    if (ncheck > 2 && N > 2) which.push_back(N / 2);
    long long bad = 0, total = 0;
    std::vector<uint8_t> hi((size_t)H * W * C), ho((size_t)oH * oW * C), tmp((size_t)H * oW * C), ref((size_t)oH * oW * C);
    for (long long n : which) {
      CK(hipMemcpy(hi.data(), in + (size_t)n * hi.size(), hi.size(), hipMemcpyDeviceToHost));
      CK(hipMemcpy(ho.data(), out + (size_t)n * ho.size(), ho.size(), hipMemcpyDeviceToHost));
      for (int y = 0; y < H; y++)
        for (int x = 0; x < oW; x++)
          for (int c = 0; c < C; c++) {
            long long acc = 1 << 21;
            for (int k = 0; k < tw.xsize[x]; k++) acc += (long long)hi[((size_t)y * W + tw.xmin[x] + k) * C + c] * tw.w[(size_t)x * tw.ksize + k];
            tmp[((size_t)y * oW + x) * C + c] = (uint8_t)clip8(acc);
          }
      for (int y = 0; y < oH; y++)
        for (int x = 0; x < oW * C; x++) {
          long long acc = 1 << 21;
          for (int k = 0; k < th.xsize[y]; k++) acc += (long long)tmp[(size_t)(th.xmin[y] + k) * oW * C + x] * th.w[(size_t)y * th.ksize + k];
          ref[(size_t)y * oW * C + x] = (uint8_t)clip8(acc);
        }
      for (size_t i = 0; i < ref.size(); i++) {
        total++;
        if (ref[i] != ho[i]) {
          if (bad < 8) printf("  image %lld row %zu byte %zu: want %d got %d\n", n, i / ((size_t)oW * C), i % ((size_t)oW * C), ref[i], ho[i]);
          bad++;
        }
      }
    }
    printf("check vs scalar Pillow restatement: %lld of %lld bytes differ -> %s\n", bad, total, bad ? "WRONG" : "bit-exact");
  }

  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; rep++) {
    const int iters = 20;
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; i++) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, p);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= iters;
    printf("ABL %d SKEL %d R %d: %.4f ms per %lld images = %.2f TB/s algorithmic (in + out) = %.3f of 8 TB/s\n", AA_MFMA_ABL, AA_MFMA_SKEL, R, ms, N, (in_bytes + out_bytes) / (ms * 1e9),
           (in_bytes + out_bytes) / (ms * 1e9) / 8.0);
  }
  return 0;
}
