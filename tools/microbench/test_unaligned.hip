// Do byte-unaligned buffer_load_dwordx4 / dwordx2 return the right bytes on gfx950 (ROCm 7.2), and what do they cost?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__global__ void check(const unsigned char *in, unsigned *out, unsigned nbytes) {
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, nbytes, 0x00020000);
  unsigned off = threadIdx.x * 17 + 1;  // every alignment
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
  u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off + 16, 64, 0);  // + soffset 64
  out[threadIdx.x * 6 + 0] = v.x; out[threadIdx.x * 6 + 1] = v.y; out[threadIdx.x * 6 + 2] = v.z; out[threadIdx.x * 6 + 3] = v.w;
  out[threadIdx.x * 6 + 4] = w.x; out[threadIdx.x * 6 + 5] = w.y;
}

// throughput: each lane reads a 20-byte window at stride 8.5 B (like the H pass), aligned+6 dwords vs unaligned+5 dwords
template <int MODE>
__global__ void __launch_bounds__(320) stream(const unsigned char *in, unsigned *out, unsigned row_bytes, int rows, unsigned total) {
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(in + (size_t)blockIdx.x * rows * row_bytes), 0,
                                                                 total - blockIdx.x * rows * row_bytes, 0x00020000);
  unsigned lane_off = (unsigned)(threadIdx.x * 8.49f);
  unsigned acc = 0;
  for (int r = 0; r < rows; r += 4) {
    unsigned d[4][6];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      unsigned off = lane_off + (r + i) * row_bytes;
      if (MODE == 0) {
        off &= ~3u;
        u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
        u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off + 16, 0, 0);
        d[i][0] = v.x; d[i][1] = v.y; d[i][2] = v.z; d[i][3] = v.w; d[i][4] = w.x; d[i][5] = w.y;
      } else {
        u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
        unsigned w = __builtin_amdgcn_raw_buffer_load_b32(rsrc, off + 16, 0, 0);
        d[i][0] = v.x; d[i][1] = v.y; d[i][2] = v.z; d[i][3] = v.w; d[i][4] = w; d[i][5] = 0;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int k = 0; k < 6; k++) acc += d[i][k];
  }
  if (acc == 0x12345678) out[0] = acc;
}

int main() {
  const unsigned N = 1 << 16;
  unsigned char *h = (unsigned char *)malloc(N);
  for (unsigned i = 0; i < N; i++) h[i] = (unsigned char)(i * 7 + (i >> 8));
  unsigned char *d; unsigned *o;
  hipMalloc(&d, N); hipMalloc(&o, 64 * 6 * 4);
  hipMemcpy(d, h, N, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(check, 1, 64, 0, 0, d, o, N);
  unsigned ho[64 * 6];
  hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int t = 0; t < 64; t++) {
    unsigned off = t * 17 + 1;
    for (int k = 0; k < 6; k++) {
      unsigned base = off + 4 * k + (k >= 4 ? 64 : 0);
      unsigned exp = h[base] | (h[base + 1] << 8) | (h[base + 2] << 16) | ((unsigned)h[base + 3] << 24);
      if (exp != ho[t * 6 + k]) { if (bad < 5) printf("lane %d dword %d off %u: got %08x exp %08x\n", t, k, base, ho[t*6+k], exp); bad++; }
    }
  }
  printf("unaligned buffer loads: %s (%d mismatches)\n", bad ? "WRONG" : "correct", bad);

  // throughput
  const unsigned row_bytes = 2718; const int rows = 219; const int blocks = 2048;
  size_t total = (size_t)blocks * rows * row_bytes;  // 1.2 GB
  unsigned char *big; hipMalloc(&big, total + 64); hipMemset(big, 1, total + 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; mode++) {
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(stream<0>, blocks, 320, 0, 0, big, o, row_bytes, rows, (unsigned)total);
      else hipLaunchKernelGGL(stream<1>, blocks, 320, 0, 0, big, o, row_bytes, rows, (unsigned)total);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("mode %d (%s): %.3f ms  %.1f GB/s\n", mode, mode ? "unaligned 5 dwords" : "aligned 6 dwords", ms, total / ms / 1e6);
    }
  }
  return 0;
}
