// ubench_mix.hip — does integer VALU work slow down when OTHER workgroups stream HBM at the same time?
// One kernel, role by workgroup index: role A = dependent-free v_mul_i32_i24 / v_add3 chains (no memory), role B = dwordx4
// streaming reads.  Modes: 0 = A only (B workgroups exit), 1 = B only, 2 = both.  If mode 2 takes ~max(mode 0, mode 1) the
// two activities coexist; if it takes much longer, something chip-wide (clock / power management, fabric) couples them.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_mix ubench_mix.hip ; run: ./ubench_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((address_space(3))) void lds_void;

// skind: 0 = plain dwordx4 loads into registers, 1 = LDS-DMA (buffer_load_dwordx4 ... lds), the staging path of the fused
// kernels.  vkind: 0 = pure VALU, 1 = VALU + one 64-lane ds_read2_b32 per 8 multiplies (the fused kernels' ratio).
__global__ void __launch_bounds__(256) mix(int mode, int skind, int vkind, const uint4 *__restrict__ buf, size_t n16_per_block,
                                           int iters, unsigned *out, unsigned long long *t_role) {
  __shared__ __attribute__((aligned(16))) unsigned lds[4096];  // 16 KiB
  const bool roleB = blockIdx.x & 1;
  if ((mode == 0 && roleB) || (mode == 1 && !roleB)) return;
  const unsigned long long t0 = wall_clock64();
  unsigned acc = 0;
  if (!roleB) {
    int a0 = threadIdx.x, a1 = threadIdx.x * 3, a2 = threadIdx.x * 5, a3 = threadIdx.x * 7, w = (int)blockIdx.x | 1;
#pragma unroll 1
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int k = 0; k < 16; k++) {
        a0 = __mul24(a0, w) + k; a1 = __mul24(a1, w) + k; a2 = __mul24(a2, w) + k; a3 = __mul24(a3, w) + k;
        if (vkind && (k & 1)) {
          const unsigned idx = ((unsigned)threadIdx.x * 3u + (unsigned)(i + k)) & 4094u;
          a0 ^= (int)lds[idx];
          a1 ^= (int)lds[idx + 1];
        }
      }
    }
    acc = (unsigned)(a0 ^ a1 ^ a2 ^ a3);
  } else {
    const uint4 *p = buf + (size_t)(blockIdx.x >> 1) * n16_per_block;
    uint4 s = {0, 0, 0, 0};
    if (skind == 1) {
      // each wave streams its quarter of the chunk with LDS-DMA into its own 4 KiB of LDS, 4 instructions in flight
      const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, (unsigned)(n16_per_block * 16), 0x00020000);
      const unsigned per_wave = (unsigned)(n16_per_block * 16 / 4);
      unsigned off = (unsigned)wv * per_wave;
      for (unsigned done = 0; done < per_wave; done += 4096) {
#pragma unroll
        for (int q = 0; q < 4; q++)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void *)((char *)lds + wv * 4096 + q * 1024), 16, (unsigned)lane * 16u,
                                                   off + done + q * 1024, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      s.x = lds[threadIdx.x];
    } else
    for (size_t i = threadIdx.x; i < n16_per_block; i += 256 * 4) {
      uint4 v0 = p[i], v1 = p[i + 256], v2 = p[i + 512], v3 = p[i + 768];
      s.x ^= v0.x ^ v1.x ^ v2.x ^ v3.x; s.y ^= v0.y ^ v1.y ^ v2.y ^ v3.y;
      s.z ^= v0.z ^ v1.z ^ v2.z ^ v3.z; s.w ^= v0.w ^ v1.w ^ v2.w ^ v3.w;
    }
    acc = s.x ^ s.y ^ s.z ^ s.w;
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
  if (threadIdx.x == 0) t_role[blockIdx.x] = wall_clock64() - t0;
}

int main(int argc, char **argv) {
  const int pairs = argc > 1 ? atoi(argv[1]) : 4096;       // workgroups per role
  const int iters = argc > 2 ? atoi(argv[2]) : 3000;       // VALU work per role-A workgroup
  const int skind = argc > 3 ? atoi(argv[3]) : 0;
  const int vkind = argc > 4 ? atoi(argv[4]) : 0;
  const size_t n16 = 1024 * 256;                           // 4 MiB per role-B workgroup
  const int blocks = pairs * 2;
  uint4 *buf; unsigned *out; unsigned long long *tr;
  hipMalloc(&buf, (size_t)pairs * n16 * 16); hipMemset(buf, 1, (size_t)pairs * n16 * 16);
  hipMalloc(&out, (size_t)blocks * 256 * 4); hipMalloc(&tr, blocks * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<unsigned long long> h(blocks);
  for (int rep = 0; rep < 2; rep++)
    for (int mode = 0; mode < 3; mode++) {
      hipMemset(tr, 0, blocks * 8);
      hipEventRecord(e0);
      hipLaunchKernelGGL(mix, dim3(blocks), dim3(256), 0, 0, mode, skind, vkind, buf, n16, iters, out, tr);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(h.data(), tr, blocks * 8, hipMemcpyDeviceToHost);
      double sa = 0, sb = 0; int na = 0, nb = 0;
      for (int b = 0; b < blocks; b++) { if (!h[b]) continue; if (b & 1) { sb += h[b]; nb++; } else { sa += h[b]; na++; } }
      printf("skind %d vkind %d rep %d mode %d: %.3f ms   roleA(VALU) avg %.1f us over %d wgs   roleB(stream %.1f GB) avg %.1f us over %d wgs  -> %.0f GB/s\n",
             skind, vkind, rep, mode, ms, na ? sa / na / 100.0 : 0.0, na, pairs * n16 * 16 / 1e9, nb ? sb / nb / 100.0 : 0.0, nb,
             mode ? pairs * n16 * 16 / (ms * 1e6) : 0.0);
    }
  return 0;
}
