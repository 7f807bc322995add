#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned *out, int a, int b) {
  unsigned d0 = 0xAAAAAAAAu, d1 = 0xAAAAAAAAu, d2 = 0xAAAAAAAAu;
  asm volatile("v_ashr_pk_u8_i32 %0, %1, %2, 22" : "+v"(d0) : "v"(a), "v"(b));
  asm volatile("v_ashr_pk_u8_i32 %0, %1, %2, 22 op_sel:[0,0,0,1]" : "+v"(d1) : "v"(a), "v"(b));
  unsigned short r = __builtin_amdgcn_ashr_pk_u8_i32(a, b, 22);
  d2 = r;
  out[0] = d0; out[1] = d1; out[2] = d2;
}
int main() {
  unsigned *d; hipMalloc(&d, 64);
  int a = (0x12 << 22) + 5, b = (0x300 << 22);  // a -> 0x12, b -> saturates 0xff
  hipLaunchKernelGGL(k, 1, 1, 0, 0, d, a, b);
  unsigned h[3]; hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
  printf("plain: %08x  opsel_hi: %08x  builtin: %08x\n", h[0], h[1], h[2]);
  int c = -(5 << 22);
  hipLaunchKernelGGL(k, 1, 1, 0, 0, d, c, a);
  hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
  printf("neg:   %08x  opsel_hi: %08x  builtin: %08x\n", h[0], h[1], h[2]);
  return 0;
}
