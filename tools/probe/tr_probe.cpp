#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(short* out, const int* addr_elems){
  __shared__ __attribute__((aligned(16))) short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
  __syncthreads();
  const int lane = threadIdx.x;
  __attribute__((address_space(3))) v4s* p = (__attribute__((address_space(3))) v4s*)(lds + addr_elems[lane]);
  v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
  for (int j = 0; j < 4; ++j) out[lane*4+j] = r[j];
}
int main(){
  int h_addr[64]; short h_out[256]; int* d_addr; short* d_out;
  hipMalloc(&d_addr, 256); hipMalloc(&d_out, 512);
  for (int pat = 0; pat < 3; ++pat) {
    for (int l = 0; l < 64; ++l) {
      int g = l >> 4, i = l & 15;
      if (pat == 0) h_addr[l] = 4 * l;                               // contiguous segments
      if (pat == 1) h_addr[l] = g * 1024 + (i >> 2) * 100 + (i & 3) * 4;   // 4 rows x 16 cols, row stride 100
      if (pat == 2) h_addr[l] = g * 1024 + i * 100;                  // each lane its own row (stride 100), 4 cols
    }
    hipMemcpy(d_addr, h_addr, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_out, d_addr);
    hipMemcpy(h_out, d_out, 512, hipMemcpyDeviceToHost);
    printf("pattern %d\n", pat);
    for (int l = 0; l < 64; ++l) { printf("lane %2d addr %4d -> %4d %4d %4d %4d\n", l, h_addr[l], h_out[l*4], h_out[l*4+1], h_out[l*4+2], h_out[l*4+3]); }
  }
  return 0;
}
