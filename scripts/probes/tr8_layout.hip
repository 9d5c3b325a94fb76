#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v2i __attribute__((ext_vector_type(2)));
__global__ void k(unsigned char* out, int stride) {
  __shared__ unsigned char lds[64 * 64];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned char)(i & 255);
  __syncthreads();
  // every lane gets a distinct base: row = lane, so that the pattern shows which bytes of which lanes' addresses are read
  typedef v2i __attribute__((address_space(3))) * lp;
  v2i v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lp)(lds + threadIdx.x * stride));
  ((v2i*)out)[threadIdx.x] = v;
}
int main() {
  unsigned char* d; hipMalloc(&d, 512);
  for (int stride : {8, 16, 64}) {
    hipLaunchKernelGGL(k, 1, 64, 0, 0, d, stride);
    unsigned char h[512]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    printf("stride %d (lds[i] = i & 255; lane l reads at l * stride):\n", stride);
    for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j = 0; j < 8; ++j) printf(" %3d", h[l * 8 + j]); printf("\n"); }
  }
  return 0;
}
