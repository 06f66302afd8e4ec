// Which XCD does workgroup i of a 1-D grid run on?  (HW_REG_XCC_ID)  Measurement tool, not product.
#include <hip/hip_runtime.h>
__global__ void xcc_map(unsigned *out) {
    if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u;
}
extern "C" int xcc_map_launch(unsigned *out, int blocks, int threads, void *stream) {
    hipLaunchKernelGGL(xcc_map, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, out);
    return (int)hipGetLastError();
}
