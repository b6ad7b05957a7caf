// fetch_cal.hip -- calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths the csnet kernels use
// (MI355X_MICROARCH.md, HBM section: FETCH_SIZE counts a 16 B/lane stream at half its bytes; other widths and WRITE_SIZE
// are to be calibrated on a known byte count in the kernel's own access pattern).  Every kernel streams N bytes ONCE
// (N = 1 GiB > the 256 MiB Infinity Cache) with buffer loads / stores of one width; run under
//   rocprofv3 --pmc FETCH_SIZE ...   and   rocprofv3 --pmc WRITE_SIZE ...
// and divide the known bytes by the counter (tools/pmc_hbm.py does).  build: hipcc --offload-arch=gfx950 -O3 -o fetch_cal fetch_cal.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __amdgpu_buffer_rsrc_t buf;
typedef unsigned u2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
#define CHUNK (1u << 30)
// W = bytes per lane (4, 8, 16); RD: read and reduce, else write
template <int W, bool RD>
__global__ __launch_bounds__(256) void stream_kernel(char* p, unsigned* out, unsigned n_elems) {
  buf b = __builtin_amdgcn_make_buffer_rsrc(p, 0, CHUNK, 0x00020000);
  unsigned acc = 0;
  for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n_elems; i += gridDim.x * 256) {
    const unsigned off = i * W;
    if (RD) {
      if (W == 4) acc += __builtin_amdgcn_raw_buffer_load_b32(b, off, 0, 0);
      if (W == 8) { u2 v = __builtin_amdgcn_raw_buffer_load_b64(b, off, 0, 0); acc += v.x + v.y; }
      if (W == 16) { u4 v = __builtin_amdgcn_raw_buffer_load_b128(b, off, 0, 0); acc += v.x + v.y + v.z + v.w; }
    } else {
      if (W == 4) __builtin_amdgcn_raw_buffer_store_b32(i, b, off, 0, 0);
      if (W == 8) { u2 v; v.x = i; v.y = i; __builtin_amdgcn_raw_buffer_store_b64(v, b, off, 0, 0); }
      if (W == 16) { u4 v; v.x = i; v.y = i; v.z = i; v.w = i; __builtin_amdgcn_raw_buffer_store_b128(v, b, off, 0, 0); }
    }
  }
  if (RD && acc == 0x12345678u) out[0] = acc;
}
// the half-used-line pattern of pw4 / c3q quads: lane l reads 8 B at 16 l (every other 8-byte word), a second pass the rest
__global__ __launch_bounds__(256) void stride2_kernel(char* p, unsigned* out, unsigned n_pairs) {
  buf b = __builtin_amdgcn_make_buffer_rsrc(p, 0, CHUNK, 0x00020000);
  unsigned acc = 0;
  for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n_pairs; i += gridDim.x * 256) {
    acc += __builtin_amdgcn_raw_buffer_load_b32(b, i * 8, 0, 0);
    acc += __builtin_amdgcn_raw_buffer_load_b32(b, i * 8 + 4, 0, 0);
  }
  if (acc == 0x12345678u) out[0] = acc;
}
int main() {
  char* p; unsigned* o;
  if (hipMalloc(&p, CHUNK) != hipSuccess || hipMalloc(&o, 4) != hipSuccess) return 1;
  (void)hipMemset(p, 1, CHUNK);
  (void)hipDeviceSynchronize();
  hipLaunchKernelGGL((stream_kernel<4, true>), dim3(4096), dim3(256), 0, 0, p, o, CHUNK / 4);
  hipLaunchKernelGGL((stream_kernel<8, true>), dim3(4096), dim3(256), 0, 0, p, o, CHUNK / 8);
  hipLaunchKernelGGL((stream_kernel<16, true>), dim3(4096), dim3(256), 0, 0, p, o, CHUNK / 16);
  hipLaunchKernelGGL(stride2_kernel, dim3(4096), dim3(256), 0, 0, p, o, CHUNK / 8);
  hipLaunchKernelGGL((stream_kernel<4, false>), dim3(4096), dim3(256), 0, 0, p, o, CHUNK / 4);
  hipLaunchKernelGGL((stream_kernel<8, false>), dim3(4096), dim3(256), 0, 0, p, o, CHUNK / 8);
  hipLaunchKernelGGL((stream_kernel<16, false>), dim3(4096), dim3(256), 0, 0, p, o, CHUNK / 16);
  (void)hipDeviceSynchronize();
  printf("fetch_cal: 7 kernels x %u bytes\n", CHUNK);
  return 0;
}
