// FETCH_SIZE calibration (developer tool, MI355X_MICROARCH.md "HBM": widths other than 16 B/lane are uncalibrated):
// streams a 1 GiB buffer (larger than the 256 MiB Infinity Cache) once with (a) 16 B per lane, (b) 4 B per lane,
// (c) the FAST kernel's pattern -- one wave per 44-byte x 37-row tile of a 1920-byte-pitch image, dword per lane.
// Run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv`; tools/fetch_calib_summary.py prints
// FETCH_SIZE * 1024 / bytes actually requested for each kernel.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void calib_b128(const uint4* p, size_t n, unsigned* out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned acc = 0;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ void calib_b32(const unsigned* p, size_t n, unsigned* out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned acc = 0;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= p[i];
  if (acc == 0x12345678u) out[0] = acc;
}
// image pitch 1920, tiles 40 x 31 pixels read with a 3-pixel halo (44-byte = 11-dword rows, 37 rows), block (tile x, tile y, image)
__global__ __launch_bounds__(64) void calib_tiles(const unsigned char* base, int pitch, int rows, size_t imgBytes, unsigned* out) {
  const unsigned char* img = base + imgBytes * blockIdx.z;
  const int x0 = blockIdx.x * 40, y0 = blockIdx.y * 31, lane = threadIdx.x, c = lane % 11, r0 = lane / 11;
  unsigned acc = 0;
  if (r0 < 5)
    for (int r = r0; r < 37 && y0 + r < rows; r += 5) acc ^= *reinterpret_cast<const unsigned*>(img + (size_t)(y0 + r) * pitch + x0 + 4 * c);
  if (acc == 0x12345678u) out[0] = acc;
}

int main() {
  const size_t bytes = 1ull << 30;
  void* d; unsigned* o;
  hipMalloc(&d, bytes + 4096); hipMalloc(&o, 64);
  hipMemset(d, 1, bytes + 4096);
  hipDeviceSynchronize();
  hipLaunchKernelGGL(calib_b128, dim3(8192), dim3(256), 0, 0, (const uint4*)d, bytes / 16, o);
  hipDeviceSynchronize();
  hipLaunchKernelGGL(calib_b32, dim3(8192), dim3(256), 0, 0, (const unsigned*)d, bytes / 4, o);
  hipDeviceSynchronize();
  const int pitch = 1920, rows = 1080;
  const size_t img = (size_t)pitch * rows;
  const int nimg = (int)(bytes / img);
  hipLaunchKernelGGL(calib_tiles, dim3(47, 34, nimg), dim3(64), 0, 0, (const unsigned char*)d, pitch, rows, img, o);   // 47 x 40 = 1880 (+4 halo) columns, 34 x 31 = 1054 (+6) rows
  hipDeviceSynchronize();
  printf("bytes requested: calib_b128 %zu, calib_b32 %zu, calib_tiles %zu (image area touched: %zu)\n", bytes, bytes,
         (size_t)nimg * 47 * 34 * 37 * 44, (size_t)nimg * 1884 * 1060);
  return 0;
}
