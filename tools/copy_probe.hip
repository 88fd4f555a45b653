// copy_probe: what a streaming read-modify-write of N bytes can reach on this GPU, by access shape.
// Reference point for k_move_t's roofline fraction (it reads and rewrites every live car once).
//   hipcc --offload-arch=gfx950 -O3 -o copy_probe tools/copy_probe.hip && ./copy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <typename V, int U, bool NT, bool INPLACE>
__global__ __launch_bounds__(256) void k_rw(V *__restrict__ a, V *__restrict__ b, size_t n_vec, size_t chunk) {
  // each wave owns contiguous chunks of `chunk` vectors (like a tile of k_move_t), walks them with
  // U wave-rows in flight
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const size_t nwaves = (size_t)gridDim.x * 4;
  const size_t n_chunks = n_vec / chunk;
  V *dst = INPLACE ? a : b;
  for (size_t c = wave; c < n_chunks; c += nwaves) {
    const size_t base = c * chunk + lane;
    for (size_t k = 0; k < chunk; k += 64 * U) {
      V v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const V *p = a + base + k + (size_t)u * 64;
        v[u] = NT ? __builtin_nontemporal_load(p) : *p;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        v[u] = v[u] * 1.0001f + 0.5f;
        V *q = dst + base + k + (size_t)u * 64;
        if (NT) __builtin_nontemporal_store(v[u], q);
        else *q = v[u];
      }
    }
  }
}


#define TIME(label, V, U, NT, INP, grid)                                                          \
  do {                                                                                            \
    const size_t n_vec = bytes / sizeof(V);                                                       \
    const size_t chunk = (size_t)chunk_bytes / sizeof(V);                                         \
    hipEvent_t s, e;                                                                              \
    hipEventCreate(&s);                                                                           \
    hipEventCreate(&e);                                                                           \
    for (int i = 0; i < 2; ++i)                                                                   \
      hipLaunchKernelGGL((k_rw<V, U, NT, INP>), dim3(grid), dim3(256), 0, 0, (V *)a, (V *)b, n_vec, chunk); \
    hipEventRecord(s);                                                                            \
    for (int i = 0; i < 10; ++i)                                                                  \
      hipLaunchKernelGGL((k_rw<V, U, NT, INP>), dim3(grid), dim3(256), 0, 0, (V *)a, (V *)b, n_vec, chunk); \
    hipEventRecord(e);                                                                            \
    hipEventSynchronize(e);                                                                       \
    float ms = 0;                                                                                 \
    hipEventElapsedTime(&ms, s, e);                                                               \
    ms /= 10;                                                                                     \
    printf("%-44s grid %5d: %.3f ms  %.2f TB/s (read+write)\n", label, grid, ms, 2.0 * bytes / ms / 1e9); \
    hipEventDestroy(s);                                                                           \
    hipEventDestroy(e);                                                                           \
  } while (0)

int main(int argc, char **argv) {
  const size_t bytes = 1835008000ull;  // ~ the live cars of cfg2 (1.84 GB), a multiple of 32 KiB
  const size_t chunk_bytes = 32768;    // one tile of k_move_t at C = 66
  void *a, *b;
  if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) return 1;
  hipMemset(a, 0, bytes);
  hipMemset(b, 0, bytes);
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cu = prop.multiProcessorCount;
  for (int per_cu : {3, 5, 8}) {
    const int g = cu * per_cu;
    TIME("in place, 8 B/lane, 4 rows in flight", f2, 4, false, true, g);
    TIME("in place, 8 B/lane, 4 rows, nt", f2, 4, true, true, g);
    TIME("in place, 8 B/lane, 8 rows, nt", f2, 8, true, true, g);
    TIME("in place, 16 B/lane, 4 rows, nt", f4, 4, true, true, g);
    TIME("in place, 16 B/lane, 8 rows, nt", f4, 8, true, true, g);
    TIME("in place, 16 B/lane, 4 rows", f4, 4, false, true, g);
    TIME("a -> b, 16 B/lane, 4 rows, nt", f4, 4, true, false, g);
    TIME("a -> b, 8 B/lane, 4 rows, nt", f2, 4, true, false, g);
  }
  hipFree(a);
  hipFree(b);
  return 0;
}
