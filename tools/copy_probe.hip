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


// k_model: the same in-place stream with k_move_t's features switched on one by one
//   RAGGED  per-lane road lengths 40..56 rows (partial rows at the tails)
//   SHIFT   one lane in ten writes its rows one row up (NOT representative of k_move_t's compaction:
//           the kernel itself is no faster with the shift ablated, TFX_DEBUG-style test; not timed)
//   WORDS   per tile: 8 dependent-free word loads per lane before the walk, 5 word stores after it
//   MATH    ~70 dependent float operations (one true division) between a row's load and its store
template <int U, bool RAGGED, bool SHIFT, bool WORDS, bool MATH>
__global__ __launch_bounds__(256) void k_model(f2 *__restrict__ a, int *__restrict__ words, size_t n_tiles) {
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const size_t nwaves = (size_t)gridDim.x * 4;
  for (size_t t = wave; t < n_tiles; t += nwaves) {
    f2 *col = a + t * 4096 + lane;  // 64 rows x 64 lanes
    unsigned h = (unsigned)(t * 64 + lane) * 2654435761u;
    int n = RAGGED ? 40 + (int)((h >> 8) % 17u) : 48;
    const int shift = (SHIFT && ((h >> 20) % 10u) == 0) ? 1 : 0;
    float acc = 0.f;
    if (WORDS) {
      int *w = words + (t * 64 + lane);
      int s0 = 0;
#pragma unroll
      for (int q = 0; q < 8; ++q) s0 += w[q * n_tiles * 64];
      n += (s0 & 0);  // keep the loads alive, leave n unchanged
      acc = (float)(s0 & 1);
    }
    int kmax = n;
    for (int off = 32; off > 0; off >>= 1) {
      const int o = __shfl_xor(kmax, off, 64);
      kmax = o > kmax ? o : kmax;
    }
    kmax = __builtin_amdgcn_readfirstlane(kmax);
    f2 pf[U];
#pragma unroll
    for (int u = 0; u < U; ++u) pf[u] = (u < n) ? __builtin_nontemporal_load(col + (size_t)u * 64) : f2{0, 0};
    float px = 1e9f, pv = 0.f;
    for (int k0 = 0; k0 < kmax; k0 += U) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int k = k0 + u;
        if (k < kmax) {
          f2 c = pf[u];
          if (k + U < kmax) pf[u] = (k + U < n) ? __builtin_nontemporal_load(col + (size_t)(k + U) * 64) : f2{0, 0};
          if (k < n) {
            f2 o = c;
            if (MATH) {
              float s = px - c.x - 4.f, q = c.y * 0.072f;
              float st = 1.f + c.y * 2.f + c.y * (c.y - pv) * 0.1178f;
              float r = st / (s + 1e-8f);
              float dv = 3.f * (1.f - q * q * q * q - r * r);
#pragma unroll
              for (int z = 0; z < 12; ++z) dv = dv * 0.999f + 0.001f * r;  // stand-in for the bookkeeping chain
              o.x = c.x + fmaxf(0.f, 0.5f * c.y + 0.125f * dv);
              o.y = fmaxf(0.f, c.y + 0.5f * dv);
              px = c.x;
              pv = c.y;
            } else {
              o.x = c.x * 1.0001f + 0.5f;
            }
            const int row = (k - shift < 0) ? 0 : k - shift;
            __builtin_nontemporal_store(o, col + (size_t)row * 64);
            acc += o.x;
          }
        }
      }
    }
    if (WORDS) {
      int *w = words + (t * 64 + lane);
#pragma unroll
      for (int q = 0; q < 5; ++q) w[(8 + q) * n_tiles * 64] = (int)acc + q;
    }
  }
}

// k_model_g: the same stream in k_rw's 8-row shape - the G loads of the NEXT group issued back to back
// at the top of a group, the group computed in registers, its G stores issued back to back - with
// k_move_t's ragged road lengths, per-road words and arithmetic
template <int G, bool RAGGED, bool WORDS, bool MATH>
__global__ __launch_bounds__(256) void k_model_g(f2 *__restrict__ a, int *__restrict__ words, size_t n_tiles) {
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const size_t nwaves = (size_t)gridDim.x * 4;
  for (size_t t = wave; t < n_tiles; t += nwaves) {
    f2 *col = a + t * 4096 + lane;
    unsigned h = (unsigned)(t * 64 + lane) * 2654435761u;
    int n = RAGGED ? 40 + (int)((h >> 8) % 17u) : 48;
    float acc = 0.f;
    if (WORDS) {
      int *w = words + (t * 64 + lane);
      int s0 = 0;
#pragma unroll
      for (int q = 0; q < 8; ++q) s0 += w[q * n_tiles * 64];
      n += (s0 & 0);
      acc = (float)(s0 & 1);
    }
    int kmax = n;
    for (int off = 32; off > 0; off >>= 1) {
      const int o = __shfl_xor(kmax, off, 64);
      kmax = o > kmax ? o : kmax;
    }
    kmax = __builtin_amdgcn_readfirstlane(kmax);
    f2 pf[G];
#pragma unroll
    for (int u = 0; u < G; ++u) pf[u] = (u < n) ? __builtin_nontemporal_load(col + (size_t)u * 64) : f2{0, 0};
    float px = 1e9f, pv = 0.f;
    for (int k0 = 0; k0 < kmax; k0 += G) {
      f2 cur[G];
#pragma unroll
      for (int u = 0; u < G; ++u) cur[u] = pf[u];
      if (k0 + G < kmax) {
#pragma unroll
        for (int u = 0; u < G; ++u)
          pf[u] = (k0 + G + u < n) ? __builtin_nontemporal_load(col + (size_t)(k0 + G + u) * 64) : f2{0, 0};
      }
#pragma unroll
      for (int u = 0; u < G; ++u) {
        const f2 c = cur[u];
        if (k0 + u < n) {
          f2 o = c;
          if (MATH) {
            float s = px - c.x - 4.f, q = c.y * 0.072f;
            float st = 1.f + c.y * 2.f + c.y * (c.y - pv) * 0.1178f;
            float r = st / (s + 1e-8f);
            float dv = 3.f * (1.f - q * q * q * q - r * r);
#pragma unroll
            for (int z = 0; z < 12; ++z) dv = dv * 0.999f + 0.001f * r;
            o.x = c.x + fmaxf(0.f, 0.5f * c.y + 0.125f * dv);
            o.y = fmaxf(0.f, c.y + 0.5f * dv);
            px = c.x;
            pv = c.y;
          } else {
            o.x = c.x * 1.0001f + 0.5f;
          }
          cur[u] = o;
          acc += o.x;
        }
      }
#pragma unroll
      for (int u = 0; u < G; ++u)
        if (k0 + u < n) __builtin_nontemporal_store(cur[u], col + (size_t)(k0 + u) * 64);
    }
    if (WORDS) {
      int *w = words + (t * 64 + lane);
#pragma unroll
      for (int q = 0; q < 5; ++q) w[(8 + q) * n_tiles * 64] = (int)acc + q;
    }
  }
}

// k_model_i: k_model's rolling stream on an INTERLEAVED layout - the rows k of the NT tiles of a
// workgroup are contiguous (NT * 512 bytes), each wavefront still walks its own tile: do DRAM pages
// like the longer contiguous runs?
template <int U, int NT, bool RAGGED, bool WORDS, bool MATH>
__global__ __launch_bounds__(256) void k_model_i(f2 *__restrict__ a, int *__restrict__ words, size_t n_tiles) {
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const size_t n_groups = n_tiles / 4;
  for (size_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
    const size_t t = g * 4 + wv;
    // group base: 4 tiles x 64 rows x 64 lanes; row k of sub-tile s at base + (k * NT + s') * 64 ...
    f2 *col = (NT == 4) ? a + g * 4 * 4096 + (size_t)wv * 64 + lane
                        : a + g * 4 * 4096 + (size_t)(wv >> 1) * 2 * 4096 + (size_t)(wv & 1) * 64 + lane;
    const size_t rstride = (size_t)NT * 64;
    unsigned h = (unsigned)(t * 64 + lane) * 2654435761u;
    int n = RAGGED ? 40 + (int)((h >> 8) % 17u) : 48;
    float acc = 0.f;
    if (WORDS) {
      int *w = words + (t * 64 + lane);
      int s0 = 0;
#pragma unroll
      for (int q = 0; q < 8; ++q) s0 += w[q * n_tiles * 64];
      n += (s0 & 0);
      acc = (float)(s0 & 1);
    }
    int kmax = n;
    for (int off = 32; off > 0; off >>= 1) {
      const int o = __shfl_xor(kmax, off, 64);
      kmax = o > kmax ? o : kmax;
    }
    kmax = __builtin_amdgcn_readfirstlane(kmax);
    f2 pf[U];
#pragma unroll
    for (int u = 0; u < U; ++u) pf[u] = (u < n) ? __builtin_nontemporal_load(col + (size_t)u * rstride) : f2{0, 0};
    float px = 1e9f, pv = 0.f;
    for (int k0 = 0; k0 < kmax; k0 += U) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int k = k0 + u;
        if (k < kmax) {
          f2 c = pf[u];
          if (k + U < kmax) pf[u] = (k + U < n) ? __builtin_nontemporal_load(col + (size_t)(k + U) * rstride) : f2{0, 0};
          if (k < n) {
            f2 o = c;
            if (MATH) {
              float s = px - c.x - 4.f, q = c.y * 0.072f;
              float st = 1.f + c.y * 2.f + c.y * (c.y - pv) * 0.1178f;
              float r = st / (s + 1e-8f);
              float dv = 3.f * (1.f - q * q * q * q - r * r);
#pragma unroll
              for (int z = 0; z < 12; ++z) dv = dv * 0.999f + 0.001f * r;
              o.x = c.x + fmaxf(0.f, 0.5f * c.y + 0.125f * dv);
              o.y = fmaxf(0.f, c.y + 0.5f * dv);
              px = c.x;
              pv = c.y;
            } else {
              o.x = c.x * 1.0001f + 0.5f;
            }
            __builtin_nontemporal_store(o, col + (size_t)k * rstride);
            acc += o.x;
          }
        }
      }
    }
    if (WORDS) {
      int *w = words + (t * 64 + lane);
#pragma unroll
      for (int q = 0; q < 5; ++q) w[(8 + q) * n_tiles * 64] = (int)acc + q;
    }
  }
}

#define MODELI(label, U, NT, RG, WD, MA, grid)                                                     \
  do {                                                                                            \
    const size_t n_tiles = bytes / 32768;                                                         \
    hipEvent_t s, e;                                                                              \
    hipEventCreate(&s);                                                                           \
    hipEventCreate(&e);                                                                           \
    for (int i = 0; i < 2; ++i)                                                                   \
      hipLaunchKernelGGL((k_model_i<U, NT, RG, WD, MA>), dim3(grid), dim3(256), 0, 0, (f2 *)a, (int *)b, n_tiles); \
    hipEventRecord(s);                                                                            \
    for (int i = 0; i < 10; ++i)                                                                  \
      hipLaunchKernelGGL((k_model_i<U, NT, RG, WD, MA>), dim3(grid), dim3(256), 0, 0, (f2 *)a, (int *)b, n_tiles); \
    hipEventRecord(e);                                                                            \
    hipEventSynchronize(e);                                                                       \
    float ms = 0;                                                                                 \
    hipEventElapsedTime(&ms, s, e);                                                               \
    ms /= 10;                                                                                     \
    printf("model %-38s grid %5d: %.3f ms\n", label, grid, ms);                                   \
    hipEventDestroy(s);                                                                           \
    hipEventDestroy(e);                                                                           \
  } while (0)

#define MODELG(label, G, RG, WD, MA, grid)                                                         \
  do {                                                                                            \
    const size_t n_tiles = bytes / 32768;                                                         \
    hipEvent_t s, e;                                                                              \
    hipEventCreate(&s);                                                                           \
    hipEventCreate(&e);                                                                           \
    for (int i = 0; i < 2; ++i)                                                                   \
      hipLaunchKernelGGL((k_model_g<G, RG, WD, MA>), dim3(grid), dim3(256), 0, 0, (f2 *)a, (int *)b, n_tiles); \
    hipEventRecord(s);                                                                            \
    for (int i = 0; i < 10; ++i)                                                                  \
      hipLaunchKernelGGL((k_model_g<G, RG, WD, MA>), dim3(grid), dim3(256), 0, 0, (f2 *)a, (int *)b, n_tiles); \
    hipEventRecord(e);                                                                            \
    hipEventSynchronize(e);                                                                       \
    float ms = 0;                                                                                 \
    hipEventElapsedTime(&ms, s, e);                                                               \
    ms /= 10;                                                                                     \
    printf("model %-38s grid %5d: %.3f ms\n", label, grid, ms);                                   \
    hipEventDestroy(s);                                                                           \
    hipEventDestroy(e);                                                                           \
  } while (0)

#define MODEL(label, U, RG, SH, WD, MA, grid)                                                      \
  do {                                                                                            \
    const size_t n_tiles = bytes / 32768;                                                         \
    hipEvent_t s, e;                                                                              \
    hipEventCreate(&s);                                                                           \
    hipEventCreate(&e);                                                                           \
    for (int i = 0; i < 2; ++i)                                                                   \
      hipLaunchKernelGGL((k_model<U, RG, SH, WD, MA>), dim3(grid), dim3(256), 0, 0, (f2 *)a, (int *)b, n_tiles); \
    hipEventRecord(s);                                                                            \
    for (int i = 0; i < 10; ++i)                                                                  \
      hipLaunchKernelGGL((k_model<U, RG, SH, WD, MA>), dim3(grid), dim3(256), 0, 0, (f2 *)a, (int *)b, n_tiles); \
    hipEventRecord(e);                                                                            \
    hipEventSynchronize(e);                                                                       \
    float ms = 0;                                                                                 \
    hipEventElapsedTime(&ms, s, e);                                                               \
    ms /= 10;                                                                                     \
    printf("model %-38s grid %5d: %.3f ms\n", label, grid, ms);                                   \
    hipEventDestroy(s);                                                                           \
    hipEventDestroy(e);                                                                           \
  } while (0)

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
    // the per-road words of k_tail: 4 bytes per lane, default caching
    TIME("in place, 4 B/lane, 4 rows", float, 4, false, true, g);
    TIME("a -> b, 4 B/lane, 4 rows", float, 4, false, false, g);
  }
  // 48 of 64 rows live per road, as in the benchmark: every model line moves ~2.75 GB
  {
    const int g = cu * 5;
    MODEL("48 rows per road, 4 in flight", 4, false, false, false, false, g);
    MODEL("road lengths 40..56 (ragged tails)", 4, true, false, false, false, g);
    MODEL("48 rows + per-road words", 4, false, false, true, false, g);
    MODEL("48 rows + arithmetic", 4, false, false, false, true, g);
    MODEL("ragged + words + arithmetic", 4, true, false, true, true, g);
    MODEL("ragged + words + arithmetic, 8 rolling", 8, true, false, true, true, g);
    MODELG("grouped 8: 48 rows per road", 8, false, false, false, g);
    MODELG("grouped 8: ragged tails", 8, true, false, false, g);
    MODELG("grouped 8: ragged + words + arithmetic", 8, true, true, true, g);
    MODELG("grouped 4: ragged + words + arithmetic", 4, true, true, true, g);
    MODELG("grouped 16: ragged + words + arithmetic", 16, true, true, true, g);
    MODELG("grouped 8: ragged + words + arith, 4/CU", 8, true, true, true, cu * 4);
    MODELG("grouped 8: ragged + words + arith, 6/CU", 8, true, true, true, cu * 6);
    MODELI("interleaved x4: 48 rows per road", 4, 4, false, false, false, g);
    MODELI("interleaved x4: ragged + words + arith", 4, 4, true, true, true, g);
    MODELI("interleaved x2: ragged + words + arith", 4, 2, true, true, true, g);
    MODELI("interleaved x4, 8 rolling: ragged+w+a", 8, 4, true, true, true, g);
  }
  hipFree(a);
  hipFree(b);
  return 0;
}
