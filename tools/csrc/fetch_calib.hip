// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 against KNOWN byte counts, per access width (VERDICT r4 weak #5:
// MI355X_MICROARCH.md calibrates "FETCH_SIZE x 2" for 16 B / lane streaming reads only; k_wg3's patch rows are buffer_load_dwordx2 and
// dword halo loads).  Every kernel touches each byte of a 1 GiB buffer (4 x the Infinity Cache) exactly once, coalesced, in the width its
// name says; the readers fold what they read into one value per lane that is stored only if it is a NaN pattern that never occurs.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/csrc/fetch_calib tools/csrc/fetch_calib.hip
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out/fetch -o run -- tools/csrc/fetch_calib
//   rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out/write -o run -- tools/csrc/fetch_calib
//   python tools/fetch_calib_table.py out   -> profiles/r05_fetch_calib.md
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// n = number of VEC-wide items; every thread strides over the buffer by the grid size (consecutive lanes = consecutive items)
template <class V>
__global__ __launch_bounds__(256) void k_read_global(const V *__restrict__ p, int64_t n, float *__restrict__ sink) {
    float acc = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const V v = p[i];
        const float *f = reinterpret_cast<const float *>(&v);
#pragma unroll
        for (int k = 0; k < (int)(sizeof(V) / 4); ++k) acc += f[k];
    }
    if (acc == 1.2345e38f) sink[threadIdx.x] = acc;
}

// raw buffer loads (the instruction family the Winograd staging uses): each block walks its own 1 MiB windows through a descriptor
template <int W>   // bytes per lane: 4, 8, 16
__global__ __launch_bounds__(256) void k_read_buffer(const float *__restrict__ p, int64_t bytes, float *__restrict__ sink) {
    constexpr int64_t WIN = 1 << 20;
    float acc = 0.0f;
    for (int64_t w = (int64_t)blockIdx.x * WIN; w < bytes; w += (int64_t)gridDim.x * WIN) {
        const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)p + w), 0, (int)WIN, 0x00020000);
        for (int off = threadIdx.x * W; off < WIN; off += 256 * W) {
            if (W == 4) {
                acc += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd, off, 0, 0));
            } else if (W == 8) {
                const i32x2 v = __builtin_amdgcn_raw_buffer_load_b64(srd, off, 0, 0);
                acc += __builtin_bit_cast(float, v[0]) + __builtin_bit_cast(float, v[1]);
            } else {
                const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(srd, off, 0, 0);
                acc += __builtin_bit_cast(float, v[0]) + __builtin_bit_cast(float, v[1]) + __builtin_bit_cast(float, v[2]) + __builtin_bit_cast(float, v[3]);
            }
        }
    }
    if (acc == 1.2345e38f) sink[threadIdx.x] = acc;
}

// the Winograd patch-row pattern: a wave reads 32 column PAIRS (8 B / lane, 256 contiguous bytes) of one image row, rows 448 B apart
// (a 112-pixel-wide map), i.e. half of every 512-byte stretch is touched by this wave and the other half by its neighbour wave
__global__ __launch_bounds__(256) void k_read_rows_b64(const float *__restrict__ p, int64_t bytes, float *__restrict__ sink) {
    constexpr int64_t WIN = 1 << 20;
    float acc = 0.0f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t w = (int64_t)blockIdx.x * WIN; w < bytes; w += (int64_t)gridDim.x * WIN) {
        const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)p + w), 0, (int)WIN, 0x00020000);
        // 1 MiB = 2048 pieces of 512 B; wave w takes pieces w, w + 4, ...; lanes 0-31 the first 256 B, lanes 32-63 the second 256 B
        for (int piece = wave; piece < 2048; piece += 4) {
            const i32x2 v = __builtin_amdgcn_raw_buffer_load_b64(srd, piece * 512 + lane * 8, 0, 0);
            acc += __builtin_bit_cast(float, v[0]) + __builtin_bit_cast(float, v[1]);
        }
    }
    if (acc == 1.2345e38f) sink[threadIdx.x] = acc;
}

template <class V>
__global__ __launch_bounds__(256) void k_write_global(V *__restrict__ p, int64_t n, float val) {
    V v;
    float *f = reinterpret_cast<float *>(&v);
#pragma unroll
    for (int k = 0; k < (int)(sizeof(V) / 4); ++k) f[k] = val + (float)k;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

int main() {
    const int64_t bytes = (int64_t)1 << 30;
    float *buf = nullptr, *sink = nullptr;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMalloc(&sink, 4096));
    CHECK(hipMemset(buf, 0, bytes));
    CHECK(hipDeviceSynchronize());
    const int grid = 256 * 8;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_read_global<float>, dim3(grid), dim3(256), 0, 0, (const float *)buf, bytes / 4, sink);
        hipLaunchKernelGGL(k_read_global<f32x2>, dim3(grid), dim3(256), 0, 0, (const f32x2 *)buf, bytes / 8, sink);
        hipLaunchKernelGGL(k_read_global<f32x4>, dim3(grid), dim3(256), 0, 0, (const f32x4 *)buf, bytes / 16, sink);
        hipLaunchKernelGGL(k_read_buffer<4>, dim3(1024), dim3(256), 0, 0, (const float *)buf, bytes, sink);
        hipLaunchKernelGGL(k_read_buffer<8>, dim3(1024), dim3(256), 0, 0, (const float *)buf, bytes, sink);
        hipLaunchKernelGGL(k_read_buffer<16>, dim3(1024), dim3(256), 0, 0, (const float *)buf, bytes, sink);
        hipLaunchKernelGGL(k_read_rows_b64, dim3(1024), dim3(256), 0, 0, (const float *)buf, bytes, sink);
        hipLaunchKernelGGL(k_write_global<float>, dim3(grid), dim3(256), 0, 0, buf, bytes / 4, 1.0f);
        hipLaunchKernelGGL(k_write_global<f32x2>, dim3(grid), dim3(256), 0, 0, (f32x2 *)buf, bytes / 8, 2.0f);
        hipLaunchKernelGGL(k_write_global<f32x4>, dim3(grid), dim3(256), 0, 0, (f32x4 *)buf, bytes / 16, 3.0f);
        CHECK(hipDeviceSynchronize());
    }
    printf("fetch_calib: every kernel touched %lld bytes once per launch, 3 launches each\n", (long long)bytes);
    return 0;
}
