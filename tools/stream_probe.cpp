// tools/stream_probe.cpp -- micro-benchmark behind DESIGN.md section 5: bytes/s of a streaming kernel as a function of how the same bytes are
// split into streams.  Build: hipcc --offload-arch=gfx950 -O3 tools/stream_probe.cpp -o _ab/stream_probe ; run it on the GPU box.  Not part of the library.
//   R read streams + W write streams, each thread moves CH x 16 B per stream (CH consecutive float4: AoSoA-style chunk)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int R, int W, int CH>
__global__ __launch_bounds__(256) void k(const f4 *__restrict__ in, f4 *__restrict__ out, size_t n_vec /* per stream, in units of CH float4 */) {
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < n_vec; v += (size_t)gridDim.x * 256) {
        f4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc += __builtin_nontemporal_load(in + ((size_t)r * n_vec + v) * CH + c);
#pragma unroll
        for (int w = 0; w < W; ++w)
#pragma unroll
            for (int c = 0; c < CH; ++c) __builtin_nontemporal_store(acc + (float)(w + c), out + ((size_t)w * n_vec + v) * CH + c);
    }
}
template <int R, int W, int CH>
double run(const f4 *in, f4 *out, size_t bytes_per_dir_read, int reps) {
    const size_t n_vec = bytes_per_dir_read / 16 / R / CH;   // read bytes fixed; write bytes = read * W / R
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int grid = 4096;
    hipLaunchKernelGGL((k<R, W, CH>), dim3(grid), dim3(256), 0, 0, in, out, n_vec);
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<R, W, CH>), dim3(grid), dim3(256), 0, 0, in, out, n_vec);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes = double(n_vec) * CH * 16 * (R + W) * reps;
    return bytes / (ms * 1e-3) / 1e12;
}
// a copy with U independent float4 in flight per thread (what a tuned streaming kernel looks like)
template <int U>
__global__ __launch_bounds__(256) void kcopy(const f4 *__restrict__ in, f4 *__restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < n; v += stride * U) {
        f4 x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = v + u * stride < n ? __builtin_nontemporal_load(in + v + u * stride) : f4{0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (v + u * stride < n) __builtin_nontemporal_store(x[u], out + v + u * stride);
    }
}
template <int U>
double run_copy(const f4 *in, f4 *out, size_t bytes, int grid, int reps) {
    const size_t n = bytes / 16;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((kcopy<U>), dim3(grid), dim3(256), 0, 0, in, out, n);
    (void)hipEventRecord(a);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((kcopy<U>), dim3(grid), dim3(256), 0, 0, in, out, n);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    return 2.0 * double(n) * 16 * reps / (ms * 1e-3) / 1e12;
}

// The same R + W streams, but interleaved at TILE bytes: tile t of stream s sits at (t * (R or W) + s) * TILE, so that the rows a workgroup walks together are
// one contiguous region (a few pages) instead of R + W regions tens of MB apart.  A wave still reads 1 KB contiguous per stream.
template <int R, int W, int TILE_VEC>
__global__ __launch_bounds__(256) void ktiled(const f4 *__restrict__ in, f4 *__restrict__ out, size_t n_vec) {
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < n_vec; v += (size_t)gridDim.x * 256) {
        const size_t tile = v / TILE_VEC, within = v % TILE_VEC;
        f4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < R; ++r) acc += __builtin_nontemporal_load(in + (tile * R + r) * TILE_VEC + within);
#pragma unroll
        for (int w = 0; w < W; ++w) __builtin_nontemporal_store(acc + (float)w, out + (tile * W + w) * TILE_VEC + within);
    }
}
template <int R, int W, int TILE_VEC>
double run_tiled(const f4 *in, f4 *out, size_t bytes_read, int reps) {
    const size_t n_vec = bytes_read / 16 / R / TILE_VEC * TILE_VEC;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((ktiled<R, W, TILE_VEC>), dim3(4096), dim3(256), 0, 0, in, out, n_vec);
    (void)hipEventRecord(a);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((ktiled<R, W, TILE_VEC>), dim3(4096), dim3(256), 0, 0, in, out, n_vec);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    return double(n_vec) * 16 * (R + W) * reps / (ms * 1e-3) / 1e12;
}

// Per-NODE tiling: the streams come in NG groups (a node's rows of one array), each group a contiguous block [tile][rows of the group][TILE]; groups
// stay tens of MB apart.  What the table would look like if only the inside of a node block were re-laid out.
template <int NG, int RG, int WG, int TILE_VEC>
__global__ __launch_bounds__(256) void kgrouped(const f4 *__restrict__ in, f4 *__restrict__ out, size_t n_vec) {
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < n_vec; v += (size_t)gridDim.x * 256) {
        const size_t tile = v / TILE_VEC, within = v % TILE_VEC;
        f4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int r = 0; r < RG; ++r) acc += __builtin_nontemporal_load(in + (size_t)g * RG * n_vec + (tile * RG + r) * TILE_VEC + within);
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int w = 0; w < WG; ++w) __builtin_nontemporal_store(acc + (float)w, out + (size_t)g * WG * n_vec + (tile * WG + w) * TILE_VEC + within);
    }
}
template <int NG, int RG, int WG, int TILE_VEC>
double run_grouped(const f4 *in, f4 *out, size_t bytes_read, int reps) {
    const size_t n_vec = bytes_read / 16 / (NG * RG) / TILE_VEC * TILE_VEC;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((kgrouped<NG, RG, WG, TILE_VEC>), dim3(4096), dim3(256), 0, 0, in, out, n_vec);
    (void)hipEventRecord(a);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((kgrouped<NG, RG, WG, TILE_VEC>), dim3(4096), dim3(256), 0, 0, in, out, n_vec);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    return double(n_vec) * 16 * NG * (RG + WG) * reps / (ms * 1e-3) / 1e12;
}

int main() {
    const size_t rd = size_t(2200) << 20;   // ~2.2 GB read per launch like the tree kernel (2.14 GB), writes scaled 2:3.. by W/R
    f4 *in, *out; hipMalloc(&in, rd + (64 << 20)); hipMalloc(&out, rd + (64 << 20));
    hipMemset(in, 0, rd); hipMemset(out, 0, rd);
    printf("copy        R1  W1  CH1 : %.2f TB/s\n", run<1, 1, 1>(in, out, rd, 10));
    printf("copy        R1  W1  CH4 : %.2f TB/s\n", run<1, 1, 4>(in, out, rd, 10));
    printf("3:2         R3  W2  CH1 : %.2f TB/s\n", run<3, 2, 1>(in, out, rd, 10));
    printf("many        R12 W8  CH1 : %.2f TB/s\n", run<12, 8, 1>(in, out, rd, 10));
    printf("many        R30 W20 CH1 : %.2f TB/s\n", run<30, 20, 1>(in, out, rd, 10));
    printf("tree-like   R57 W38 CH1 : %.2f TB/s\n", run<57, 38, 1>(in, out, rd, 10));
    printf("AoSoA-like  R19 W13 CH3 : %.2f TB/s\n", run<19, 13, 3>(in, out, rd, 10));
    printf("AoSoA-like  R10 W7  CH6 : %.2f TB/s\n", run<10, 7, 6>(in, out, rd, 10));

    for (int grid : {1024, 4096, 16384})
        printf("tuned copy grid %5d: U=1 %.2f  U=4 %.2f  U=8 %.2f TB/s\n", grid, run_copy<1>(in, out, rd, grid, 10), run_copy<4>(in, out, rd, grid, 10),
               run_copy<8>(in, out, rd, grid, 10));
    printf("tree-like R57 W38, rows interleaved in tiles of 4 KB: %.2f TB/s, 16 KB: %.2f TB/s, 64 KB: %.2f TB/s; rows apart (as the table is laid out): %.2f TB/s\n",
           run_tiled<57, 38, 256>(in, out, rd, 10), run_tiled<57, 38, 1024>(in, out, rd, 10), run_tiled<57, 38, 4096>(in, out, rd, 10), run<57, 38, 1>(in, out, rd, 10));
    printf("19 groups of 3 read + 2 written rows, tiled inside the group: 4 KB %.2f TB/s, 16 KB %.2f TB/s, 64 KB %.2f TB/s\n", run_grouped<19, 3, 2, 256>(in, out, rd, 10),
           run_grouped<19, 3, 2, 1024>(in, out, rd, 10), run_grouped<19, 3, 2, 4096>(in, out, rd, 10));
    return 0;
}
