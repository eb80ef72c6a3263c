// Achieved HBM bandwidth of the access patterns the transform kernels could use (tools only).
//   hipcc -O3 --offload-arch=gfx950 tools/mem_probe.hip -o tools/_mem_probe && tools/_mem_probe
// Each kernel moves a 3840x2176 f32 plane set (3 planes, 100 MB in + 100 MB out) per launch; eight
// distinct buffer sets are cycled so nothing survives in the 256 MB Infinity Cache.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

constexpr int W = 3840, H = 2176, W8 = W / 8, H8 = H / 8;
constexpr size_t PLANE = (size_t)W * H;

// pattern A: plain copy, 16 B per lane, fully coalesced
__global__ __launch_bounds__(256) void copy16(const float4* __restrict__ in, float4* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[i];
}
// pattern B: read = row-lane from CELL-TILED input (lane = (block, row): 2 x 16 B at 32 B stride),
//            write = dword stores, lane = (block, x) of a ROW-MAJOR plane, 8 blocks per wave taken
//            from a shuffled list (like a class-sorted work list)
__global__ __launch_bounds__(64) void rowlane_dword(const float* __restrict__ in, float* __restrict__ out,
                                                     const uint32_t* __restrict__ cells, int ncells) {
    __shared__ float T[64 * 9];
    const int lane = threadIdx.x, blk = lane >> 3, r = lane & 7;
    for (int c = 0; c < 3; ++c) {
        const uint32_t cell = cells[min((int)blockIdx.x * 8 + blk, ncells - 1)];
        const uint32_t cx = cell & 0xffff, cy = cell >> 16;
        const float* src = in + (((size_t)cy * W8 + cx) * 3 + c) * 64 + r * 8;
        const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
        float* row = T + lane * 9;
        row[0] = a.x; row[1] = a.y; row[2] = a.z; row[3] = a.w; row[4] = b.x; row[5] = b.y; row[6] = b.z; row[7] = b.w;
        __builtin_amdgcn_wave_barrier();
        float* dst = out + c * PLANE + (size_t)(cy * 8) * W + cx * 8 + r;  // lane = (blk, x = r)
#pragma unroll
        for (int y = 0; y < 8; ++y) dst[(size_t)y * W] = T[(blk * 8 + y) * 9 + r];
        __builtin_amdgcn_wave_barrier();
    }
}
// pattern C: same read, write = 2 x 16 B per lane (lane = (block, row)) into a ROW-MAJOR plane
__global__ __launch_bounds__(64) void rowlane_x4(const float* __restrict__ in, float* __restrict__ out,
                                                  const uint32_t* __restrict__ cells, int ncells) {
    const int lane = threadIdx.x, blk = lane >> 3, r = lane & 7;
    for (int c = 0; c < 3; ++c) {
        const uint32_t cell = cells[min((int)blockIdx.x * 8 + blk, ncells - 1)];
        const uint32_t cx = cell & 0xffff, cy = cell >> 16;
        const float* src = in + (((size_t)cy * W8 + cx) * 3 + c) * 64 + r * 8;
        const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
        float* dst = out + c * PLANE + (size_t)(cy * 8 + r) * W + cx * 8;
        *reinterpret_cast<float4*>(dst) = a;
        *reinterpret_cast<float4*>(dst + 4) = b;
    }
}
// pattern D: same read, write = CELL-TILED output (2 x 16 B per lane, whole 256-byte cells)
__global__ __launch_bounds__(64) void rowlane_tiled(const float* __restrict__ in, float* __restrict__ out,
                                                     const uint32_t* __restrict__ cells, int ncells) {
    const int lane = threadIdx.x, blk = lane >> 3, r = lane & 7;
    for (int c = 0; c < 3; ++c) {
        const uint32_t cell = cells[min((int)blockIdx.x * 8 + blk, ncells - 1)];
        const uint32_t cx = cell & 0xffff, cy = cell >> 16;
        const size_t off = (((size_t)cy * W8 + cx) * 3 + c) * 64 + r * 8;
        const float4 a = *reinterpret_cast<const float4*>(in + off), b = *reinterpret_cast<const float4*>(in + off + 4);
        *reinterpret_cast<float4*>(out + off) = a;
        *reinterpret_cast<float4*>(out + off + 4) = b;
    }
}
// pattern E: coalesced read of the tiled input (lane i = 16-byte chunk i), tiled write
__global__ __launch_bounds__(64) void chunk_tiled(const float* __restrict__ in, float* __restrict__ out,
                                                   const uint32_t* __restrict__ cells, int ncells) {
    const int lane = threadIdx.x;
    for (int j = 0; j < 6; ++j) {  // 8 cells x 3 channels x 256 B = 6 KiB = 6 wave loads
        const int chunk = j * 64 + lane, cellidx = chunk / 48, within = chunk % 48;
        const uint32_t cell = cells[min((int)blockIdx.x * 8 + cellidx, ncells - 1)];
        const uint32_t cx = cell & 0xffff, cy = cell >> 16;
        const size_t off = ((size_t)cy * W8 + cx) * 192 + within * 4;
        *reinterpret_cast<float4*>(out + off) = *reinterpret_cast<const float4*>(in + off);
    }
}
// pattern F: post-kernel style read of a ROW-MAJOR plane (lane = column, walks 56 rows), row-major write
__global__ __launch_bounds__(256) void walk_rowmajor(const float* __restrict__ in, float* __restrict__ out) {
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int strips = W / 64, strip = wave % strips, seg = wave / strips;
    if (seg * 48 >= H) return;
    const int x = strip * 64 + lane;
    for (int y = seg * 48; y < min(seg * 48 + 48, H); ++y)
        for (int c = 0; c < 3; ++c) out[c * PLANE + (size_t)y * W + x] = in[c * PLANE + (size_t)y * W + x];
}
// pattern G: the same walk reading a CELL-TILED input (8 x 32-byte pieces per wave load, rows of a cell
//            share cache lines), row-major write
__global__ __launch_bounds__(256) void walk_tiled(const float* __restrict__ in, float* __restrict__ out) {
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int strips = W / 64, strip = wave % strips, seg = wave / strips;
    if (seg * 48 >= H) return;
    const int x = strip * 64 + lane;
    for (int y = seg * 48; y < min(seg * 48 + 48, H); ++y)
        for (int c = 0; c < 3; ++c)
            out[c * PLANE + (size_t)y * W + x] = in[(((size_t)(y >> 3) * W8 + (x >> 3)) * 3 + c) * 64 + (y & 7) * 8 + (x & 7)];
}

int main() {
    const int NSET = 8;
    std::vector<float*> in(NSET), out(NSET);
    for (int i = 0; i < NSET; ++i) {
        if (hipMalloc(&in[i], PLANE * 12) != hipSuccess || hipMalloc(&out[i], PLANE * 12) != hipSuccess) return 3;
        (void)hipMemset(in[i], 1, PLANE * 12);
        (void)hipMemset(out[i], 0, PLANE * 12);
    }
    // 40 % of the cells, in raster order (a class-sorted list of scattered 8x8 varblocks) / all cells
    std::vector<uint32_t> sparse, dense;
    srand(1);
    for (int cy = 0; cy < H8; ++cy)
        for (int cx = 0; cx < W8; ++cx) {
            dense.push_back(cx | (cy << 16));
            if (rand() % 100 < 40) sparse.push_back(cx | (cy << 16));
        }
    uint32_t *d_sparse, *d_dense;
    (void)hipMalloc(&d_sparse, sparse.size() * 4); (void)hipMalloc(&d_dense, dense.size() * 4);
    (void)hipMemcpy(d_sparse, sparse.data(), sparse.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_dense, dense.data(), dense.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto time = [&](const char* name, double bytes, auto launch) {
        for (int i = 0; i < NSET; ++i) launch(i);
        (void)hipEventRecord(e0, 0);
        const int reps = 5;
        for (int r = 0; r < reps; ++r)
            for (int i = 0; i < NSET; ++i) launch(i);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / (reps * NSET);
        printf("%-58s %8.1f us  %6.2f TB/s (read + write)\n", name, us, bytes / us / 1e6);
    };
    const double full = PLANE * 24.0;
    const size_t n16 = PLANE * 3 / 4;
    time("A copy 16B/lane coalesced", full, [&](int i) { copy16<<<(n16 + 255) / 256, 256>>>((const float4*)in[i], (float4*)out[i], n16); });
    for (int pass = 0; pass < 2; ++pass) {
        const uint32_t* cells = pass ? d_dense : d_sparse;
        const int nc = pass ? (int)dense.size() : (int)sparse.size();
        const double bytes = nc * 64.0 * 3 * 8;
        const char* tag = pass ? "all cells  " : "40% of cells";
        char name[128];
        snprintf(name, sizeof name, "B tiled row-lane read, dword stores row-major  [%s]", tag);
        time(name, bytes, [&](int i) { rowlane_dword<<<(nc + 7) / 8, 64>>>(in[i], out[i], cells, nc); });
        snprintf(name, sizeof name, "C tiled row-lane read, 16B stores row-major    [%s]", tag);
        time(name, bytes, [&](int i) { rowlane_x4<<<(nc + 7) / 8, 64>>>(in[i], out[i], cells, nc); });
        snprintf(name, sizeof name, "D tiled row-lane read, 16B stores cell-tiled   [%s]", tag);
        time(name, bytes, [&](int i) { rowlane_tiled<<<(nc + 7) / 8, 64>>>(in[i], out[i], cells, nc); });
        snprintf(name, sizeof name, "E tiled chunk read,    16B stores cell-tiled   [%s]", tag);
        time(name, bytes, [&](int i) { chunk_tiled<<<(nc + 7) / 8, 64>>>(in[i], out[i], cells, nc); });
    }
    const int waves = (W / 64) * ((H + 47) / 48);
    time("F column walk, row-major read, row-major write", full, [&](int i) { walk_rowmajor<<<(waves + 3) / 4, 256>>>(in[i], out[i]); });
    time("G column walk, cell-tiled read, row-major write", full, [&](int i) { walk_tiled<<<(waves + 3) / 4, 256>>>(in[i], out[i]); });
    return 0;
}
