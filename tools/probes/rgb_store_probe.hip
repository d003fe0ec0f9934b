// Probe (round 6): what do stage B's picture stores cost by themselves, by shape?  A kernel that does nothing but the stores of
// k_idct_color's pixel phase -- same grid, same workgroups per CU (dynamic LDS), same addresses (2048 pictures of 3840x2160,
// tiles of 32 MCUs, 16 tiles per workgroup) -- in several shapes:
//   0  as built: lane -> (MCU, 4-pixel strip), 8 x global_store_dwordx3 nt per lane and tile (a wave writes 768 contiguous bytes)
//   1  the same, plain stores
//   2  lane -> (MCU, row pair): 6 x global_store_dwordx4 nt per lane and tile, a lane's 16 bytes 48 apart from its neighbour's
//   3  the same, plain stores
//   4  every instruction 1 KiB contiguous (dwordx4 nt), 24 per tile and workgroup: the shape a transposition through LDS would give
//   5  the same, plain stores
//   6  lane -> (MCU, half MCU row = 8 pixels): 2 x dwordx3 per row (24 bytes per lane and row), 512 lanes: scheme with 16 lanes per MCU, nt
//   7  the same, plain
// usage: rgb_store_probe [pictures] [lds_kb]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr uint32_t W = 3840, H = 2160, MCUX = 240, NMCU = 240 * 135, T = 32, NT = (NMCU + T - 1) / T, TPW = 16;
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int V>
__global__ __launch_bounds__(512) void k_store(uint8_t *rgb, uint32_t seed)
{
    extern __shared__ unsigned char smem[];
    const uint32_t tid = threadIdx.x;
    uint8_t *img = rgb + size_t(blockIdx.y) * W * H * 3;
    const uint32_t tile0 = blockIdx.x * TPW, tile1 = min(NT, tile0 + TPW);
    uint32_t a = seed + tid, b = seed * 3 + tid, c = seed * 7 + tid, d = seed ^ tid;
    if (seed == 0xdeadbeef) smem[tid] = 1;
    for (uint32_t tile = tile0; tile < tile1; tile++) {
        const uint32_t m0 = tile * T;
        if (V == 0 || V == 1) {
            const uint32_t q = tid % 128, t = q >> 2, sx = q & 3;
            const uint32_t m = m0 + t;
            if (m >= NMCU) continue;
            const uint32_t mx = m % MCUX, my = m / MCUX;
            uint8_t *col = img + (size_t(my) * 16 * W + mx * 16 + sx * 4) * 3;
#pragma unroll
            for (uint32_t j = 0; j < 4; j++) {
                const uint32_t rp = tid / 128 + 2 * j;
                uint8_t *dst = col + size_t(rp) * 2 * W * 3;
                u32x3 v = {a + j, b + j, c + j};
                if (V == 0) {
                    __builtin_nontemporal_store(v, reinterpret_cast<u32x3 *>(dst));
                    __builtin_nontemporal_store(v, reinterpret_cast<u32x3 *>(dst + W * 3));
                } else {
                    *reinterpret_cast<u32x3 *>(dst) = v;
                    *reinterpret_cast<u32x3 *>(dst + W * 3) = v;
                }
            }
        } else if (V == 2 || V == 3) {
            const uint32_t t = tid & 31, rp = tid >> 5;
            const uint32_t m = m0 + t;
            if (m >= NMCU) continue;
            const uint32_t mx = m % MCUX, my = m / MCUX;
            uint8_t *dst = img + (size_t(my * 16 + rp * 2) * W + mx * 16) * 3;
#pragma unroll
            for (uint32_t r = 0; r < 2; r++)
#pragma unroll
                for (uint32_t k = 0; k < 3; k++) {
                    u32x4 v = {a + k, b + r, c, d};
                    if (V == 2) __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(dst + size_t(r) * W * 3 + k * 16));
                    else *reinterpret_cast<u32x4 *>(dst + size_t(r) * W * 3 + k * 16) = v;
                }
        } else if (V == 4 || V == 5) {
            // the tile's 16 rows x 1536 bytes as 24 KiB; instruction i of the workgroup's wave w covers bytes (6 w + i) * 1024 ...
            const uint32_t mx = m0 % MCUX, my = m0 / MCUX;
            if (mx + T > MCUX) continue;            // (tiles that wrap: skipped, one in 7.5)
            const uint32_t w = tid >> 6, l = tid & 63;
#pragma unroll
            for (uint32_t i = 0; i < 6; i++) {
                const uint32_t off = (6 * w + i) * 1024 + l * 16, row = off / 1536, x = off - row * 1536;
                uint8_t *dst = img + (size_t(my * 16 + row) * W + mx * 16) * 3 + x;
                u32x4 v = {a + i, b, c, d};
                if (V == 4) __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(dst));
                else *reinterpret_cast<u32x4 *>(dst) = v;
            }
        } else if (V == 6 || V == 7) {
            // 512 lanes: lane -> (MCU t, half h, row pair rp)
            const uint32_t h = tid & 1, t = (tid >> 1) & 31, rp = tid >> 6;
            const uint32_t m = m0 + t;
            if (m >= NMCU) continue;
            const uint32_t mx = m % MCUX, my = m / MCUX;
            uint8_t *dst = img + (size_t(my * 16 + rp * 2) * W + mx * 16 + h * 8) * 3;
#pragma unroll
            for (uint32_t r = 0; r < 2; r++)
#pragma unroll
                for (uint32_t k = 0; k < 2; k++) {
                    u32x3 v = {a + k, b + r, c};
                    if (V == 6) __builtin_nontemporal_store(v, reinterpret_cast<u32x3 *>(dst + size_t(r) * W * 3 + k * 12));
                    else *reinterpret_cast<u32x3 *>(dst + size_t(r) * W * 3 + k * 12) = v;
                }
        }
    }
}

template <int V>
static float run(uint8_t *rgb, uint32_t nimg, size_t lds, uint32_t lanes)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_store<V>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
    dim3 grid((NT + TPW - 1) / TPW, nimg);
    hipLaunchKernelGGL(k_store<V>, grid, dim3(lanes), lds, 0, rgb, 1u);
    hipEventRecord(e0);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k_store<V>, grid, dim3(lanes), lds, 0, rgb, 2u + i);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 3;
}

int main(int argc, char **argv)
{
    const uint32_t nimg = argc > 1 ? atoi(argv[1]) : 2048;
    const size_t lds = size_t(argc > 2 ? atoi(argv[2]) : 52) * 1024;
    uint8_t *rgb;
    const size_t bytes = size_t(nimg) * W * H * 3;
    if (hipMalloc(&rgb, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(rgb, 0, bytes);
    const double gb = double(bytes) / 1e9;
    const char *names[8] = {"dwordx3 nt (as built)", "dwordx3 plain", "dwordx4 nt, 48-byte stride", "dwordx4 plain, 48-byte stride",
                            "dwordx4 nt, 1 KiB contiguous", "dwordx4 plain, 1 KiB contiguous", "2 x dwordx3 nt per 24 bytes, 512 lanes", "the same, plain"};
    float ms[8];
    ms[0] = run<0>(rgb, nimg, lds, 256); ms[1] = run<1>(rgb, nimg, lds, 256);
    ms[2] = run<2>(rgb, nimg, lds, 256); ms[3] = run<3>(rgb, nimg, lds, 256);
    ms[4] = run<4>(rgb, nimg, lds, 256); ms[5] = run<5>(rgb, nimg, lds, 256);
    ms[6] = run<6>(rgb, nimg, lds, 512); ms[7] = run<7>(rgb, nimg, lds, 512);
    for (int v = 0; v < 8; v++) printf("lds %zu KB  %-44s %8.3f ms  %7.1f GB/s\n", lds / 1024, names[v], ms[v], gb / ms[v] * 1e3 * (v == 4 || v == 5 ? 6.5 / 7.5 : 1.0));
    return 0;
}
