// Probe: what do many slowly growing output streams cost on this part?  (The write pass of the entropy stage keeps one open
// stream per lane -- ~1000 per CU -- and adds 32 bytes to each every few microseconds)
// Every lane owns a stream of `bytes_per_lane` bytes, `stride` bytes from its neighbour's, and appends G bytes per round
// (G / 16 back-to-back 16-byte stores), with `spin` dependent multiply-adds between two rounds to set the pace.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/stream_store_probe tools/probes/stream_store_probe.hip && /tmp/stream_store_probe
// Prints, per configuration: time with stores, time with the same loop and the stores going to an L2-resident window, and the
// difference as GB/s of the stores' payload.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int G16, bool SMALL>
__global__ __launch_bounds__(512) void k_streams(uint4 *out, uint32_t stride16, uint32_t rounds, uint32_t spin, float *sink)
{
    const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
    uint4 *p = out + size_t(lane) * stride16;
    float a = float(lane), b = 1.0001f;
    for (uint32_t r = 0; r < rounds; r++) {
        for (uint32_t s = 0; s < spin; s++) a = __builtin_fmaf(a, b, 0.5f);          // (dependent chain: the pace of a decode loop)
        const uint32_t w = __float_as_uint(a);
        uint4 *q = SMALL ? out + ((size_t(lane) * stride16 + size_t(r) * G16) & 0x1ffu) + (blockIdx.x & 1023u) * 512u : p + size_t(r) * G16;
#pragma unroll
        for (int g = 0; g < G16; g++) q[g] = make_uint4(w, r, g, lane);
    }
    if (a == 12345.678f) *sink = a;
}

template <int G16>
static void run(uint4 *buf, float *sink, uint32_t lanes, uint32_t bytes_per_lane, uint32_t stride, uint32_t spin_per_32B)
{
    const uint32_t rounds = bytes_per_lane / (16u * G16), spin = spin_per_32B * G16 / 2u;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms[2] = {0, 0};
    for (int small = 0; small < 2; small++) {
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            if (small) hipLaunchKernelGGL((k_streams<G16, true>), dim3(lanes / 512), dim3(512), 0, 0, buf, stride / 16, rounds, spin, sink);
            else hipLaunchKernelGGL((k_streams<G16, false>), dim3(lanes / 512), dim3(512), 0, 0, buf, stride / 16, rounds, spin, sink);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms[small], e0, e1);
        }
    }
    const double gb = double(lanes) * bytes_per_lane / 1e9;
    printf("group %3d B  lanes %7u  stride %5u  spin/32B %4u : to HBM %7.3f ms  to L2 window %7.3f ms  difference %7.3f ms  (%.2f GB payload -> %.0f GB/s over the whole kernel)\n",
           16 * G16, lanes, stride, spin_per_32B, ms[0], ms[1], ms[0] - ms[1], gb, gb / (ms[0] * 1e-3));
}

int main(int argc, char **argv)
{
    const uint32_t bytes_per_lane = 3072, stride = 3072;
    uint4 *buf;
    float *sink;
    const uint32_t max_lanes = 256u * 2048u;
    if (hipMalloc(&buf, size_t(max_lanes) * stride + (1u << 24)) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("no device memory\n"); return 1; }
    hipMemset(buf, 0, size_t(max_lanes) * stride);
    for (uint32_t lanes : {256u * 1024u, 256u * 512u, 256u * 2048u})
        for (uint32_t spin : {400u, 100u}) {
            run<1>(buf, sink, lanes, bytes_per_lane, stride, spin);
            run<2>(buf, sink, lanes, bytes_per_lane, stride, spin);
            run<4>(buf, sink, lanes, bytes_per_lane, stride, spin);
            run<8>(buf, sink, lanes, bytes_per_lane, stride, spin);
        }
    return 0;
}
