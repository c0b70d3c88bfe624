// EXPERIMENT: how fast does a KERNEL fetch a pinned host buffer (the pull ring of sre_vm_api.cpp)?
// by buffer kind (coherent / non-coherent), launch shape, size, and whether the CPU has just written
// the buffer (regular or streaming stores) or it has long been in memory; and the DMA engine next to it.
//   hipcc -O2 --offload-arch=gfx950 -o tools/exp/bin/pull_rate tools/exp/pull_rate.cpp -lpthread
#include <hip/hip_runtime.h>
#include <emmintrin.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <thread>
#include <vector>
#include <atomic>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ __launch_bounds__(256) void
k_pull(const uint4 *__restrict__ src, uint4 *__restrict__ dst, uint32_t n16)
{
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n16; i += gridDim.x * 256u) dst[i] = src[i];
}
// four loads in flight per lane
__global__ __launch_bounds__(256) void
k_pull4(const uint4 *__restrict__ src, uint4 *__restrict__ dst, uint32_t n16)
{
    const uint32_t stride = gridDim.x * 256u;
    uint32_t       i = blockIdx.x * 256u + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}

static void copy_nt(uint8_t *dst, const uint8_t *src, size_t n)
{
    for (size_t i = 0; i + 64 <= n; i += 64) {
        __m128i a = _mm_loadu_si128((const __m128i *) (src + i)), b = _mm_loadu_si128((const __m128i *) (src + i + 16));
        __m128i c = _mm_loadu_si128((const __m128i *) (src + i + 32)), d = _mm_loadu_si128((const __m128i *) (src + i + 48));
        _mm_stream_si128((__m128i *) (dst + i), a); _mm_stream_si128((__m128i *) (dst + i + 16), b);
        _mm_stream_si128((__m128i *) (dst + i + 32), c); _mm_stream_si128((__m128i *) (dst + i + 48), d);
    }
    _mm_sfence();
}

int main()
{
    const size_t cap = 64u << 20;
    uint8_t *src = (uint8_t *) malloc(256u << 20);
    memset(src, 1, 256u << 20);
    uint8_t *pin[2], *dpin[2], *dev;
    CK(hipHostMalloc((void **) &pin[0], cap, hipHostMallocMapped));                              // coherent
    CK(hipHostMalloc((void **) &pin[1], cap, hipHostMallocMapped | hipHostMallocNonCoherent));
    for (int k = 0; k < 2; k++) CK(hipHostGetDevicePointer((void **) &dpin[k], pin[k], 0));
    CK(hipMalloc((void **) &dev, cap));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    const char *kind_name[2] = {"coherent", "non-coherent"};
    const char *fresh_name[3] = {"long in memory", "just written (memcpy)", "just written (streaming stores)"};
    for (int kind = 0; kind < 2; kind++) {
        for (size_t sz : {(size_t) 256 << 10, (size_t) 1 << 20, (size_t) 4 << 20, (size_t) 16 << 20}) {
            for (int fresh = 0; fresh < 3; fresh++) {
                for (int shape = 0; shape < 5; shape++) {
                    const uint32_t n16 = (uint32_t) (sz / 16);
                    uint32_t       blocks;
                    bool           four = false;
                    switch (shape) {
                    case 0: blocks = (n16 + 1023) / 1024 < 64 ? (n16 + 1023) / 1024 : 64; break;
                    case 1: blocks = (n16 + 1023) / 1024 < 256 ? (n16 + 1023) / 1024 : 256; break;
                    case 2: blocks = (n16 + 255) / 256 < 2048 ? (n16 + 255) / 256 : 2048; break;
                    case 3: blocks = 256; four = true; break;
                    default: blocks = 1024; four = true; break;
                    }
                    if (blocks > (n16 + 255) / 256) blocks = (n16 + 255) / 256;
                    double best = 1e9, sum = 0;
                    const int reps = 6;
                    for (int r = 0; r < reps; r++) {
                        const size_t off = ((size_t) r * sz) % (cap - sz + 1) & ~(size_t) 4095;
                        if (fresh == 1) memcpy(pin[kind] + off, src + (size_t) r * sz, sz);
                        if (fresh == 2) copy_nt(pin[kind] + off, src + (size_t) r * sz, sz);
                        CK(hipStreamSynchronize(st));
                        const double t0 = now();
                        if (four) hipLaunchKernelGGL(k_pull4, dim3(blocks), dim3(256), 0, st, (const uint4 *) (dpin[kind] + off), (uint4 *) dev, n16);
                        else      hipLaunchKernelGGL(k_pull, dim3(blocks), dim3(256), 0, st, (const uint4 *) (dpin[kind] + off), (uint4 *) dev, n16);
                        CK(hipStreamSynchronize(st));
                        const double dt = now() - t0;
                        if (r) { sum += dt; if (dt < best) best = dt; }
                    }
                    printf("pull %-12s %8zu B  %-32s shape %d (%4u blocks%s): avg %.1f us  %.1f GB/s   best %.1f GB/s\n", kind_name[kind], sz,
                           fresh_name[fresh], shape, blocks, four ? " x4" : "", sum / (reps - 1) * 1e6, sz / (sum / (reps - 1)) / 1e9, sz / best / 1e9);
                }
            }
        }
    }
    // the DMA engine from the same buffers, same states
    for (int kind = 0; kind < 2; kind++) {
        for (size_t sz : {(size_t) 1 << 20, (size_t) 4 << 20, (size_t) 16 << 20}) {
            for (int fresh = 0; fresh < 3; fresh++) {
                double sum = 0;
                const int reps = 6;
                for (int r = 0; r < reps; r++) {
                    const size_t off = ((size_t) r * sz) % (cap - sz + 1) & ~(size_t) 4095;
                    if (fresh == 1) memcpy(pin[kind] + off, src + (size_t) r * sz, sz);
                    if (fresh == 2) copy_nt(pin[kind] + off, src + (size_t) r * sz, sz);
                    CK(hipStreamSynchronize(st));
                    const double t0 = now();
                    CK(hipMemcpyAsync(dev, pin[kind] + off, sz, hipMemcpyHostToDevice, st));
                    CK(hipStreamSynchronize(st));
                    if (r) sum += now() - t0;
                }
                printf("DMA  %-12s %8zu B  %-32s: avg %.1f us  %.1f GB/s\n", kind_name[kind], sz, fresh_name[fresh], sum / (reps - 1) * 1e6,
                       sz / (sum / (reps - 1)) / 1e9);
            }
        }
    }
    // pull while three threads keep copying into another part of the same buffer
    for (int kind = 0; kind < 2; kind++) {
        std::atomic<bool> stop{false};
        std::vector<std::thread> th;
        for (int t = 0; t < 3; t++) th.emplace_back([&, t] {
            size_t o = 0;
            while (!stop.load()) { copy_nt(pin[kind] + (32u << 20) + (size_t) t * (4u << 20), src + o, 4u << 20); o = (o + (4u << 20)) % (128u << 20); }
        });
        const size_t   sz = 16u << 20;
        const uint32_t n16 = (uint32_t) (sz / 16);
        double         sum = 0;
        for (int r = 0; r < 6; r++) {
            CK(hipStreamSynchronize(st));
            const double t0 = now();
            hipLaunchKernelGGL(k_pull4, dim3(256), dim3(256), 0, st, (const uint4 *) dpin[kind], (uint4 *) dev, n16);
            CK(hipStreamSynchronize(st));
            if (r) sum += now() - t0;
        }
        printf("pull %-12s 16 MiB with 3 threads copying next to it: %.1f GB/s\n", kind_name[kind], sz / (sum / 5) / 1e9);
        sum = 0;
        for (int r = 0; r < 6; r++) {
            CK(hipStreamSynchronize(st));
            const double t0 = now();
            CK(hipMemcpyAsync(dev, pin[kind], sz, hipMemcpyHostToDevice, st));
            CK(hipStreamSynchronize(st));
            if (r) sum += now() - t0;
        }
        printf("DMA  %-12s 16 MiB with 3 threads copying next to it: %.1f GB/s\n", kind_name[kind], sz / (sum / 5) / 1e9);
        stop.store(true);
        for (auto &x : th) x.join();
    }
    return 0;
}
