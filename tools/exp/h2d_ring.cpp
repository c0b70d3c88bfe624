// EXPERIMENT: what do the pieces of a staged host->device upload cost on this box?
//   hipcc -O2 -o tools/exp/bin/h2d_ring tools/exp/h2d_ring.cpp -lpthread
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <thread>
#include <vector>
#include <atomic>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t total = 256u << 20, piece = 2u << 20;
    uint8_t *src = (uint8_t *) malloc(total);
    memset(src, 1, total);
    uint8_t *pin_nc, *pin_c, *dev;
    CK(hipHostMalloc((void **) &pin_nc, 64u << 20, hipHostMallocNonCoherent));
    CK(hipHostMalloc((void **) &pin_c, 64u << 20, hipHostMallocDefault));
    CK(hipMalloc((void **) &dev, 64u << 20));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    // 1. memcpy pageable -> pinned, one thread, fresh source region each time
    for (int kind = 0; kind < 2; kind++) {
        uint8_t *dst = kind ? pin_c : pin_nc;
        double t0 = now();
        for (size_t off = 0; off + piece <= total; off += piece) memcpy(dst + (off % (64u << 20)), src + off, piece);
        double dt = now() - t0;
        printf("memcpy 1 thread -> pinned %s: %.1f GB/s\n", kind ? "coherent" : "non-coherent", total / dt / 1e9);
    }
    // 2. two / four threads
    for (int nt : {2, 4}) {
        double t0 = now();
        std::vector<std::thread> th;
        for (int t = 0; t < nt; t++) th.emplace_back([&, t] {
            for (size_t off = (size_t) t * piece; off + piece <= total; off += (size_t) nt * piece) memcpy(pin_nc + (off % (64u << 20)), src + off, piece);
        });
        for (auto &x : th) x.join();
        double dt = now() - t0;
        printf("memcpy %d threads -> pinned non-coherent: %.1f GB/s (incl. thread start)\n", nt, total / dt / 1e9);
    }
    // 3. DMA pinned -> device, by size
    for (size_t sz : {(size_t) 256 << 10, (size_t) 1 << 20, (size_t) 2 << 20, (size_t) 16 << 20, (size_t) 64 << 20}) {
        CK(hipMemcpyAsync(dev, pin_nc, sz, hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st));
        double t0 = now();
        for (int r = 0; r < 8; r++) CK(hipMemcpyAsync(dev, pin_nc, sz, hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st));
        double dt = (now() - t0) / 8;
        printf("DMA pinned(non-coherent) -> device %8zu B: %.1f us  %.1f GB/s\n", sz, dt * 1e6, sz / dt / 1e9);
    }
    // 4. DMA pageable -> device (runtime path), fresh region each call vs same region
    for (size_t sz : {(size_t) 1 << 20, (size_t) 4 << 20, (size_t) 16 << 20, (size_t) 64 << 20}) {
        double t0 = now();
        size_t n = 0;
        for (size_t off = 0; off + sz <= total; off += sz, n++) { CK(hipMemcpyAsync(dev, src + off, sz, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); }
        double dt = (now() - t0) / n;
        double t1 = now();
        for (int r = 0; r < 8; r++) { CK(hipMemcpyAsync(dev, src, sz, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); }
        double dt2 = (now() - t1) / 8;
        printf("DMA pageable -> device %8zu B: fresh region %.1f us %.1f GB/s   same region %.1f us %.1f GB/s\n", sz, dt * 1e6, sz / dt / 1e9, dt2 * 1e6, sz / dt2 / 1e9);
    }
    // 5. hipHostRegister cost
    for (size_t sz : {(size_t) 1 << 20, (size_t) 16 << 20}) {
        double t0 = now();
        CK(hipHostRegister(src + (128u << 20), sz, hipHostRegisterDefault));
        double t1 = now();
        CK(hipMemcpyAsync(dev, src + (128u << 20), sz, hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st));
        double t2 = now();
        CK(hipHostUnregister(src + (128u << 20)));
        double t3 = now();
        printf("hipHostRegister %8zu B: register %.1f us, copy %.1f us, unregister %.1f us\n", sz, (t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6);
    }
    return 0;
}
