/*
 * sre_hip_batch.cpp — the additive device-resident batched API (sregex_hip.h).
 *
 * A scanner binds one compiled program, one mode and one engine:
 *   ENGINE_VM    exact bytecode VM kernel, one lane per stream (sre_hip_vm.hip)
 *   ENGINE_SCAN  table-driven segment-parallel scanner        (sre_hip_scan.hip)
 * Stream pointers/lengths are staged to the device per call; results come back
 * as fixed-stride records.  Everything enqueues on the caller's hipStream_t so
 * a driver can bracket the scan with its own events.
 */
#include <sregex_hip.h>
#include "sre_hip_runtime.h"
#include "sre_hip_scan.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct sre_hip_scanner_s {
    sre_program_t     *prog;
    sre_hip_program_s *dp;
    int                mode, engine;
    uint32_t           ovec_slots;      /* 2 * (max_ncaps + 1) */
    /* per-call staging, grown on demand */
    size_t             cap_streams;
    const void       **d_ptrs;
    uint64_t          *d_lens;
    int64_t           *d_records;
    void              *d_ctx;           /* ENGINE_VM: per-stream VM state */
    size_t             ctx_stride, ctx_cap;
    uint64_t          *h_lens;          /* pinned staging */
    const void       **h_ptrs;
    size_t             last_n;
    hipStream_t        last_stream;
};

static void
scanner_release(void *data)
{
    sre_hip_scanner_t *sc = static_cast<sre_hip_scanner_t *>(data);
    if (sc->d_ptrs) (void) hipFree(sc->d_ptrs);
    if (sc->d_lens) (void) hipFree(sc->d_lens);
    if (sc->d_records) (void) hipFree(sc->d_records);
    if (sc->d_ctx) (void) hipFree(sc->d_ctx);
    if (sc->h_lens) (void) hipHostFree(sc->h_lens);
    if (sc->h_ptrs) (void) hipHostFree(sc->h_ptrs);
    free(sc);
}

extern "C" SRE_API int
sre_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" SRE_API int
sre_hip_set_device(int ordinal)
{
    hipError_t e = hipSetDevice(ordinal);
    return e == hipSuccess ? 0 : sre_hip_fail("hipSetDevice", e);
}

extern "C" SRE_API sre_hip_scanner_t *
sre_hip_scanner_create(sre_pool_t *pool, sre_program_t *prog, int mode, int engine)
{
    if (mode < SRE_HIP_THOMPSON || mode > SRE_HIP_PIKE_COUNT) return NULL;
    sre_hip_program_s *dp = sre_hip_program_get(prog);
    if (dp == NULL) return NULL;

    sre_hip_scanner_t *sc = static_cast<sre_hip_scanner_t *>(calloc(1, sizeof(*sc)));
    if (sc == NULL) return NULL;
    sc->prog = prog;
    sc->dp = dp;
    sc->mode = mode;
    uint32_t maxcaps = 0;
    for (uint32_t i = 0; i < prog->nregexes; i++) {
        if (prog->multi_ncaps[i] > maxcaps) maxcaps = prog->multi_ncaps[i];
    }
    sc->ovec_slots = 2 * (maxcaps + 1);

    if (engine == SRE_HIP_ENGINE_SCAN) {
        fprintf(stderr, "[sregex-hip] the table-driven scanner is not available for this program\n");
        free(sc);
        return NULL;
    }
    sc->engine = SRE_HIP_ENGINE_VM;
    sc->ctx_stride = mode == SRE_HIP_THOMPSON ? dp->thompson_layout.total : dp->pike_layout.total;

    if (sre_pool_add_cleanup(pool, scanner_release, sc) != SRE_OK) {
        free(sc);
        return NULL;
    }
    return sc;
}

extern "C" SRE_API int
sre_hip_scanner_engine(sre_hip_scanner_t *sc)
{
    return sc->engine;
}

extern "C" SRE_API size_t
sre_hip_scanner_result_slots(sre_hip_scanner_t *sc)
{
    return 2 + (size_t) sc->ovec_slots;
}

static int
scanner_reserve(sre_hip_scanner_t *sc, size_t n)
{
    if (n > sc->cap_streams) {
        if (sc->d_ptrs) (void) hipFree(sc->d_ptrs);
        if (sc->d_lens) (void) hipFree(sc->d_lens);
        if (sc->d_records) (void) hipFree(sc->d_records);
        if (sc->h_lens) (void) hipHostFree(sc->h_lens);
        if (sc->h_ptrs) (void) hipHostFree(sc->h_ptrs);
        sc->d_ptrs = NULL;
        sc->d_lens = NULL;
        sc->d_records = NULL;
        sc->h_lens = NULL;
        sc->h_ptrs = NULL;
        sc->cap_streams = 0;
        SRE_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&sc->d_ptrs), n * sizeof(void *)));
        SRE_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&sc->d_lens), n * sizeof(uint64_t)));
        SRE_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&sc->d_records),
                              n * (2 + (size_t) sc->ovec_slots) * sizeof(int64_t)));
        SRE_HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&sc->h_lens), n * sizeof(uint64_t), 0));
        SRE_HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&sc->h_ptrs), n * sizeof(void *), 0));
        sc->cap_streams = n;
    }
    if (sc->engine == SRE_HIP_ENGINE_VM && n * sc->ctx_stride > sc->ctx_cap) {
        if (sc->d_ctx) (void) hipFree(sc->d_ctx);
        sc->d_ctx = NULL;
        sc->ctx_cap = 0;
        SRE_HIP_TRY(hipMalloc(&sc->d_ctx, n * sc->ctx_stride));
        sc->ctx_cap = n * sc->ctx_stride;
    }
    return 0;
hip_failed:
    return -1;
}

extern "C" SRE_API int
sre_hip_scan_enqueue(sre_hip_scanner_t *sc, const void *const *d_streams, const size_t *lens,
    size_t nstreams, void *hip_stream)
{
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    if (nstreams == 0) {
        sc->last_n = 0;
        return 0;
    }
    if (scanner_reserve(sc, nstreams) != 0) return -1;
    for (size_t i = 0; i < nstreams; i++) {
        sc->h_ptrs[i] = d_streams[i];
        sc->h_lens[i] = lens[i];
    }
    SRE_HIP_TRY(hipMemcpyAsync(sc->d_ptrs, sc->h_ptrs, nstreams * sizeof(void *),
                               hipMemcpyHostToDevice, stream));
    SRE_HIP_TRY(hipMemcpyAsync(sc->d_lens, sc->h_lens, nstreams * sizeof(uint64_t),
                               hipMemcpyHostToDevice, stream));
    if (sc->engine == SRE_HIP_ENGINE_VM) {
        /* zero-filled state == fresh context */
        SRE_HIP_TRY(hipMemsetAsync(sc->d_ctx, 0, nstreams * sc->ctx_stride, stream));
        SRE_HIP_TRY(sre_launch_vm_scan(sc->dp->d_blob, sc->mode, sc->d_ptrs, sc->d_lens,
                                       (uint32_t) nstreams, sc->d_ctx, sc->ctx_stride,
                                       sc->d_records, sc->ovec_slots, stream));
    }
    sc->last_n = nstreams;
    sc->last_stream = stream;
    return 0;
hip_failed:
    return -1;
}

extern "C" SRE_API int
sre_hip_scan_results(sre_hip_scanner_t *sc, sre_int_t *results)
{
    if (sc->last_n == 0) return 0;
    size_t bytes = sc->last_n * (2 + (size_t) sc->ovec_slots) * sizeof(int64_t);
    SRE_HIP_TRY(hipMemcpyAsync(results, sc->d_records, bytes, hipMemcpyDeviceToHost,
                               sc->last_stream));
    SRE_HIP_TRY(hipStreamSynchronize(sc->last_stream));
    return 0;
hip_failed:
    return -1;
}

extern "C" SRE_API int
sre_hip_scan_batch(sre_hip_scanner_t *sc, const void *const *d_streams, const size_t *lens,
    size_t nstreams, sre_int_t *results, void *hip_stream)
{
    if (sre_hip_scan_enqueue(sc, d_streams, lens, nstreams, hip_stream) != 0) return -1;
    return sre_hip_scan_results(sc, results);
}

/* ------------------------------------------------------------------ helpers */

extern "C" SRE_API void *
sre_hip_alloc(size_t bytes)
{
    void *p = NULL;
    if (sre_hip_ready() != 0) return NULL;
    hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
    if (e != hipSuccess) {
        sre_hip_fail("hipMalloc", e);
        return NULL;
    }
    return p;
}

extern "C" SRE_API void
sre_hip_free(void *d_ptr)
{
    if (d_ptr) (void) hipFree(d_ptr);
}

extern "C" SRE_API int
sre_hip_upload(void *d_dst, const void *h_src, size_t bytes)
{
    hipError_t e = hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice);
    return e == hipSuccess ? 0 : sre_hip_fail("hipMemcpy H2D", e);
}

extern "C" SRE_API int
sre_hip_download(void *h_dst, const void *d_src, size_t bytes)
{
    hipError_t e = hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost);
    return e == hipSuccess ? 0 : sre_hip_fail("hipMemcpy D2H", e);
}

extern "C" SRE_API int
sre_hip_synchronize(void *hip_stream)
{
    hipError_t e = hipStreamSynchronize(static_cast<hipStream_t>(hip_stream));
    return e == hipSuccess ? 0 : sre_hip_fail("hipStreamSynchronize", e);
}

extern "C" SRE_API int
sre_hip_gen_data(void *d_dst, size_t n, const void *h_tail, size_t tail_len, void *hip_stream)
{
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    void       *d_tail = NULL;
    int         rc = -1;
    if (sre_hip_ready() != 0 || tail_len > n) return -1;
    SRE_HIP_TRY(hipMalloc(&d_tail, tail_len ? tail_len : 1));
    if (tail_len) {
        SRE_HIP_TRY(hipMemcpyAsync(d_tail, h_tail, tail_len, hipMemcpyHostToDevice, stream));
    }
    SRE_HIP_TRY(sre_launch_gen_data(d_dst, n, tail_len, d_tail, stream));
    SRE_HIP_TRY(hipStreamSynchronize(stream));
    rc = 0;
hip_failed:
    if (d_tail) (void) hipFree(d_tail);
    return rc;
}

extern "C" SRE_API int
sre_hip_read_ceiling(const void *d_src, size_t n, void *hip_stream)
{
    static uint32_t *d_sink = NULL;
    if (sre_hip_ready() != 0) return -1;
    if (d_sink == NULL) {
        SRE_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d_sink), SRE_CEILING_GRID * sizeof(uint32_t)));
    }
    SRE_HIP_TRY(sre_launch_read_ceiling(d_src, n, d_sink, static_cast<hipStream_t>(hip_stream)));
    return 0;
hip_failed:
    return -1;
}
