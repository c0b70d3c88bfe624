/*
 * sre_vm_api.cpp — the reference's executor entry points over the HIP kernels.
 *
 *   sre_vm_pike_create_ctx / sre_vm_pike_exec          reference sregex.h:130-134
 *   sre_vm_thompson_create_ctx / sre_vm_thompson_exec  reference sregex.h:144-148
 *   sre_vm_thompson_jit_*                              reference sregex.h:162-171
 *
 * A context is one stream: its VM state stays resident in HBM between exec()
 * calls, each call runs one kernel over the chunk and is synchronous on return
 * (rc, ovector and *pending_matched are valid host memory), exactly the
 * reference's contract.  Chunks of <= 8 bytes ride in the kernel argument,
 * larger ones are staged with one H2D copy.  Requests and results live in one
 * pinned, device-mapped host block per context, so an exec() is: fill request,
 * one launch, one stream sync.
 *
 * There is no CPU matcher behind these calls: without a HIP device exec()
 * returns SRE_ERROR after a diagnostic on stderr.
 */
#include "sre_hip_runtime.h"
#include "sre_hip_scan.h"
#include <sregex_hip.h>
#include <stddef.h>
#include <stdio.h>
#include <atomic>
#if defined(__x86_64__)
#include <emmintrin.h>
#endif
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>
#include <stdlib.h>
#include <string.h>

/* sre_hip_batch.cpp */
extern "C" int sre_hip_scan_one(sre_hip_scanner_t *sc, const void *d_buf, size_t len,
    int init_variant, sre_int_t *rec, int *poisoned, hipStream_t stream);
extern "C" int sre_hip_scan_stream_chunk(sre_hip_scanner_t *sc, const void *d_buf, size_t len,
    int init_variant, int continues, uint32_t entry_state, int eof, int64_t base, sre_stream_ctx_t *d_ctx,
    sre_stream_result_t *d_res, const sre_stream_result_t *h_res, uint32_t ovec_slots, hipStream_t stream,
    void (*midway)(void *), void *midway_arg);
extern "C" int sre_hip_scanner_streams(sre_hip_scanner_t *sc);
extern "C" sre_hip_scanner_t *sre_hip_scanner_create_chunked(sre_pool_t *pool, sre_program_t *prog, int mode);
extern "C" uint32_t sre_hip_scanner_chunk_entry(sre_hip_scanner_t *sc, uint32_t state, int flags);

/* whole-buffer calls at least this long go through a throughput engine (the
 * table-driven scanner, else the NFA tier) when the program admits one: below it
 * the exact VM's ~1 us per byte is cheaper than the scanners' fixed launch cost
 * (profiles/r02_crossover.json) */
#define SRE_COMPAT_SCAN_MIN_BYTES  16u
/* ... and up to this size as ONE chunk of a stream (the chain check by the tail's own workgroup) */
#define SRE_COMPAT_ONE_CHUNK_MAX   (1u << 20)
/* a chunked stream starts on the scanner when its first chunk is at least this long: a chunk
 * costs the scanner ~60-110 us whatever its size, the exact VM kernel ~1 us per byte — and a
 * stream that starts on the VM stays there.  (Byte-at-a-time feeding stays on the VM; the
 * chunk-boundary model itself is exact down to one byte per call, tests/test_dfa_model.py.) */
#define SRE_COMPAT_STREAM_MIN_BYTES 256u

namespace {

#define SRE_EXEC_PENDING INT64_MIN    /* res.rc until the kernel has published its result */

/* pinned host block: request, result header, ovector */
struct HostBlock {
    sre_dev_req_t    req;
    sre_dev_result_t res;
    int64_t          ov[1];   /* ovec_slots */
};

struct DeviceStream {
    sre_hip_program_s *dp;
    int                device;
    void              *d_ctx;       /* VM state, zero-filled == fresh */
    size_t             ctx_bytes, ctx_cap, blk_cap;
    void              *d_in;        /* staging for chunks > 8 bytes */
    size_t             in_cap;
    HostBlock         *h_blk;       /* pinned + mapped */
    HostBlock         *d_blk;       /* device alias of h_blk */
    hipStream_t        stream;
    int                failed;
    int                ctx_fresh;   /* nothing has run on d_ctx since it was opened / handed back: the next request says so
                                     * (sre_dev_req_t.fresh) instead of a fill of the context in front of it */
    size_t             small_off;   /* offset of the small-input area (SRE_SMALL_INPUT bytes) inside the pinned block */
    /* context state accumulated while its searches ran on the scanner */
    uint32_t           preset_valid, preset_flags;
    int64_t            preset_processed;
    /* chunked streaming on the scanner: the carried list (device) and the result block
     * (pinned, device-visible), allocated on first use */
    sre_stream_ctx_t    *d_sctx;
    sre_stream_result_t *h_sres, *d_sres;
    /* pinned double buffer of stage_input */
    uint8_t            *h_stage[2];
    uint8_t            *d_stage;            /* device view of h_stage[0] */
    hipEvent_t          ev_stage[2];
    int                 stage_busy[2];
};

/*
 * Contexts come and go with the caller's pools (the reference's clients make one
 * per subject, src/sre_cli.c:298-660; a filter makes one per request), and a HIP
 * stream + two device allocations + a pinned block cost milliseconds to create.
 * Released device streams are therefore parked in a process-wide free list and
 * handed to the next context that fits (same device, large enough), with the VM
 * state zero-filled again.  The list is bounded, and so is what a parked stream may
 * keep: a staging buffer above SRE_PARK_KEEP_BYTES is freed when its stream is parked
 * (a client that once fed multi-GiB buffers through short-lived contexts would pin that
 * much HBM for the life of the process), and sre_hip_compat_trim() empties the list.
 * The reference has no globals — separate programs and pools may be used from different
 * threads — so the process-wide state of this file is guarded: g_mutex for
 * the parked list and for the creation of a program's scanners, atomics for the counters.
 * (One PROGRAM is not re-entrant, as in the reference, whose instructions carry the VM's
 * generation tags: its contexts share the program's scanners here.)
 */
#define SRE_STREAM_CACHE_MAX 32
#define SRE_PARK_KEEP_BYTES  (24u << 20)
std::mutex    g_mutex;
DeviceStream *g_parked[SRE_STREAM_CACHE_MAX];
int           g_nparked = 0;

void
device_stream_destroy(DeviceStream *ds)
{
    if (ds->d_ctx) (void) hipFree(ds->d_ctx);
    if (ds->d_in) (void) hipFree(ds->d_in);
    for (int b = 0; b < 2; b++) {
        if (ds->h_stage[b]) (void) hipHostFree(ds->h_stage[b]);
        if (ds->ev_stage[b]) (void) hipEventDestroy(ds->ev_stage[b]);
    }
    if (ds->d_sctx) (void) hipFree(ds->d_sctx);
    if (ds->h_sres) (void) hipHostFree(ds->h_sres);
    if (ds->h_blk) (void) hipHostFree(ds->h_blk);
    if (ds->stream) (void) hipStreamDestroy(ds->stream);
    free(ds);
}

void
device_stream_release(void *data)
{
    DeviceStream *ds = static_cast<DeviceStream *>(data);
    if (!ds->failed) {
        if (ds->in_cap > SRE_PARK_KEEP_BYTES) {
            /* (the stream has no work in flight: every exec is synchronous on return) */
            (void) hipFree(ds->d_in);
            ds->d_in = NULL;
            ds->in_cap = 0;
        }
        std::lock_guard<std::mutex> lock(g_mutex);
        if (g_nparked < SRE_STREAM_CACHE_MAX) {
            ds->dp = NULL;                  /* the program may be gone before the next use */
            g_parked[g_nparked++] = ds;
            return;
        }
    }
    device_stream_destroy(ds);
}

DeviceStream *
device_stream_open(sre_pool_t *pool, sre_program_t *prog, size_t ctx_bytes, size_t ovec_slots)
{
    sre_hip_program_s *dp = sre_hip_program_get(prog);
    if (dp == NULL) return NULL;
    /* [request][result][ovector][a chunk of up to SRE_SMALL_INPUT bytes] */
    const size_t small_off = (sizeof(HostBlock) + ovec_slots * sizeof(int64_t) + 15) & ~(size_t) 15;
    const size_t blk = small_off + SRE_SMALL_INPUT;

    DeviceStream *ds = NULL;
    {
        std::lock_guard<std::mutex> lock(g_mutex);
        for (int i = 0; i < g_nparked; i++) {
            DeviceStream *c = g_parked[i];
            if (c->device == dp->device && c->ctx_cap >= ctx_bytes && c->blk_cap >= blk) {
                ds = c;
                g_parked[i] = g_parked[--g_nparked];
                break;
            }
        }
    }
    if (ds != NULL) {
        ds->dp = dp;
        ds->ctx_bytes = ctx_bytes;
        ds->failed = 0;
        ds->preset_valid = ds->preset_flags = 0;
        ds->preset_processed = 0;
        ds->ctx_fresh = 1;
        ds->small_off = small_off;
        memset(ds->h_blk, 0, small_off);
        if (sre_pool_add_cleanup(pool, device_stream_release, ds) != SRE_OK) goto hip_failed;
        return ds;
    }

    ds = static_cast<DeviceStream *>(calloc(1, sizeof(DeviceStream)));
    if (ds == NULL) return NULL;
    ds->dp = dp;
    ds->device = dp->device;
    ds->ctx_bytes = ctx_bytes;
    /* room to be reused by the next program too */
    ds->ctx_cap = ctx_bytes < 64 * 1024 ? 64 * 1024 : ctx_bytes;
    ds->blk_cap = blk < 4096 ? 4096 : blk;
    ds->ctx_fresh = 1;
    ds->small_off = small_off;

    SRE_HIP_TRY(hipStreamCreateWithFlags(&ds->stream, hipStreamNonBlocking));
    SRE_HIP_TRY(hipMalloc(&ds->d_ctx, ds->ctx_cap));
    SRE_HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&ds->h_blk), ds->blk_cap, hipHostMallocMapped));
    SRE_HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&ds->d_blk), ds->h_blk, 0));
    memset(ds->h_blk, 0, ds->blk_cap);
    if (sre_pool_add_cleanup(pool, device_stream_release, ds) != SRE_OK) goto hip_failed;
    return ds;

hip_failed:
    device_stream_destroy(ds);
    return NULL;
}

/* run one chunk through `launch`; returns 0 when h_blk->res is valid */
int
device_stream_exec(DeviceStream *ds, const sre_char *input, size_t len, unsigned eof,
                   unsigned want_pending, size_t ovec_slots,
                   hipError_t (*launch)(const void *, size_t, const sre_dev_req_t *, uint32_t, size_t, hipStream_t))
{
    sre_dev_req_t *rq = &ds->h_blk->req;

    rq->size = len;
    rq->eof = eof ? 1u : 0u;
    rq->want_pending = want_pending;
    rq->ctx = ds->d_ctx;
    rq->result = &ds->d_blk->res;
    rq->ovec_slots = ovec_slots;
    rq->input = NULL;
    rq->inline_bytes = 0;
    rq->preset_valid = ds->preset_valid;
    rq->preset_processed = ds->preset_processed;
    rq->preset_flags = ds->preset_flags;
    rq->fresh = ds->ctx_fresh ? 1u : 0u;
    rq->input_pinned = 0;
    ds->ctx_fresh = 0;
    if (len > 0 && len <= 8) {
        memcpy(&rq->inline_bytes, input, len);
    } else if (len <= SRE_SMALL_INPUT && len > 8) {
        /* next to the request, in the pinned block: the kernel stages it into LDS itself (a copy
         * of its own costs more than the whole call should) */
        memcpy(reinterpret_cast<uint8_t *>(ds->h_blk) + ds->small_off, input, len);
        rq->input = reinterpret_cast<const uint8_t *>(ds->d_blk) + ds->small_off;
        rq->input_pinned = 1;
    } else if (len > 8) {
        if (len > ds->in_cap) {
            if (ds->d_in) (void) hipFree(ds->d_in);
            ds->d_in = NULL;
            ds->in_cap = 0;
            size_t cap = len + (len >> 2) + 4096;
            SRE_HIP_TRY(hipMalloc(&ds->d_in, cap));
            ds->in_cap = cap;
        }
        SRE_HIP_TRY(hipMemcpyAsync(ds->d_in, input, len, hipMemcpyHostToDevice, ds->stream));
        rq->input = static_cast<const uint8_t *>(ds->d_in);
    }
    *const_cast<volatile int64_t *>(&ds->h_blk->res.rc) = SRE_EXEC_PENDING;

    if (launch == sre_launch_pike_exec && ds->dp->d_pwave != NULL) {
        /* the exact step by a wavefront (sre_hip_pwave.hip) */
        SRE_HIP_TRY(sre_launch_pike_exec_wave(ds->dp->d_pwave, ds->dp->h_pwave, &ds->d_blk->req, 1, ds->stream));
    } else {
        SRE_HIP_TRY(launch(ds->dp->d_blob, ds->dp->blob_bytes, &ds->d_blk->req, 1, ds->ctx_bytes, ds->stream));
    }
    {
        /* the kernel writes res->rc last, behind a system-scope fence: watching that word costs
         * a few us less than the runtime's wait (an interrupt and a wake-up: 63-byte calls took
         * 59 us, profiles/r02_crossover.json) — for as long as a short call can take, then the
         * ordinary wait.  (What the kernel still does behind that store — the context's way back
         * from LDS — is in front of the next call's work on the same queue.) */
        const volatile int64_t *prc = &ds->h_blk->res.rc;
        const auto              t0 = std::chrono::steady_clock::now();
        uint32_t                polls = 0;
        while (*prc == SRE_EXEC_PENDING) {
            if ((++polls & 63u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(500)) {
                SRE_HIP_TRY(hipStreamSynchronize(ds->stream));
                break;
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        if (*prc == SRE_EXEC_PENDING) {
            /* the kernel never published: a device fault */
            ds->h_blk->res.rc = SRE_ERROR;
            goto hip_failed;
        }
    }
    return 0;

hip_failed:
    ds->failed = 1;
    return -1;
}

}  // namespace

/* ------------------------------------------------------------------ Pike */

struct sre_vm_pike_ctx_s {
    sre_pool_t    *pool;
    sre_program_t *prog;
    sre_int_t     *ovector;         /* caller-owned (reference sre_vm_pike.c:131-132) */
    size_t         ovec_slots;
    sre_int_t      pending[2];
    DeviceStream  *ds;
    /* host shadow of the context between searches (sre_vm_pike.c:47-76): valid
     * while the device context has not been touched by the VM kernel */
    int            at_boundary;     /* no search in flight on the device context */
    int            vm_touched;      /* the VM kernel owns the state from now on */
    int            eof, empty_capture, seen_newline, seen_word;
    sre_int_t      processed_bytes;
    sre_hip_scanner_t *scanner;     /* lazily created, NULL if the program is not admitted */
    int            scanner_tried;
    sre_int_t     *rec;
    /* a search that runs chunk by chunk on the scanner (pike_stream_route) */
    sre_hip_scanner_t *stream_scanner;  /* look-ahead programs: the chunked automaton's scanner */
    int            stream_scanner_tried;
    int            stream_mode;     /* the device holds the carried list of a search under way */
    uint32_t       stream_state;    /* host shadow of the automaton state in front of the next byte */
};

/*
 * Host -> device copy of a chunk from the caller's (pageable) buffer, by size — measured
 * with tools/h2d_sizes.py (profiles/r02_h2d_sizes.txt), microseconds per copy incl. the wait:
 *            hipMemcpyAsync from pageable    memcpy to pinned + DMA    DMA from pinned alone
 *   64 KiB              13                          13                        12
 *    1 MiB              66                          62                        29
 *    4 MiB              94                         219                        85
 *   16 MiB             307  (55 GB/s)              823  (20 GB/s)            303
 * From 2 MiB on the runtime's own path (it pins the caller's pages for the transfer) runs at
 * DMA speed and the CPU copy into a staging buffer (30 GB/s) is the slower way; below, the
 * two are equal and the pinned buffer saves the runtime's bookkeeping.  (The pinned buffer
 * must be hipHostMallocNonCoherent: DMA out of the default coherent kind runs at 1.5 GB/s.)
 */
/* copy of a pinned host buffer (mapped into the device's address space) to device memory;
 * 16 bytes per lane and access, whole buffer rounded up to 16 (both are allocated larger) */
__global__ __launch_bounds__(256) void
sre_k_pull(const uint4 *__restrict__ src, uint4 *__restrict__ dst, uint32_t n16)
{
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n16; i += gridDim.x * 256u) dst[i] = src[i];
}

/*
 * The fetch of a chunk that is still being copied into the pinned ring (ring_upload): ONE launch
 * per chunk, in flight while the CPUs copy.  The grid is SRE_FETCH_GROUPS groups of
 * SRE_FETCH_GROUP_WGS workgroups; group g takes units g, g + GROUPS, ...: in front of a unit lane
 * 0 of each of its workgroups waits for the unit's flag in host memory (the CPU publishes it behind
 * the unit's last store), then the group copies the unit, four 16-byte loads in flight per lane.
 * The groups wait for DIFFERENT units at the same time, so a flag's round trip over the link is
 * paid once per GROUPS units, not once per unit (a first version walked the units one after the
 * other with the whole grid: 6 us per unit, as slow as a launch per unit).  The last workgroup of
 * a group to finish a unit says so in host memory (`ctl->done[slot]`): that frees the unit's ring
 * slot for chunks longer than the ring.  Every wait is bounded (SRE_RING_WAIT_TICKS of the
 * 100 MHz clock): a workgroup that gives up sets ctl->timed_out and every workgroup leaves at its
 * next wait — the grid always drains, and the host fails the stream.
 */
struct sre_ring_ctl_t {
    uint64_t flag[64];          /* per ring slot: (epoch << 32) | unit, written by the CPU when the unit is in the ring */
    uint64_t done[64];          /* per ring slot: the same tag, written by the GPU when the unit has been fetched */
    uint64_t timed_out;         /* set by the GPU */
    uint64_t chunk_done;        /* epoch of the last chunk the GPU has fetched completely (or given up on) */
};
#define SRE_RING_WAIT_TICKS  200000000ull   /* 2 s */
#define SRE_FETCH_GROUPS     8u
#define SRE_FETCH_GROUP_WGS  8u

__global__ __launch_bounds__(256) void
sre_k_ring_fetch(const uint8_t *__restrict__ ring, uint8_t *__restrict__ dst, sre_ring_ctl_t *ctl, uint32_t *__restrict__ counters,
                 uint64_t len, uint32_t unit, uint32_t nunits, uint32_t ring_units, uint32_t epoch)
{
    __shared__ uint32_t sh_go;
    const uint32_t group = blockIdx.x / SRE_FETCH_GROUP_WGS, member = blockIdx.x % SRE_FETCH_GROUP_WGS;
    const uint32_t stride = SRE_FETCH_GROUP_WGS * 256u;
    /* the last workgroup to leave says that nothing reads the ring any more (another context's
     * chunk may then use it: ring_begin) */
    auto leave = [&]() {
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            if (atomicAdd(&counters[64], 1u) == gridDim.x - 1) {
                counters[64] = 0;
                __hip_atomic_store(&ctl->chunk_done, (uint64_t) epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    };
    for (uint32_t u = group; u < nunits; u += SRE_FETCH_GROUPS) {
        const uint32_t slot = u % ring_units;
        const uint64_t tag = ((uint64_t) epoch << 32) | u;
        if (threadIdx.x == 0) {
            const uint64_t t0 = wall_clock64();
            uint32_t       go = 1;
            while (__hip_atomic_load(&ctl->flag[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != tag) {
                if (__hip_atomic_load(&ctl->timed_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0
                    || wall_clock64() - t0 > SRE_RING_WAIT_TICKS)
                {
                    __hip_atomic_store(&ctl->timed_out, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    go = 0;
                    break;
                }
                __builtin_amdgcn_s_sleep(4);
            }
            sh_go = go;
        }
        __syncthreads();
        if (!sh_go) {
            leave();
            return;
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);        /* nothing of the slot's previous content is kept */
        const uint64_t off = (uint64_t) u * unit;
        const uint32_t n16 = (uint32_t) (((len - off < unit ? len - off : unit) + 15) / 16);
        const uint4   *s = reinterpret_cast<const uint4 *>(ring + (size_t) slot * unit);
        uint4         *d = reinterpret_cast<uint4 *>(dst + off);
        uint32_t       i = member * 256u + threadIdx.x;
        for (; i + 3 * stride < n16; i += 4 * stride) {
            const uint4 a = s[i], b = s[i + stride], c = s[i + 2 * stride], e = s[i + 3 * stride];
            d[i] = a;
            d[i + stride] = b;
            d[i + 2 * stride] = c;
            d[i + 3 * stride] = e;
        }
        for (; i < n16; i += stride) d[i] = s[i];
        if (nunits > ring_units) {
            /* the slot is reused: say when the whole group is through with the unit */
            __syncthreads();
            if (threadIdx.x == 0) {
                __threadfence();
                if (atomicAdd(&counters[slot], 1u) == SRE_FETCH_GROUP_WGS - 1) {
                    counters[slot] = 0;
                    __hip_atomic_store(&ctl->done[slot], tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
        }
    }
    leave();
}

#define SRE_STAGE_PIECE   (2u << 20)
#define SRE_STAGE_MIN     (64u << 10)
#define SRE_RING_BYTES    (16u << 20)       /* the process-wide pinned ring */
#define SRE_RING_MIN      (512u << 10)      /* chunks from here on go through it */
#define SRE_RING_SPIN_US  300               /* a helper polls this long for the next chunk before it sleeps */

/*
 * Larger chunks (round 3).  A chunk in the caller's pageable memory has to be copied once by
 * the CPU (one core: 22-30 GB/s) and fetched once over the link (~54 GB/s); done one after the
 * other, and in front of the scan, that was 9 GB/s for 1 MiB chunks and 22 GB/s for 16 MiB
 * (profiles/r03_stream_rate.json, first build; the runtime's own path for pageable sources is
 * erratic on this pool: 55 GB/s in a loop of its own, 4-27 GB/s inside a stream's calls).  Now
 * the chunk is cut into UNITS (128 KiB .. 1 MiB): sre_k_ring_fetch is launched FIRST, then a
 * small pool of helper threads and the caller copy units side by side into ONE process-wide
 * pinned ring and publish a flag per unit; the kernel fetches every unit as soon as its flag is
 * up — the GPU fetches the head of the chunk while the CPUs copy its tail, one launch per chunk,
 * on the context's own queue in front of the scan.  (A launch per unit: ~6 us each, in a row; the
 * DMA engine: ~10 us per transfer and a queue-to-engine dependency in front of the scan.)  A ring
 * slot is reused (chunks above 16 MiB) when the kernel reports the unit it held as fetched.  The
 * helpers poll for the next chunk for SRE_RING_SPIN_US after one (a stream's calls come
 * back-to-back; waking a sleeping thread costs 30-60 us, a 1 MiB chunk's whole budget), then
 * sleep; SRE_HIP_COPY_THREADS=0 leaves everything to the caller's thread.  One chunk at a time
 * goes through the ring (g_copy_mutex); a chunk of ANOTHER context first waits for the fetch of
 * the previous one (ctl->chunk_done).
 */
namespace {

/* host copy into a pinned buffer the GPU reads next: streaming stores, so that the lines go to
 * memory instead of sitting dirty in this core's cache (the device's reads of freshly written
 * lines were served out of the CPU caches at ~36 GB/s; from memory the link runs at ~50) and
 * are not read first (no read-for-ownership).  dst is 64-byte aligned. */
inline void
copy_streaming(uint8_t *dst, const uint8_t *src, size_t n)
{
#if defined(__x86_64__)
    size_t i = 0;
    for (; i + 64 <= n; i += 64) {
        const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + i));
        const __m128i b = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + i + 16));
        const __m128i c = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + i + 32));
        const __m128i d = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + i + 48));
        _mm_stream_si128(reinterpret_cast<__m128i *>(dst + i), a);
        _mm_stream_si128(reinterpret_cast<__m128i *>(dst + i + 16), b);
        _mm_stream_si128(reinterpret_cast<__m128i *>(dst + i + 32), c);
        _mm_stream_si128(reinterpret_cast<__m128i *>(dst + i + 48), d);
    }
    if (i < n) memcpy(dst + i, src + i, n - i);
    _mm_sfence();
#else
    memcpy(dst, src, n);
#endif
}

struct CopyJob {
    const uint8_t *src;
    size_t         len, unit, nunits, ring_units;
    uint32_t       epoch;
};

struct CopyPool {
    std::mutex               m;
    std::condition_variable  cv_work;
    std::vector<std::thread> threads;
    CopyJob                  job{};                 /* written under m */
    std::atomic<uint64_t>    generation{0};
    std::atomic<int>         parked{0};
    std::atomic<uint64_t>    next{0};               /* (epoch << 32) | next unit to copy */
    bool                     stop = false;
    uint8_t                 *h_ring = nullptr, *d_ring = nullptr;
    sre_ring_ctl_t          *h_ctl = nullptr, *d_ctl = nullptr;     /* pinned: unit flags (CPU -> GPU), progress (GPU -> CPU) */
    uint32_t                *d_counters = nullptr;                  /* per slot: workgroups through with the unit */
    int                      device = -1;
    uint32_t                 last_epoch = 0;        /* the previous chunk's epoch (0: no fetch was launched) */

    /* may unit u of job j be written into its ring slot?  (the unit that held it has been fetched) */
    bool slot_free(const CopyJob &j, size_t u) const
    {
        return u < j.ring_units
               || __atomic_load_n(&h_ctl->done[u % j.ring_units], __ATOMIC_ACQUIRE) == (((uint64_t) j.epoch << 32) | (uint32_t) (u - j.ring_units));
    }

    /* copy units of job `j` until none is left (or `max_units` are done).  helper == false (the
     * caller's thread, which is also the one that frees slots): never take a unit whose slot is
     * still being fetched */
    void run(const CopyJob &j, size_t max_units, bool helper)
    {
        for (size_t done = 0; done < max_units; done++) {
            uint64_t cur = next.load(std::memory_order_acquire);
            for (;;) {
                if ((uint32_t) (cur >> 32) != j.epoch || (uint32_t) cur >= j.nunits) return;
                if (!helper && !slot_free(j, (uint32_t) cur)) return;
                if (next.compare_exchange_weak(cur, cur + 1, std::memory_order_acq_rel)) break;
            }
            const size_t u = (uint32_t) cur;
            while (!slot_free(j, u)) {
                /* the slot is still being fetched */
                if (__atomic_load_n(&h_ctl->timed_out, __ATOMIC_RELAXED)) break;    /* (the chunk is lost anyway) */
#if defined(__x86_64__)
                __builtin_ia32_pause();
#endif
            }
            const size_t off = u * j.unit, n = j.len - off < j.unit ? j.len - off : j.unit;
            copy_streaming(h_ring + (u % j.ring_units) * j.unit, j.src + off, n);
            __atomic_store_n(&h_ctl->flag[u % j.ring_units], ((uint64_t) j.epoch << 32) | (uint32_t) u, __ATOMIC_RELEASE);
        }
    }

    void worker()
    {
        uint64_t seen = 0;
        for (;;) {
            const auto t0 = std::chrono::steady_clock::now();
            uint32_t   polls = 0;
            while (generation.load(std::memory_order_acquire) == seen) {
#if defined(__x86_64__)
                __builtin_ia32_pause();
#endif
                if ((++polls & 255u) == 0
                    && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(SRE_RING_SPIN_US))
                {
                    std::unique_lock<std::mutex> lk(m);
                    parked.fetch_add(1);
                    cv_work.wait(lk, [&] { return stop || generation.load() != seen; });
                    parked.fetch_sub(1);
                    break;
                }
            }
            CopyJob j;
            {
                std::lock_guard<std::mutex> lk(m);
                if (stop) return;
                j = job;
                seen = generation.load();
            }
            run(j, ~(size_t) 0, true);
        }
    }
};

CopyPool  *g_copy_pool;
std::mutex g_copy_mutex;        /* one chunk at a time through the ring */

int
copy_threads()
{
    const char *e = getenv("SRE_HIP_COPY_THREADS");
    /* never more than the machine has to spare (the caller's thread copies too) */
    const int   hw = (int) std::thread::hardware_concurrency();
    const int   n = e ? atoi(e) : (hw >= 8 ? 3 : hw >= 4 ? 1 : 0);
    return n < 0 ? 0 : n > 8 ? 8 : n;
}

/* a chunk on its way through the ring: between ring_begin and ring_finish the caller may queue the
 * kernels that consume ds->d_in behind the fetch (their launch latency passes while the CPUs copy) */
struct RingTicket {
    std::unique_lock<std::mutex> lock;      /* g_copy_mutex */
    CopyJob                      j;
    DeviceStream                *ds = nullptr;
    bool                         active = false;
};

int ring_finish(RingTicket &t);

/* the chunk -> ds->d_in by a kernel queued on ds->stream; the helpers are copying on return.
 * 0: under way, ring_finish() must follow; 1: not taken; -1: failed */
int
ring_begin(DeviceStream *ds, const sre_char *input, size_t len, RingTicket &t)
{
    t.lock = std::unique_lock<std::mutex>(g_copy_mutex);
    if (g_copy_pool == NULL) {
        CopyPool *P = new CopyPool();
        /* read by a kernel while the CPU writes other parts of it: the coherent (fine-grained) kind */
        if (hipHostMalloc(reinterpret_cast<void **>(&P->h_ring), SRE_RING_BYTES, hipHostMallocMapped) != hipSuccess
            || hipHostGetDevicePointer(reinterpret_cast<void **>(&P->d_ring), P->h_ring, 0) != hipSuccess
            || hipHostMalloc(reinterpret_cast<void **>(&P->h_ctl), sizeof(sre_ring_ctl_t), hipHostMallocMapped) != hipSuccess
            || hipHostGetDevicePointer(reinterpret_cast<void **>(&P->d_ctl), P->h_ctl, 0) != hipSuccess
            || hipMalloc(reinterpret_cast<void **>(&P->d_counters), 65 * sizeof(uint32_t)) != hipSuccess
            || hipMemset(P->d_counters, 0, 65 * sizeof(uint32_t)) != hipSuccess)
        {
            t.lock.unlock();
            return -1;      /* (what was allocated stays with the never-used pool object: a dead device) */
        }
        memset(P->h_ctl, 0xff, sizeof(P->h_ctl->flag));
        memset(P->h_ctl->done, 0xff, sizeof(P->h_ctl->done));
        P->h_ctl->timed_out = 0;
        P->h_ctl->chunk_done = 0;
        P->device = ds->device;
        g_copy_pool = P;
        const int nt = copy_threads();
        for (int i = 0; i < nt; i++) P->threads.emplace_back([P] { P->worker(); });
    }
    CopyPool &P = *g_copy_pool;
    if (P.device != ds->device) {                   /* (one process drives one GPU; another device: the plain path) */
        t.lock.unlock();
        return 1;
    }
    /* the previous chunk's fetch reads the ring: a chunk of the same context has seen it finish
     * (exec is synchronous), another context's may still run — its last workgroup says when it is
     * over (no event in the queue: recording one between the fetch and the scan cost 5 us a call) */
    if (P.last_epoch != 0) {
        const auto t0 = std::chrono::steady_clock::now();
        uint32_t   polls = 0;
        while ((uint32_t) __atomic_load_n(&P.h_ctl->chunk_done, __ATOMIC_ACQUIRE) != P.last_epoch) {
            if ((++polls & 1023u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) {
                t.lock.unlock();
                return -1;
            }
        }
    }
    if (__atomic_load_n(&P.h_ctl->timed_out, __ATOMIC_RELAXED)) {      /* a fetch gave up: the ring is not trusted again */
        t.lock.unlock();
        return -1;
    }

    CopyJob j;
    j.src = input;
    j.len = len;
    /* small units for small chunks: the fetch of the first ones starts while the rest is copied */
    j.unit = len <= (2u << 20) ? (128u << 10) : len <= (8u << 20) ? (256u << 10) : (1u << 20);
    j.nunits = (len + j.unit - 1) / j.unit;
    j.ring_units = SRE_RING_BYTES / j.unit < 64 ? SRE_RING_BYTES / j.unit : 64;
    {
        std::lock_guard<std::mutex> lk(P.m);
        j.epoch = P.job.epoch + 1;
        P.job = j;
        P.next.store((uint64_t) j.epoch << 32, std::memory_order_release);
    }
    /* the helpers start copying ... */
    {
        std::lock_guard<std::mutex> lk(P.m);
        P.generation.fetch_add(1, std::memory_order_release);
    }
    if (P.parked.load() > 0) P.cv_work.notify_all();
    /* ... while the fetch is launched (it finds the first flags up, or waits for them) */
    {
        /* few workgroups walking a unit with a grid stride: 54 GB/s; one 16-byte access per lane
         * over as many workgroups as that takes: 37-50 GB/s (tools/exp/pull_rate.cpp) */
        hipLaunchKernelGGL(sre_k_ring_fetch, dim3(SRE_FETCH_GROUPS * SRE_FETCH_GROUP_WGS), dim3(256), 0, ds->stream, P.d_ring,
                           static_cast<uint8_t *>(ds->d_in), P.d_ctl, P.d_counters, (uint64_t) len, (uint32_t) j.unit,
                           (uint32_t) j.nunits, (uint32_t) j.ring_units, j.epoch);
        const bool launched = hipGetLastError() == hipSuccess;
        P.last_epoch = launched ? j.epoch : 0;
        t.j = j;
        t.ds = ds;
        t.active = true;
        if (!launched) {
            /* nothing reads the ring: let the copies run dry (they touch the ring and the caller's
             * chunk only), then give the chunk to the plain path */
            (void) ring_finish(t);
            return 1;
        }
    }
    return 0;
}

/* the caller's share of the copies; returns when every unit is in the ring (the fetch is at most a
 * few units behind) */
int
ring_finish(RingTicket &t)
{
    if (!t.active) return 0;
    CopyPool      &P = *g_copy_pool;
    const CopyJob &j = t.j;
    DeviceStream  *ds = t.ds;
    for (size_t u = 0; u < j.nunits; u++) {
        /* (a slot's tags only grow within a chunk: the flag of unit u may already be that of unit
         * u + ring_units when the fetch and the helpers are ahead of this loop) */
        for (;;) {
            const uint64_t f = __atomic_load_n(&P.h_ctl->flag[u % j.ring_units], __ATOMIC_ACQUIRE);
            if ((uint32_t) (f >> 32) == j.epoch && (uint32_t) f >= u) break;
            P.run(j, 1, false);     /* returns at once when every unit is taken (or none has a free slot) */
            if (__atomic_load_n(&P.h_ctl->timed_out, __ATOMIC_RELAXED)) break;
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        }
    }
    int rc = 0;
    if (__atomic_load_n(&P.h_ctl->timed_out, __ATOMIC_RELAXED)) {
        fprintf(stderr, "[sregex-hip] the fetch of a chunk gave up waiting for the host copy\n");
        (void) hipStreamSynchronize(ds->stream);
        rc = -1;
    }
    t.active = false;
    t.lock.unlock();
    return rc;
}

int
ring_upload(DeviceStream *ds, const sre_char *input, size_t len)
{
    RingTicket t;
    const int  r = ring_begin(ds, input, len, t);
    return r != 0 ? r : ring_finish(t);
}

/* called by sre_hip_scan_stream_chunk between its launches and its wait */
void
ring_midway(void *arg)
{
    RingTicket *t = static_cast<RingTicket *>(arg);
    if (!t->active) return;
    DeviceStream *ds = t->ds;
    if (ring_finish(*t) != 0) t->ds = ds + 1;       /* (marks the failure for the caller) */
}

}  // namespace

/* `ticket`: a chunk that goes through the ring is only STARTED (ring_begin); the caller queues
 * its kernels and calls ring_finish(*ticket) before it waits for them */
static int
stage_input(DeviceStream *ds, const sre_char *input, size_t len, RingTicket *ticket = nullptr)
{
    if (len > ds->in_cap) {
        if (ds->d_in) (void) hipFree(ds->d_in);
        ds->d_in = NULL;
        ds->in_cap = 0;
        if (hipMalloc(&ds->d_in, len + (len >> 2) + 4096) != hipSuccess) return -1;
        ds->in_cap = len + (len >> 2) + 4096;
    }
    if (len == 0) return 0;
    static const bool use_ring = getenv("SRE_HIP_NO_RING") == NULL;
    if (len >= SRE_RING_MIN && use_ring) {
        /* (without helper threads the copies are the caller's alone: they come first — a
         * device-wide wait inside the scan's set-up would otherwise wait for a fetch that waits
         * for the caller) */
        static const bool helpers = copy_threads() > 0;
        const int r = ticket && helpers ? ring_begin(ds, input, len, *ticket) : ring_upload(ds, input, len);
        if (r <= 0) return r;
    }
    if (len > SRE_STAGE_PIECE) {
        return hipMemcpyAsync(ds->d_in, input, len, hipMemcpyHostToDevice, ds->stream) == hipSuccess ? 0 : -1;
    }
    /* one pinned buffer: exec() is synchronous, the copy out of it has finished (its
     * consumer has) before the next call writes into it */
    static const bool pull = getenv("SRE_HIP_NO_PULL") == NULL;
    if (ds->h_stage[0] == NULL) {
        /* read by a kernel: the coherent (fine-grained) kind, which the GPU never caches — the
         * buffer is rewritten by the CPU between kernels; read by the DMA engine: the
         * non-coherent kind (out of the coherent one it runs at 1.5 GB/s) */
        if (hipHostMalloc(reinterpret_cast<void **>(&ds->h_stage[0]), SRE_STAGE_PIECE,
                          pull ? hipHostMallocMapped : hipHostMallocNonCoherent | hipHostMallocMapped) != hipSuccess
            || hipHostGetDevicePointer(reinterpret_cast<void **>(&ds->d_stage), ds->h_stage[0], 0) != hipSuccess)
        {
            return -1;
        }
    }
    if (pull) {
        copy_streaming(ds->h_stage[0], input, len);
    } else {
        memcpy(ds->h_stage[0], input, len);
    }
    if (pull) {
        /* the GPU fetches the buffer itself: a kernel in front of the scan on the same queue
         * instead of a DMA-engine copy the scan has to be synchronised with */
        const uint32_t n16 = (uint32_t) ((len + 15) / 16);
        hipLaunchKernelGGL(sre_k_pull, dim3((n16 + 1023) / 1024 < 64 ? (n16 + 1023) / 1024 : 64), dim3(256), 0,
                           ds->stream, reinterpret_cast<const uint4 *>(ds->d_stage), static_cast<uint4 *>(ds->d_in), n16);
        return hipGetLastError() == hipSuccess ? 0 : -1;
    }
    return hipMemcpyAsync(ds->d_in, ds->h_stage[0], len, hipMemcpyHostToDevice, ds->stream) == hipSuccess ? 0 : -1;
}

/* The throughput scanner of a program for the compat path: built once per program
 * and mode (automaton + device tables cost milliseconds), owned by the PROGRAM's
 * pool, shared by every context of the program — exec() is synchronous, so one call
 * is in flight at a time.  NULL when only the exact VM takes the program. */
static sre_hip_scanner_t *
compat_scanner(sre_program_t *prog, int mode, int chunked = 0)
{
    sre_hip_program_s *dp = sre_hip_program_get(prog);
    if (dp == NULL) return NULL;
    /* chunked: the automaton of a look-ahead program whose stream arrives in chunks (other
     * programs: the ordinary one) */
    const int slot = (chunked ? 2 : 0) + (mode == SRE_HIP_THOMPSON ? 0 : 1);
    std::lock_guard<std::mutex> lock(g_mutex);
    if (!dp->compat_tried[slot]) {
        dp->compat_tried[slot] = 1;
        sre_hip_scanner_t *sc = chunked ? sre_hip_scanner_create_chunked(prog->pool, prog, mode)
                                        : sre_hip_scanner_create(prog->pool, prog, mode, SRE_HIP_ENGINE_AUTO);
        if (sc && sre_hip_scanner_engine(sc) == SRE_HIP_ENGINE_VM) sc = NULL;
        dp->compat_scanner[slot] = sc;
    }
    return dp->compat_scanner[slot];
}

static std::atomic<unsigned long long> g_route_counts[3];

extern "C" SRE_API void
sre_hip_compat_route_counts(unsigned long long out[3])
{
    for (int i = 0; i < 3; i++) out[i] = g_route_counts[i].load();
}

/* free every parked device stream (HIP stream, context, staging buffers); returns how many */
extern "C" SRE_API int
sre_hip_compat_trim(void)
{
    DeviceStream *gone[SRE_STREAM_CACHE_MAX];
    int           n = 0;
    {
        std::lock_guard<std::mutex> lock(g_mutex);
        for (int i = 0; i < g_nparked; i++) gone[n++] = g_parked[i];
        g_nparked = 0;
    }
    for (int i = 0; i < n; i++) device_stream_destroy(gone[i]);
    /* ... and the pinned ring with its helper threads (started again by the next large chunk) */
    {
        std::lock_guard<std::mutex> job_lock(g_copy_mutex);
        CopyPool *P = g_copy_pool;
        if (P != NULL) {
            /* the last chunk's fetch may still read the ring */
            if (P->last_epoch != 0) {
                const auto t0 = std::chrono::steady_clock::now();
                while ((uint32_t) __atomic_load_n(&P->h_ctl->chunk_done, __ATOMIC_ACQUIRE) != P->last_epoch
                       && std::chrono::steady_clock::now() - t0 < std::chrono::seconds(5))
                {
                }
            }
            {
                std::lock_guard<std::mutex> lk(P->m);
                P->stop = true;
                P->generation.fetch_add(1, std::memory_order_release);
            }
            P->cv_work.notify_all();
            for (auto &t : P->threads) t.join();
            (void) hipHostFree(P->h_ring);
            (void) hipHostFree(P->h_ctl);
            (void) hipFree(P->d_counters);
            delete P;
            g_copy_pool = NULL;
        }
    }
    return n;
}

static int
pike_scan_route(sre_vm_pike_ctx_t *ctx, sre_char *input, size_t len, sre_int_t *prc)
{
    /* one reference exec() of a context that sits between two searches, on a
     * whole buffer (eof): sre_vm_pike.c:165-233 prologue, scan, :586-636 epilogue */
    if (ctx->eof) {
        *prc = SRE_ERROR;                                   /* :165-168 */
        return 1;
    }
    if (!ctx->scanner_tried) {
        ctx->scanner_tried = 1;
        ctx->scanner = compat_scanner(ctx->prog, SRE_HIP_PIKE_FIRST);
        if (ctx->scanner) {
            ctx->rec = static_cast<sre_int_t *>(
                sre_palloc(ctx->pool, sre_hip_scanner_result_slots(ctx->scanner) * sizeof(sre_int_t)));
            if (ctx->rec == NULL) ctx->scanner = NULL;
        }
    }
    if (ctx->scanner == NULL) return 0;
    /* the NFA tier's exact window runs a FRESH context from a clean position */
    if (sre_hip_scanner_engine(ctx->scanner) == SRE_HIP_ENGINE_NFA
        && (ctx->empty_capture || ctx->processed_bytes != 0))
    {
        return 0;
    }

    size_t skip = 0;
    int    variant;
    /* which initial list: ^ goes by the context's seen_newline, \b / \B by its seen_word
     * (sre_vm_pike.c:472-473, 586-601, 851-860) — SRE_DFA_INIT_* */
    if (ctx->empty_capture) {                               /* :179-196 */
        skip = 1;
        variant = input[0] == '\n' ? 1 : sre_isword(input[0]) ? 3 : 2;
    } else if (ctx->processed_bytes == 0) {
        variant = 0;
    } else {
        variant = ctx->seen_newline ? 1 : ctx->seen_word ? 3 : 2;
    }

    DeviceStream *ds = ctx->ds;
    if (stage_input(ds, input, len) != 0) return 0;
    int poisoned = 0;
    if (sre_hip_scan_one(ctx->scanner, static_cast<const uint8_t *>(ds->d_in) + skip, len - skip,
                         variant, ctx->rec, &poisoned, ds->stream) != 0)
    {
        *prc = SRE_ERROR;
        return 1;
    }
    ctx->empty_capture = 0;
    const sre_int_t rc = ctx->rec[0];
    if (rc < 0) {                                           /* no match at eof: :660-666 */
        ctx->eof = 1;
        *prc = rc == SRE_DECLINED ? SRE_DECLINED : SRE_ERROR;
        return 1;
    }
    /* buffer-relative offsets -> absolute (sre_vm_pike.c:826-828) */
    const sre_int_t base = ctx->processed_bytes + (sre_int_t) skip;
    const size_t have = sre_hip_scanner_result_slots(ctx->scanner) - 2;
    for (size_t k = 0; k < ctx->ovec_slots; k++) {
        sre_int_t v = k < have ? ctx->rec[2 + k] : -1;
        ctx->ovector[k] = v < 0 ? -1 : v + base;
    }
    /* :586-601 — only a match of regex 0 carries slot 1 */
    if (rc == 0) {
        const sre_int_t p = ctx->ovector[1] - ctx->processed_bytes;     /* chunk-relative end */
        if (p > 0) {
            ctx->seen_newline = input[p - 1] == '\n';
            ctx->seen_word = sre_isword(input[p - 1]);
        }
    }
    if (poisoned) ctx->eof = 1;                              /* :616-622 */
    ctx->processed_bytes = ctx->ovector[1];                  /* :624-628 */
    ctx->empty_capture = (ctx->ovector[0] == ctx->ovector[1]);
    *prc = rc;
    return 1;
}


/*
 * One reference exec() of a stream that is fed in chunks (eof or not), on the
 * table-driven scanner: the ordered thread list travels from chunk to chunk as the
 * automaton state plus one capture vector per listed thread (sre_k_stream_tail), so
 * SRE_AGAIN, the temporary match range and pending matches are the reference's
 * (sre_vm_pike.c:640-688, 692-735).  Returns 0 when the call is not taken (the exact
 * VM kernel then owns the context for good).
 */
static int
pike_stream_route(sre_vm_pike_ctx_t *ctx, sre_char *input, size_t len, unsigned eof,
                  sre_int_t **pending_matched, sre_int_t *prc)
{
    if (ctx->eof) {
        *prc = SRE_ERROR;                                   /* :165-168 */
        return 1;
    }
    if (!ctx->scanner_tried) {
        ctx->scanner_tried = 1;
        ctx->scanner = compat_scanner(ctx->prog, SRE_HIP_PIKE_FIRST);
        if (ctx->scanner) {
            ctx->rec = static_cast<sre_int_t *>(
                sre_palloc(ctx->pool, sre_hip_scanner_result_slots(ctx->scanner) * sizeof(sre_int_t)));
            if (ctx->rec == NULL) ctx->scanner = NULL;
        }
    }
    /* look-ahead programs: a splice at the first byte of a chunk does not see the byte in
     * front of it but the context's seen_newline / seen_word (sre_vm_pike.c:276-285, 492):
     * their chunks run on an automaton that has the states a chunk boundary makes of a list
     * (sre_dfa.h `rekind`) */
    sre_hip_scanner_t *sc = ctx->scanner;
    if (ctx->prog->lookahead_asserts) {
        if (!ctx->stream_scanner_tried) {
            ctx->stream_scanner_tried = 1;
            ctx->stream_scanner = compat_scanner(ctx->prog, SRE_HIP_PIKE_FIRST, 1);
        }
        sc = ctx->stream_scanner;
    }
    if (sc == NULL || !sre_hip_scanner_streams(sc) || ctx->ovec_slots > SRE_STREAM_MAX_SLOTS) return 0;
    DeviceStream *ds = ctx->ds;
    size_t        skip = 0;
    int           variant = 0;
    uint32_t      entry = 0;
    if (!ctx->stream_mode) {
        /* a search starts with this chunk (prologue of :165-233): which initial list —
         * SRE_DFA_INIT_*, as in pike_scan_route */
        if (ctx->empty_capture) {                           /* :179-196 */
            if (len == 0) return 0;
            skip = 1;
            variant = input[0] == '\n' ? 1 : sre_isword(input[0]) ? 3 : 2;
        } else if (ctx->processed_bytes != 0) {
            variant = ctx->seen_newline ? 1 : ctx->seen_word ? 3 : 2;
        }
    } else {
        entry = sre_hip_scanner_chunk_entry(sc, ctx->stream_state, ctx->seen_newline ? 1 : ctx->seen_word ? 2 : 0);
    }
    if (ds->d_sctx == NULL) {
        if (hipMalloc(reinterpret_cast<void **>(&ds->d_sctx), sizeof(sre_stream_ctx_t)) != hipSuccess
            || hipHostMalloc(reinterpret_cast<void **>(&ds->h_sres), sizeof(sre_stream_result_t), hipHostMallocMapped)
                   != hipSuccess
            || hipHostGetDevicePointer(reinterpret_cast<void **>(&ds->d_sres), ds->h_sres, 0) != hipSuccess)
        {
            return 0;
        }
    }
    /* (a search that starts with this chunk: the tail kernel ignores what the context holds) */
    static const bool dbg_t = getenv("SRE_HIP_DEBUG_TIMING") != NULL;
    const auto t_a = std::chrono::steady_clock::now();
    /* (a chunk that travels through the ring is still being copied when the scan is queued) */
    RingTicket ticket;
    if (stage_input(ds, input, len, dbg_t ? nullptr : &ticket) != 0) return 0;
    const auto t_b = std::chrono::steady_clock::now();
    if (dbg_t) {
        (void) hipStreamSynchronize(ds->stream);
        const auto t_c = std::chrono::steady_clock::now();
        fprintf(stderr, "[sregex-hip] chunk %zu B: stage %.1f us (+%.1f us until landed)", len,
                std::chrono::duration<double, std::micro>(t_b - t_a).count(), std::chrono::duration<double, std::micro>(t_c - t_b).count());
    }
    const auto t_d = std::chrono::steady_clock::now();
    struct ChunkTimer {
        bool on; std::chrono::steady_clock::time_point t0;
        ~ChunkTimer() { if (on) fprintf(stderr, ", scan+tail %.1f us\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count()); }
    } chunk_timer{dbg_t, t_d};
    ds->h_sres->rc = SRE_STREAM_PENDING;
    if (sre_hip_scan_stream_chunk(sc, static_cast<const uint8_t *>(ds->d_in) + skip, len - skip, variant,
                                  ctx->stream_mode, entry, eof ? 1 : 0,
                                  (int64_t) ctx->processed_bytes + (int64_t) skip, ds->d_sctx, ds->d_sres, ds->h_sres,
                                  (uint32_t) ctx->ovec_slots, ds->stream, ring_midway, &ticket) != 0
        || ticket.ds == ds + 1)     /* (ring_midway marks a failed copy) */
    {
        ds->failed = 1;
        *prc = SRE_ERROR;
        return 1;
    }
    const sre_stream_result_t *res = ds->h_sres;
    const sre_int_t            rc = (sre_int_t) res->rc;
#ifdef SRE_DEBUG_TAIL
    if (dbg_t) {
        fprintf(stderr, "  tail: staging %.1f us, chain check %.1f us, lane 0 %.1f us\n", res->ov[SRE_STREAM_MAX_SLOTS - 3] / 100.0,
                res->ov[SRE_STREAM_MAX_SLOTS - 2] / 100.0, res->ov[SRE_STREAM_MAX_SLOTS - 1] / 100.0);
    }
#endif
    if (rc == SRE_STREAM_PENDING || rc == SRE_STREAM_UNSETTLED) {
        /* the tail kernel never published: a device fault.  The stream is not parked for re-use */
        ds->failed = 1;
        *prc = SRE_ERROR;
        return 1;
    }
    ctx->empty_capture = 0;
    ctx->at_boundary = 0;
    /* What EVERY call does on its way out (sre_vm_pike.c:586-601), SRE_AGAIN included: a MATCH
     * reached during this call — pending or final — that ends behind the call's first byte
     * refreshes seen_newline / seen_word from the byte in front of its end (last_matched_pos =
     * slot 1 of the internal vector: a match of regex 0 only).  The next chunk's look-ahead
     * threads go by these flags (sre_dfa.h `rekind`). */
    if (res->ev_in_chunk && res->ev_slot1 >= 0) {
        const sre_int_t p = (sre_int_t) res->ev_slot1 - ctx->processed_bytes;      /* chunk-relative end */
        if (p > 0 && (size_t) p <= len) {
            ctx->seen_newline = input[p - 1] == '\n';
            ctx->seen_word = sre_isword(input[p - 1]);
        }
    }
    if (rc == SRE_AGAIN) {
        for (size_t k = 0; k < ctx->ovec_slots && k < 2; k++) ctx->ovector[k] = (sre_int_t) res->ov[k];
        if (pending_matched) {
            if (res->has_pending) {
                ctx->pending[0] = (sre_int_t) res->pending[0];
                ctx->pending[1] = (sre_int_t) res->pending[1];
                *pending_matched = ctx->pending;
            } else {
                *pending_matched = NULL;
            }
        }
        ctx->processed_bytes += (sre_int_t) len;            /* :673 */
        ctx->stream_mode = 1;
        ctx->stream_state = (uint32_t) res->next_state;
        *prc = SRE_AGAIN;
        return 1;
    }
    ctx->stream_mode = 0;
    ctx->stream_state = 0;
    if (rc < 0) {
        ctx->eof = 1;
        *prc = rc == SRE_DECLINED ? SRE_DECLINED : SRE_ERROR;
        return 1;
    }
    for (size_t k = 0; k < ctx->ovec_slots; k++) ctx->ovector[k] = (sre_int_t) res->ov[k];
    if (res->poisoned) ctx->eof = 1;                         /* :616-622 */
    ctx->processed_bytes = ctx->ovector[1];                  /* :624-628 */
    ctx->empty_capture = (ctx->ovector[0] == ctx->ovector[1]);
    ctx->at_boundary = 1;
    *prc = rc;
    return 1;
}

extern "C" SRE_API sre_vm_pike_ctx_t *
sre_vm_pike_create_ctx(sre_pool_t *pool, sre_program_t *prog, sre_int_t *ovector,
    size_t ovecsize)
{
    sre_vm_pike_ctx_t *ctx =
        static_cast<sre_vm_pike_ctx_t *>(sre_pcalloc(pool, sizeof(sre_vm_pike_ctx_t)));
    if (ctx == NULL) return NULL;
    ctx->pool = pool;
    ctx->prog = prog;
    ctx->ovector = ovector;
    ctx->ovec_slots = ovecsize / sizeof(sre_int_t);
    ctx->ds = NULL;     /* device state is opened by the first exec() */
    ctx->at_boundary = 1;
    return ctx;
}

extern "C" SRE_API sre_int_t
sre_vm_pike_exec(sre_vm_pike_ctx_t *ctx, sre_char *input, size_t len, unsigned eof,
    sre_int_t **pending_matched)
{
    if (ctx->ds == NULL) {
        sre_hip_program_s *dp = sre_hip_program_get(ctx->prog);
        if (dp == NULL) return SRE_ERROR;
        size_t ctx_bytes = dp->pike_layout.total;
        if (sre_hip_program_pwave(dp, ctx->prog) != NULL) ctx_bytes = sre_pwave_ctx_bytes(dp->h_pwave);
        ctx->ds = device_stream_open(ctx->pool, ctx->prog, ctx_bytes, ctx->ovec_slots);
        if (ctx->ds == NULL) return SRE_ERROR;
    }
    DeviceStream *ds = ctx->ds;
    if (ds->failed) return SRE_ERROR;

    /* large whole-buffer call on a context that sits between two searches: the
     * table-driven scanner does it at device speed; the context's state stays
     * on the host until the VM kernel is needed */
    if (!ctx->vm_touched && ctx->at_boundary && eof && len >= SRE_COMPAT_SCAN_MIN_BYTES
        && ctx->ovec_slots >= 2)
    {
        sre_int_t rc;
        /* a small buffer: as the one chunk of a stream — scan + tail, the result in pinned memory
         * the host watches; the batch path below ends with two copies back and a wait on the
         * queue (a 63-byte call: 53 us) */
        if (len <= SRE_COMPAT_ONE_CHUNK_MAX && !ctx->prog->lookahead_asserts
            && pike_stream_route(ctx, input, len, 1, pending_matched, &rc))
        {
            g_route_counts[0]++;
            return rc;
        }
        if (pike_scan_route(ctx, input, len, &rc)) {
            g_route_counts[0]++;
            return rc;
        }
    }
    /* a stream fed in chunks: the scanner carries the thread list from chunk to chunk */
    if (!ctx->vm_touched && ctx->ovec_slots >= 2
        && (ctx->stream_mode || (ctx->at_boundary && !eof && len >= SRE_COMPAT_STREAM_MIN_BYTES)))
    {
        sre_int_t rc;
        if (pike_stream_route(ctx, input, len, eof, pending_matched, &rc)) {
            g_route_counts[1]++;
            return rc;
        }
        if (ctx->stream_mode) return SRE_ERROR;     /* (not reached: a stream in scanner mode stays there) */
    }
    if (!ctx->vm_touched) {
        ctx->vm_touched = 1;
        if (ctx->processed_bytes || ctx->eof || ctx->empty_capture || ctx->seen_newline || ctx->seen_word) {
            ds->preset_valid = 1;
            ds->preset_processed = ctx->processed_bytes;
            ds->preset_flags = (ctx->empty_capture ? SRE_PRESET_EMPTY_CAPTURE : 0)
                               | (ctx->seen_newline ? SRE_PRESET_SEEN_NEWLINE : 0)
                               | (ctx->seen_word ? SRE_PRESET_SEEN_WORD : 0)
                               | (ctx->eof ? SRE_PRESET_EOF : 0);
        }
    }

    g_route_counts[2]++;
    if (device_stream_exec(ds, input, len, eof, pending_matched ? 1u : 0u, ctx->ovec_slots,
                           sre_launch_pike_exec) != 0)
    {
        return SRE_ERROR;
    }

    const sre_dev_result_t *res = &ds->h_blk->res;
    sre_int_t               rc = (sre_int_t) res->rc;
    ctx->at_boundary = 0;       /* from here on the device context is authoritative */

    if (rc >= 0) {
        /* complete match: the whole caller ovector is defined (reference
         * sre_vm_pike.c:978-986) */
        for (size_t k = 0; k < ctx->ovec_slots; k++) ctx->ovector[k] = (sre_int_t) ds->h_blk->ov[k];
        if (res->pad[0] & 16) {
            /* The context is between two searches again (:624-628), and everything it holds there
             * came back with the result: the host takes the state over and the next call is routed
             * afresh — a stream whose first chunk was a short header line no longer keeps every
             * later search on the exact VM (round-2 advisor finding).  The device context is
             * flagged fresh so that the VM kernel, if it is needed again, starts from the preset. */
            ctx->processed_bytes = (sre_int_t) res->pad[1];
            ctx->empty_capture = (res->pad[0] & SRE_PRESET_EMPTY_CAPTURE) != 0;
            ctx->seen_newline = (res->pad[0] & SRE_PRESET_SEEN_NEWLINE) != 0;
            ctx->seen_word = (res->pad[0] & SRE_PRESET_SEEN_WORD) != 0;
            ctx->eof = (res->pad[0] & SRE_PRESET_EOF) != 0;
            ds->ctx_fresh = 1;
            ds->preset_valid = ds->preset_flags = 0;
            ds->preset_processed = 0;
            ctx->vm_touched = 0;
            ctx->at_boundary = 1;
        }
    } else if (rc == SRE_AGAIN) {
        /* temporary $& range only (reference sre_vm_pike.c:700-701) */
        for (size_t k = 0; k < ctx->ovec_slots && k < 2; k++) {
            ctx->ovector[k] = (sre_int_t) ds->h_blk->ov[k];
        }
    }
    if (pending_matched && (rc == SRE_AGAIN)) {
        if (res->has_pending) {
            ctx->pending[0] = (sre_int_t) res->pending[0];
            ctx->pending[1] = (sre_int_t) res->pending[1];
            *pending_matched = ctx->pending;
        } else {
            *pending_matched = NULL;
        }
    }
    return rc;
}

/* -------------------------------------------------------------- Thompson */

struct sre_vm_thompson_ctx_s {
    sre_pool_t    *pool;
    sre_program_t *prog;
    DeviceStream  *ds;
    int            started;
    sre_hip_scanner_t *scanner;
    int            scanner_tried;
    /* a stream fed in chunks on the scanner: the list travels as the automaton state */
    int            stream_mode, finished;
    uint32_t       stream_state;
};

/* One chunk of a chunked stream (sre_vm_thompson.c:63-270) on the scanner.  Without
 * look-ahead assertions no closure runs at the first byte of a later chunk, so the
 * chunk-local \A / ^ of this VM (:302-317) cannot be observed and the whole-buffer
 * automaton is exact. */
static int
thompson_stream_route(sre_vm_thompson_ctx_t *ctx, sre_char *input, size_t len, unsigned eof, sre_int_t *prc)
{
    if (!ctx->scanner_tried) {
        ctx->scanner_tried = 1;
        ctx->scanner = compat_scanner(ctx->prog, SRE_HIP_THOMPSON);
    }
    /* Look-ahead programs stay on the exact VM kernel: \A, ^ and the word flag of \b / \B are
     * local to the buffer of a call in this VM (sre_vm_thompson.c:302-325), so a splice at the
     * first byte of a later chunk lets threads through that the whole-buffer run never lists —
     * and exactly there this VM's plain de-duplication (no SPLIT re-descent, splices appended
     * at the END of the list, :226-230, :280-284) parts from the Pike automaton's: the chunked
     * automaton's "start of the buffer" boundary kind (sre_dfa.h `rekind`, 3) reproduces the
     * assertions but not that, found by the CPU model on a random pattern
     * (tests/test_dfa_model.py).  Without look-ahead no closure runs at a chunk's first byte. */
    sre_hip_scanner_t *sc = ctx->scanner;
    if (sc == NULL || !sre_hip_scanner_streams(sc) || ctx->prog->lookahead_asserts) return 0;
    DeviceStream *ds = ctx->ds;
    if (ds->d_sctx == NULL) {
        if (hipMalloc(reinterpret_cast<void **>(&ds->d_sctx), sizeof(sre_stream_ctx_t)) != hipSuccess
            || hipHostMalloc(reinterpret_cast<void **>(&ds->h_sres), sizeof(sre_stream_result_t), hipHostMallocMapped)
                   != hipSuccess
            || hipHostGetDevicePointer(reinterpret_cast<void **>(&ds->d_sres), ds->h_sres, 0) != hipSuccess)
        {
            return 0;
        }
    }
    RingTicket ticket;
    if (stage_input(ds, input, len, &ticket) != 0) return 0;
    ds->h_sres->rc = SRE_STREAM_PENDING;
    if (sre_hip_scan_stream_chunk(sc, ds->d_in, len, 0, ctx->stream_mode, ctx->stream_state, eof ? 1 : 0, 0,
                                  ds->d_sctx, ds->d_sres, ds->h_sres, 0, ds->stream, ring_midway, &ticket) != 0
        || ticket.ds == ds + 1)
    {
        ds->failed = 1;
        *prc = SRE_ERROR;
        return 1;
    }
    const sre_int_t rc = (sre_int_t) ds->h_sres->rc;
    if (rc == SRE_STREAM_PENDING || rc == SRE_STREAM_UNSETTLED) {
        ds->failed = 1;                     /* the tail kernel never published: a device fault */
        *prc = SRE_ERROR;
        return 1;
    }
    ctx->started = 1;
    if (rc == SRE_AGAIN) {
        ctx->stream_mode = 1;
        ctx->stream_state = (uint32_t) ds->h_sres->next_state;
    } else {
        ctx->stream_mode = 0;
        ctx->finished = 1;
    }
    *prc = rc == SRE_OK || rc == SRE_AGAIN || rc == SRE_DECLINED ? rc : SRE_ERROR;
    return 1;
}

extern "C" SRE_API sre_vm_thompson_ctx_t *
sre_vm_thompson_create_ctx(sre_pool_t *pool, sre_program_t *prog)
{
    sre_vm_thompson_ctx_t *ctx =
        static_cast<sre_vm_thompson_ctx_t *>(sre_pcalloc(pool, sizeof(sre_vm_thompson_ctx_t)));
    if (ctx == NULL) return NULL;
    ctx->pool = pool;
    ctx->prog = prog;
    return ctx;
}

extern "C" SRE_API sre_int_t
sre_vm_thompson_exec(sre_vm_thompson_ctx_t *ctx, sre_char *input, size_t len, unsigned eof)
{
    if (ctx->ds == NULL) {
        sre_hip_program_s *dp = sre_hip_program_get(ctx->prog);
        if (dp == NULL) return SRE_ERROR;
        ctx->ds = device_stream_open(ctx->pool, ctx->prog, dp->thompson_layout.total, 0);
        if (ctx->ds == NULL) return SRE_ERROR;
    }
    if (ctx->ds->failed) return SRE_ERROR;
    if (!ctx->finished && (ctx->stream_mode || (!ctx->started && !eof && len >= SRE_COMPAT_STREAM_MIN_BYTES))) {
        sre_int_t rc;
        if (thompson_stream_route(ctx, input, len, eof, &rc)) {
            g_route_counts[1]++;
            return rc;
        }
        if (ctx->stream_mode) return SRE_ERROR;
    }
    if (!ctx->started && eof && len >= SRE_COMPAT_SCAN_MIN_BYTES && len <= SRE_COMPAT_ONE_CHUNK_MAX) {
        /* a small buffer: as the one chunk of a stream (see sre_vm_pike_exec) */
        sre_int_t rc;
        if (thompson_stream_route(ctx, input, len, 1, &rc)) {
            g_route_counts[0]++;
            return rc;
        }
    }
    if (!ctx->started && eof && len >= SRE_COMPAT_SCAN_MIN_BYTES) {
        /* first and only chunk of a large stream: match / no match from the scanner */
        if (!ctx->scanner_tried) {
            ctx->scanner_tried = 1;
            ctx->scanner = compat_scanner(ctx->prog, SRE_HIP_THOMPSON);
        }
        DeviceStream *ds = ctx->ds;
        if (ctx->scanner) {
            sre_int_t *rec = static_cast<sre_int_t *>(
                sre_palloc(ctx->pool, sre_hip_scanner_result_slots(ctx->scanner) * sizeof(sre_int_t)));
            if (rec == NULL) return SRE_ERROR;
            rec[0] = SRE_ERROR;
            if (stage_input(ds, input, len) == 0
                && sre_hip_scan_one(ctx->scanner, ds->d_in, len, 0, rec, NULL, ds->stream) == 0)
            {
                ctx->started = 1;
                return rec[0] == SRE_OK ? SRE_OK : (rec[0] == SRE_DECLINED ? SRE_DECLINED : SRE_ERROR);
            }
        }
    }
    ctx->started = 1;
    if (device_stream_exec(ctx->ds, input, len, eof, 0, 0, sre_launch_thompson_exec) != 0) {
        return SRE_ERROR;
    }
    return (sre_int_t) ctx->ds->h_blk->res.rc;
}

/* ------------------------------------------------------------------- JIT */
/* The x86-64 DynASM JIT (reference sre_vm_thompson_jit.c) is dropped; the
 * entry points remain so that the reference's clients link.  compile() says
 * SRE_DECLINED, which they treat as "JIT disabled" (src/sre_cli.c:419-424). */

extern "C" SRE_API sre_int_t
sre_vm_thompson_jit_compile(sre_pool_t *pool, sre_program_t *prog,
    sre_vm_thompson_code_t **pcode)
{
    (void) pool;
    (void) prog;
    if (pcode) *pcode = NULL;
    return SRE_DECLINED;
}

extern "C" SRE_API sre_vm_thompson_ctx_t *
sre_vm_thompson_jit_create_ctx(sre_pool_t *pool, sre_program_t *prog)
{
    (void) pool;
    (void) prog;
    return NULL;
}

extern "C" SRE_API sre_vm_thompson_exec_pt
sre_vm_thompson_jit_get_handler(sre_vm_thompson_code_t *code)
{
    (void) code;
    return NULL;
}

extern "C" SRE_API sre_int_t
sre_vm_thompson_jit_free(sre_vm_thompson_code_t *code)
{
    (void) code;
    return SRE_OK;
}
