/*
 * sregex_hip.h — ADDITIVE device-resident, batched entry points (C ABI).
 *
 * The reference's executors take host pointers
 * (sre_vm_pike_exec / sre_vm_thompson_exec, reference src/sregex/sregex.h:133-134,
 * 147-148), so through them a GPU matcher is PCIe-bound.  These entry points
 * are what a maintainer binds when the streams already live in HBM: many
 * independent streams (one context each, reference README.markdown:376) are
 * scanned in one call with the same compiled sre_program_t.  They replace, per
 * stream, the call sequence
 *     ctx = sre_vm_pike_create_ctx(pool, prog, ovector, ovecsize);   sre_vm_pike.c:94-145
 *     rc  = sre_vm_pike_exec(ctx, stream, len, 1, NULL);             sre_vm_pike.c:148-689
 * (resp. the Thompson pair, sre_vm_thompson.c:25-60 / :63-270), and for
 * SRE_HIP_PIKE_COUNT the find-all iteration a caller writes around it
 * (re-feeding from ovector[1]; sre_vm_pike.c:179-196, 624-628).
 *
 * Plain pointers and sizes only; no torch / C++ types.  See INTEGRATION.md for
 * the C, ctypes and cgo-style bindings.
 */
#ifndef SREGEX_AMD_SREGEX_HIP_H
#define SREGEX_AMD_SREGEX_HIP_H

#include <sregex/sregex.h>

#ifdef __cplusplus
extern "C" {
#endif

/* what to compute per stream */
enum {
    SRE_HIP_THOMPSON   = 0,  /* match / no match                 (sre_vm_thompson_exec) */
    SRE_HIP_PIKE_FIRST = 1,  /* first match: regex id + captures (sre_vm_pike_exec)     */
    SRE_HIP_PIKE_COUNT = 2   /* iterate sre_vm_pike_exec from each match end: count     */
};

/* which device engine runs it */
enum {
    SRE_HIP_ENGINE_AUTO = 0, /* table-driven scanner when the program admits one  */
    SRE_HIP_ENGINE_VM   = 1, /* exact bytecode VM kernel, one lane per stream     */
    SRE_HIP_ENGINE_SCAN = 2, /* table-driven segment-parallel scanner, or fail    */
    SRE_HIP_ENGINE_NFA  = 3  /* bit-parallel NFA scanner (<= 64 list-able threads held as
                                a 64-bit mask per lane), or fail: the tier for programs
                                whose ordered-list automaton is too large */
};

typedef struct sre_hip_scanner_s  sre_hip_scanner_t;

/* number of HIP devices visible to this process (0 if none) */
SRE_API int sre_hip_device_count(void);

/* select the device used by subsequently created programs/scanners */
SRE_API int sre_hip_set_device(int ordinal);

/*
 * Create a scanner for `prog`.  Owned by `pool` (freed by sre_destroy_pool).
 * Returns NULL (with a diagnostic on stderr) when no HIP device is usable or
 * when `engine` == SRE_HIP_ENGINE_SCAN and the program admits no table.
 */
SRE_API sre_hip_scanner_t *sre_hip_scanner_create(sre_pool_t *pool,
    sre_program_t *prog, int mode, int engine);

/* engine actually chosen: SRE_HIP_ENGINE_VM, SRE_HIP_ENGINE_SCAN or SRE_HIP_ENGINE_NFA */
SRE_API int sre_hip_scanner_engine(sre_hip_scanner_t *sc);

/*
 * Tuning / testing knob of the table-driven scanner: bytes per segment (one
 * lane walks one segment).  0 restores the automatic choice (about 256K lanes
 * per call, at least 4 KiB per segment); otherwise a multiple of 64.
 */
SRE_API int sre_hip_scanner_set_segment_bytes(sre_hip_scanner_t *sc, size_t bytes);

/* diagnostics: fix-up rounds the last scan needed (0 = every assumed segment
 * entry state was right) */
SRE_API int sre_hip_scanner_last_fixups(sre_hip_scanner_t *sc);

/* diagnostics: 1 when the last scan's speculative fix-up rounds did not settle and the
 * exact entry state of every remaining segment was computed by composing the
 * segments' transition functions (FIRST / Thompson; an automaton that never forgets) */
SRE_API int sre_hip_scanner_last_exact_passes(sre_hip_scanner_t *sc);

/* diagnostics: how many sre_vm_pike_exec / sre_vm_thompson_exec calls of this process went
 * where — out[0] one whole buffer through a throughput scanner, out[1] a chunk of a chunked
 * stream through the table-driven scanner, out[2] the exact VM kernel */
SRE_API void sre_hip_compat_route_counts(unsigned long long out[3]);

/* The compat entry points (sregex.h) keep released device streams — a HIP stream, a VM context, staging
 * buffers of at most 8 MiB each — in a process-wide free list of at most 32 for the next context.  This
 * call frees them all and returns how many there were.
 * THREADS: as in the reference, distinct programs and pools may be used from different threads (the
 * library's process-wide state is locked); the contexts of ONE program must not run concurrently (they
 * share the program's device scanners — the reference's programs carry the VM's generation tags,
 * src/sregex/sre_vm_bytecode.h:51, and are not re-entrant either).  A scanner of the batched API below
 * belongs to one thread at a time. */
SRE_API int sre_hip_compat_trim(void);

/* diagnostics: find-all counting on the NFA tier (a program the step automaton declines) is a loop of
 * first-match searches, run in rounds over all streams of the call: how many rounds the last call took */
SRE_API int sre_hip_scanner_last_count_rounds(sre_hip_scanner_t *sc);

/* diagnostics: 1 when the last scan had to build per-segment ancestor maps to
 * reconstruct the captures of a match spanning many segments */
SRE_API int sre_hip_scanner_last_lineage_passes(sre_hip_scanner_t *sc);

/* class bits per input byte of the scanner's fast table (1, 2, 4 or 8: one table
 * lookup advances 8 / bits bytes); 0 for the VM engine.  Names the kernel
 * variant (sre_k_scan<mode, bits>) in profiles. */
SRE_API int sre_hip_scanner_class_bits(sre_hip_scanner_t *sc);

/* name of the dominant kernel of a scan with this scanner, as rocprofv3 prints it
 * (e.g. "sre_k_scan<1, 2>"); owned by the scanner */
SRE_API const char *sre_hip_scanner_kernel_name(sre_hip_scanner_t *sc);

/* measurement: duration (ms) of the segment-scan kernel of the last enqueued
 * scan, from hipEvents recorded on the caller's stream around that launch;
 * -1 when the exact VM engine ran.  Waits for the kernel. */
SRE_API double sre_hip_scanner_last_kernel_ms(sre_hip_scanner_t *sc);

/* Two scanners taking turns on ONE stream: with a tail stream set, everything a call queues
 * behind its scan kernel (chain check, capture walk, the copy of the records) goes to that
 * stream, ordered after the scan by an event; the next call's scan kernel then follows
 * on the scan stream without a gap.  NULL: everything on the stream of the call. */
SRE_API int sre_hip_scanner_set_tail_stream(sre_hip_scanner_t *sc, void *hip_stream);

/*
 * Make `hip_stream` wait until the dominant (segment-scan) kernel of sc's last enqueued
 * scan has finished — not for the small kernels and copies behind it.  A driver that
 * alternates two scanners on two streams calls this on the OTHER scanner before each
 * enqueue: the big kernels then run one after the other (neither is slowed down by
 * sharing the GPU), while the chain check, the capture walk and the copy of the
 * records of one step overlap with the scan of the next.  No-op before the first scan.
 */
SRE_API int sre_hip_scanner_order_after_scan(sre_hip_scanner_t *sc, void *hip_stream);

/* segment size the last scan used (0 for the VM engine) */
SRE_API size_t sre_hip_scanner_last_segment_bytes(sre_hip_scanner_t *sc);

/*
 * Per-stream result record, in sre_int_t units:
 *     [0] rc     regex id (>= 0; SRE_OK for Thompson) of the (last) match, or
 *                SRE_DECLINED when there is none; SRE_ERROR when the iteration
 *                of COUNT mode ended with SRE_ERROR (sre_vm_pike.c:165-168 after
 *                :616-622) — [1] and the ovector are still those of the matches
 *                found before
 *     [1] count  matches found (COUNT mode; 0/1 otherwise)
 *     [2..]      ovector of the (last) match, 2 * (max_ncaps + 1) slots,
 *                absolute byte offsets, -1 = unset   (sre_vm_pike.c:945-989)
 * sre_hip_scanner_result_slots() = 2 + 2 * (max_ncaps + 1).
 */
SRE_API size_t sre_hip_scanner_result_slots(sre_hip_scanner_t *sc);

/*
 * Enqueue the scan of `nstreams` device-resident streams on `hip_stream`
 * (a hipStream_t, NULL = default stream).  `d_streams[i]` is a DEVICE pointer
 * to `lens[i]` bytes; both arrays are HOST arrays.  Asynchronous: returns
 * after the kernels are queued.  0 on success, -1 on failure.
 *
 * The copy of the result records to pinned host memory is part of the queued
 * work and sre_hip_scan_results() waits for an event behind it, not for the
 * whole stream: a caller that alternates two scanners (enqueue on one, then
 * collect from the other) keeps the GPU busy without a gap between scans.
 * One call per scanner may be in flight; the streams must stay unchanged until
 * sre_hip_scan_results() has returned.
 */
SRE_API int sre_hip_scan_enqueue(sre_hip_scanner_t *sc,
    const void *const *d_streams, const size_t *lens, size_t nstreams,
    void *hip_stream);

/*
 * Wait for the last enqueued scan and copy its records to `results`
 * (host, nstreams * result_slots entries).  0 on success.
 */
SRE_API int sre_hip_scan_results(sre_hip_scanner_t *sc, sre_int_t *results);

/* convenience: enqueue + results */
SRE_API int sre_hip_scan_batch(sre_hip_scanner_t *sc,
    const void *const *d_streams, const size_t *lens, size_t nstreams,
    sre_int_t *results, void *hip_stream);

/* ---- helpers for drivers that have no HIP runtime binding of their own ---- */

/* device buffer management (hipMalloc / hipFree / hipMemcpy) */
SRE_API void *sre_hip_alloc(size_t bytes);
SRE_API void  sre_hip_free(void *d_ptr);
SRE_API int   sre_hip_upload(void *d_dst, const void *h_src, size_t bytes);
SRE_API int   sre_hip_download(void *h_dst, const void *d_src, size_t bytes);
SRE_API int   sre_hip_synchronize(void *hip_stream);

/*
 * Fill d_dst[0..n) with the reference benchmark stream, generated on device
 * (bench/gen-data.pl:9 restated):  byte i is "abccc"[i % 5] for
 * i < n - tail_len, and the last tail_len bytes are `tail`.  With
 * n = 5 * k + tail_len this is exactly  "abccc" x k . tail.
 */
SRE_API int sre_hip_gen_data(void *d_dst, size_t n, const void *h_tail,
    size_t tail_len, void *hip_stream);

/*
 * Plain streaming read of n bytes (16 B per lane, grid-stride) — the box's
 * measured HBM read ceiling, reported next to the scanner's rate.  Writes one
 * checksum word per workgroup into an internal buffer.  Asynchronous.
 */
SRE_API int sre_hip_read_ceiling(const void *d_src, size_t n, void *hip_stream);

/*
 * The scanner's staging access pattern with no automaton work: the first
 * n / seg_bytes rows of seg_bytes (a multiple of 128) each, one row per lane,
 * tile (64, 128 or 256) bytes of every row per round, fetched one round ahead.
 * lds_bytes of dynamic LDS are requested only to pin the workgroups per CU to
 * what the scanner gets.  The rate this reaches is the ceiling the access
 * pattern itself allows; bench.py reports it next to the plain read ceiling.
 * Asynchronous.
 */
SRE_API int sre_hip_read_pattern(const void *d_src, size_t n, unsigned seg_bytes,
    unsigned tile, unsigned lds_bytes, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif
