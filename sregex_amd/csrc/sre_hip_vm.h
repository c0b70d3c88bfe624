/*
 * sre_hip_vm.h — host-visible interface of the exact VM kernels
 * (sre_hip_vm.hip): per-stream state layouts and launchers.
 */
#ifndef SRE_HIP_VM_H
#define SRE_HIP_VM_H

#include <hip/hip_runtime.h>
#include "sre_hip_common.h"

/* byte offsets inside one Pike stream context (zero-filled == fresh) */
typedef struct {
    size_t   tags;          /* uint32_t[len + 1]     generation per instruction  */
    size_t   initial;       /* uint32_t[nthreads+1]  initial closure snapshot     */
    size_t   nodes[2];      /* two thread lists, (nthreads + 1) nodes each        */
    size_t   matched;       /* int64_t[nslots]       capture of the pending match */
    size_t   work;          /* int64_t[nslots]       closure working vector       */
    size_t   stack;         /* closure stack, len + 2 records                     */
    size_t   total;
    uint32_t node_bytes;    /* 16 + 8 * nslots                                    */
} sre_pike_layout_t;

typedef struct {
    size_t   tags, list[2], stack, total;
} sre_thompson_layout_t;

__host__ __device__ sre_pike_layout_t sre_pike_layout(uint32_t len, uint32_t nthreads,
    uint32_t nslots);
__host__ __device__ sre_thompson_layout_t sre_thompson_layout(uint32_t len);

/* per-stream window of the exact VM behind the NFA scanner: layout-identical to
 * sre_nfa_status_t (sre_hip_nfa.h) */
typedef struct {
    int64_t first_bad;
    int64_t ev_pos;         /* first MATCH event, -1 none: no window */
    int64_t clean_pos;      /* the VM starts here */
    int32_t done;
    int32_t clean_mode;     /* bit 0: the reference reaches clean_pos as a leading-byte skip target;
                               SRE_NFA_WINDOW_POISONED: set by the window kernel (see there) */
} sre_nfa_window_t;
#define SRE_NFA_WINDOW_POISONED 0x100
#define SRE_NFA_CLEAN_AFTER_NL  0x2     /* set by the chain check: the byte in front of clean_pos is a newline */
#define SRE_NFA_MATCH_AFTER_NL  0x200   /* set by the window kernel: the byte in front of the match's end is one */

/* find-all rounds on the NFA tier (sre_hip_batch.cpp): the scanned buffer of a stream starts at a
 * clean position of the current SEARCH; the exact VM runs on the stream from where that search
 * began, as the reference's re-armed context does (sre_vm_pike.c:179-196, 624-628) */
typedef struct {
    const uint8_t *vptr;        /* the stream from the start of the current search */
    uint64_t       vlen;        /* ... to its end */
    int64_t        processed;   /* absolute offset of vptr: the re-armed context's processed_bytes */
    int64_t        start_add;   /* offset of the scanned buffer inside vptr */
    uint32_t       preset_flags;/* SRE_PRESET_SEEN_NEWLINE */
    uint32_t       pad;
} sre_nfa_count_req_t;

#ifdef __cplusplus
extern "C" {
#endif
hipError_t sre_launch_pike_window(const void *blob, size_t blob_bytes, const void *const *d_streams,
    const uint64_t *d_lens, uint32_t nstreams, void *d_ctx, uint64_t ctx_stride,
    int64_t *d_records, uint32_t ovec_slots, sre_nfa_window_t *d_win, const int64_t *d_lo,
    const sre_nfa_count_req_t *d_creq, hipStream_t stream);
/* the same windows taken by one wavefront per stream (sre_hip_pwave.hip; programs with a wave form,
 * sre_pwave.h) */
struct sre_pwave_hdr_s;
hipError_t sre_launch_pike_window_wave(const void *d_wave, const void *h_wave, const void *const *d_streams,
    const uint64_t *d_lens, uint32_t nstreams, int64_t *d_records, uint32_t ovec_slots,
    sre_nfa_window_t *d_win, const int64_t *d_lo, const sre_nfa_count_req_t *d_creq, hipStream_t stream);
int sre_pwave_fits(const void *h_wave);
/* the compat API's streaming VM by a wavefront: context bytes, and one exec() per request */
size_t sre_pwave_ctx_bytes(const void *h_wave);
hipError_t sre_launch_pike_exec_wave(const void *d_wave, const void *h_wave, const sre_dev_req_t *d_reqs, uint32_t nreqs,
                                     hipStream_t stream);
/* ENGINE_VM Pike scans by the same step: whole streams, first match or the find-all iteration */
hipError_t sre_launch_pike_scan_wave(const void *d_wave, const void *h_wave, int mode, const void *const *d_streams,
    const uint64_t *d_lens, uint32_t nstreams, int64_t *d_records, uint32_t ovec_slots, hipStream_t stream);
/* ctx_bytes: size of one stream context (copied into LDS for the call when it fits) */
hipError_t sre_launch_pike_exec(const void *blob, size_t blob_bytes, const sre_dev_req_t *d_reqs,
    uint32_t nreqs, size_t ctx_bytes, hipStream_t stream);
hipError_t sre_launch_thompson_exec(const void *blob, size_t blob_bytes, const sre_dev_req_t *d_reqs,
    uint32_t nreqs, size_t ctx_bytes, hipStream_t stream);
/* whole-stream scan, one lane per stream; mode = SRE_HIP_THOMPSON / PIKE_FIRST / PIKE_COUNT */
hipError_t sre_launch_vm_scan(const void *blob, int mode, const void *const *d_streams,
    const uint64_t *d_lens, uint32_t nstreams, void *d_ctx, uint64_t ctx_stride,
    int64_t *d_records, uint32_t ovec_slots, int has_wave, hipStream_t stream);
#ifdef __cplusplus
}
#endif

#endif
