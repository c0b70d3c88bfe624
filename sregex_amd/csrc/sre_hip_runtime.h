/*
 * sre_hip_runtime.h — host-side HIP plumbing shared by the compat API
 * (sre_vm_api.cpp) and the batched device API (sre_hip_batch.cpp).
 */
#ifndef SRE_HIP_RUNTIME_H
#define SRE_HIP_RUNTIME_H

#include <hip/hip_runtime.h>
#include "sre_program.h"
#include "sre_hip_common.h"
#include "sre_hip_vm.h"

struct sre_dfa_s;           /* sre_dfa.h */

/* device images of one compiled program; owned by the program's pool */
struct sre_hip_program_s {
    int         device;
    void       *d_blob;         /* sre_dev_prog_hdr_t + arrays */
    size_t      blob_bytes;
    uint32_t    nclasses;
    int         has_wave;       /* the blob carries a sre_dev_wave_t: Thompson runs one wavefront per stream */
    sre_pike_layout_t      pike_layout;
    sre_thompson_layout_t  thompson_layout;
    /* step automata for the scanner (NULL until first use / if not buildable) */
    struct sre_dfa_s *dfa_pike;
    struct sre_dfa_s *dfa_thompson;
    int         dfa_pike_tried, dfa_thompson_tried;
    /* throughput scanners of the compat path (sre_vm_api.cpp), [0] Thompson [1] Pike first match */
    struct sre_hip_scanner_s *compat_scanner[4];     /* Thompson, Pike first match, and their chunked forms (look-ahead programs) */
    int         compat_tried[4];
    /* the wave form of the exact Pike step (sre_pwave.h) for the streaming VM, built on first use */
    void       *h_pwave, *d_pwave;
    int         pwave_tried;
};

#ifdef __cplusplus
extern "C" {
#endif

/* 0 when a HIP device is usable; otherwise prints ONE diagnostic to stderr and
 * returns -1.  There is no CPU matcher to fall back to. */
SRE_NOAPI int sre_hip_ready(void);

/* compute units of the current device */
SRE_NOAPI int sre_hip_cu_count(void);

/* report a HIP failure loudly; returns -1 */
SRE_NOAPI int sre_hip_fail(const char *what, hipError_t err);

/* build (once) and return the device image of `prog`, or NULL */
SRE_NOAPI struct sre_hip_program_s *sre_hip_program_get(sre_program_t *prog);

/* the wave form of `prog`'s Pike step on the device (dp->d_pwave), or NULL when the program has none */
SRE_NOAPI void *sre_hip_program_pwave(struct sre_hip_program_s *dp, sre_program_t *prog);

#define SRE_HIP_TRY(expr)                                                     \
    do {                                                                      \
        hipError_t e_ = (expr);                                               \
        if (e_ != hipSuccess) { sre_hip_fail(#expr, e_); goto hip_failed; }   \
    } while (0)

#ifdef __cplusplus
}
#endif
#endif
