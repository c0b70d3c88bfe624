/*
 * sre_hip_nfa.h — device tables, per-segment summaries and launchers of the
 * bit-parallel NFA scanner (sre_hip_nfa.hip; host form: sre_nfa.h).
 */
#ifndef SRE_HIP_NFA_H
#define SRE_HIP_NFA_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sre_hip_scan.h"

typedef struct {
    uint32_t nbits, nslices;
    uint64_t init[3];               /* SRE_DFA_INIT_* -> initial thread set */
    uint64_t any_bits, match_bits;
    const uint64_t *accept;         /* [256]           device */
    const uint64_t *follow;         /* [nslices][256]  device */
    /* look-ahead assertions (sre_nfa.h): their bits are byte nslices - 1 of the mask */
    uint32_t nassert, pad;
    const uint64_t *expand;         /* [4][4][256]     device */
    const uint8_t  *kind;           /* [256]           device */
} sre_nfa_tables_t;

/* the shift-and form (sre_nfa.h sre_nfa_sa_t) as the kernel takes it */
typedef struct {
    uint32_t w64, carry, masked, evacc, nlut;
    uint32_t perm;                  /* v_perm_b32 selector: byte k = the byte of the mask that indexes lut[k] */
    uint64_t init[3];
    uint64_t seed, any_bits, match_bits, msrc, valid, self, shift_src;
    const uint64_t *accept;         /* [256]        device */
    const uint64_t *lut;            /* [nlut][256]  device (NULL when nlut == 0) */
    /* look-ahead assertions: bits 0 .. nassert - 1 of the mask (always masked, with MATCH bits) */
    uint32_t nassert, pad2;
    const uint64_t *expand;         /* [16][1 << nassert]  device */
    const uint8_t  *kind;           /* [256]               device (sre_nfa.h SRE_NFA_KIND_*) */
} sre_nfa_sa_tables_t;

/* what one lane learnt about its segment */
typedef struct {
    uint64_t s_in;          /* thread set assumed at the segment start */
    uint64_t s_out;         /* thread set at the segment end (undefined behind first_ev) */
    int64_t  first_ev;      /* position of the first step that reached MATCH, -1 none */
    int64_t  last_clean;    /* -1 none seen, else 2 * q + mode: q <= first_ev (<= segment end) is a
                               position at which the list is the freshly seeded initial closure
                               only; mode 1: the reference arrives there as the target of its
                               leading-byte skip, 0: by an ordinary step (sre_hip_nfa.hip) */
} sre_nfa_summary_t;

/* per-stream outcome of the chain check */
typedef struct {
    int64_t first_bad;      /* first segment whose assumed entry set was wrong, or nseg */
    int64_t ev_pos;         /* first MATCH event of the stream, -1 none */
    int64_t clean_pos;      /* where the exact VM starts its window (Pike) */
    int32_t done;           /* 1: verified up to the event / the end of the stream */
    int32_t clean_mode;     /* how the reference arrives at clean_pos (see sre_nfa_summary_t) */
} sre_nfa_status_t;

#ifdef __cplusplus
extern "C" {
#endif
size_t sre_nfa_lds_bytes(uint32_t nslices, int la);
int sre_nfa_blocks_per_cu(int mode, uint32_t nslices, int la);
const char *sre_nfa_kernel_name(int mode, uint32_t nslices, int la, char *buf, size_t n);
/* pass over segments [lo[s], nseg_s) of every stream (lo == NULL: all, speculative
 * entry sets from a 128-byte warm-up).  With lo: segment lo[s] enters with
 * belief[g] (its verified predecessor's exit set), later ones with belief[g] where
 * bvalid[g], i.e. with what the previous round's lane in front of them ended in. */
hipError_t sre_launch_nfa_scan(int mode, sre_nfa_tables_t tab, sre_scan_geom_t geom,
    sre_nfa_summary_t *d_sum, const int64_t *d_lo, const uint64_t *d_belief,
    const uint8_t *d_bvalid, hipStream_t stream);
/* the same pass by the shift-and kernel (sre_k_nfa_sa): summaries, beliefs and the chain check are
 * shared, the masks are in the numbering of the shift-and form */
hipError_t sre_launch_nfa_sa_scan(sre_nfa_sa_tables_t tab, sre_scan_geom_t geom,
    sre_nfa_summary_t *d_sum, const int64_t *d_lo, const uint64_t *d_belief,
    const uint8_t *d_bvalid, hipStream_t stream);
/* exact entry sets for the segments [lo[s], ...) of every unsettled stream (a program that never forgets):
 * the segments' singleton exit sets (d_mat: 64 x uint64 per segment of the batch), then the recurrence that
 * leaves every lane's exact entry set in d_belief / d_bvalid — the next pass is exact (sre_hip_nfa.hip) */
hipError_t sre_launch_nfa_exact_entries(int use_sa, sre_nfa_tables_t ptab, sre_nfa_sa_tables_t atab, sre_scan_geom_t geom,
    const sre_nfa_summary_t *d_sum, const int64_t *d_lo, uint64_t *d_mat, uint64_t *d_belief, uint8_t *d_bvalid,
    hipStream_t stream);
int sre_nfa_sa_blocks_per_cu(const sre_nfa_sa_tables_t *tab);
const char *sre_nfa_sa_kernel_name(const sre_nfa_sa_tables_t *tab, char *buf, size_t n);
size_t sre_nfa_verify_acc_bytes(uint32_t nstreams);
hipError_t sre_nfa_verify_acc_init(void *d_acc, uint32_t nstreams, hipStream_t stream);
/* chain check -> status; also refreshes belief / bvalid for a possible next round and
 * writes the records of every stream that needs no VM window (no event, or Thompson) */
hipError_t sre_launch_nfa_verify(int mode, sre_scan_geom_t geom, const sre_nfa_summary_t *d_sum,
    void *d_acc, sre_nfa_status_t *d_status, uint64_t *d_belief, uint8_t *d_bvalid,
    int64_t *d_records, uint32_t ovec_slots, const int64_t *d_lo, hipStream_t stream);
#ifdef __cplusplus
}
#endif
#endif
