/*
 * sre_pwave.h — the WAVE form of a program for the exact Pike step (sre_hip_pwave.hip): what one
 * wavefront needs to take a byte step of the reference loop (sre_vm_pike.c:314-581, :756-942) with
 * its lanes instead of one lane's loop.
 *
 * The reference's closure (add_thread, :756-942) started from one source thread always completes
 * before the next source starts, so for a given source instruction and a given answer to ^ / \A its
 * order is STATIC: a depth-first walk with the SPLIT re-descent (:774-784) from `pc + 1`.  What is
 * dynamic is only which of its list-able targets an EARLIER source of the same step has listed
 * already (the generation tags, :770, :792): a target is listed by the first source that reaches it.
 * (An interior instruction an earlier closure has tagged prunes nothing new: everything reachable
 * from it was listed by that closure.)  So the builder tabulates, per entry instruction and per
 * context, the ordered list of (target thread, capture slots saved on the way), cut at the first
 * MATCH for closures started from the byte loop (SRE_DONE, :895-898) — the reference JIT's "paths"
 * (sre_vm_thompson_x64.dasc:323-395) with captures — and the kernel takes a step as: every lane
 * tests its thread against the byte, then source by source in priority order the lanes load the
 * source's list, drop what is listed already, rank the rest by a prefix count and hand the
 * captures over.
 *
 * Programs with look-ahead assertions (their splice re-enters the CURRENT list, :506-526), more than
 * 64 list-able threads or more than 64 capture slots have no wave form: the one-lane VM takes them.
 */
#ifndef SRE_PWAVE_H
#define SRE_PWAVE_H

#include <stdint.h>
#include "sre_program.h"

#define SRE_PWAVE_MAX_THREADS 64u
#define SRE_PWAVE_MAX_SLOTS   64u
#define SRE_PWAVE_NCTX        3u      /* 0: ^ and \A fail; 1: ^ holds; 2: ^ and \A hold (offset 0 of a fresh context) */

typedef struct {
    uint32_t off;           /* first entry */
    uint16_t len;           /* threads listed */
    uint8_t  done;          /* the walk reached MATCH (entry off + len holds it): SRE_DONE */
    uint8_t  sss;           /* the walk passed instruction 0's SPLIT: seen_start_state (:799-802) */
} sre_pwave_list_t;

typedef struct {
    uint16_t tid;           /* target thread */
    uint16_t pad[3];
    uint64_t saves;         /* capture slots written on the way (value: the position behind the byte) */
} sre_pwave_entry_t;

typedef struct {
    uint32_t nthreads, nslots, nlists, nentries;
    uint32_t nleading;              /* 0: the program has no leading-byte skip */
    uint32_t bytes;                 /* size of the whole block */
    uint32_t off_lists, off_entries;/* byte offsets of lists[nlists][SRE_PWAVE_NCTX], entries[nentries] */
    uint32_t tid_pc[SRE_PWAVE_MAX_THREADS];
    uint16_t tid_list[SRE_PWAVE_MAX_THREADS];   /* list of the closure behind the thread's instruction (0xffff: MATCH) */
    uint16_t tid_match[SRE_PWAVE_MAX_THREADS];  /* MATCH threads: regex id + 1, else 0 */
    uint32_t accept[SRE_PWAVE_MAX_THREADS][8];  /* bytes the thread consumes */
    uint32_t lead[8];                           /* bytes that can start a match (:992-1061) */
    uint32_t multi_ncaps_off;                   /* uint32 multi_ncaps[nregexes] behind the entries */
    uint32_t nregexes;
    /* lists[0]: the closure of instruction 0 as the start of a search lists it (MATCH is a thread);
     * lists[1 + k]: behind consuming thread k's instruction, from the byte loop */
} sre_pwave_hdr_t;

#ifdef __cplusplus
extern "C" {
#endif
/* malloc()ed block (header + arrays, position independent), or NULL when the program has no wave form */
sre_pwave_hdr_t *sre_pwave_build(const sre_program_t *prog);
#ifdef __cplusplus
}
#endif
#endif
