/*
 * sre_scan_fast.h — the scanner's fast table as the host builds it (sre_scan_fast.cpp): entry format and builder.
 * No device types here: the CPU model of the COUNT lane (tests/scan_sim.cpp) includes it too.
 */
#ifndef SRE_SCAN_FAST_H
#define SRE_SCAN_FAST_H

#include <stdint.h>

/* ---- one 32-bit entry per (state, 8-bit index); see sre_hip_scan.h for how the index packs byte classes
 *   bits 31..10  byte offset of the next state's row (state * 1024)
 *   bit  0       SLOW: a sub-step carries a match event or kills the list
 *   bits 1..4    COUNT mode: matches completed inside this step */
#define SRE_FAST_SLOW       1u
#define SRE_FAST_CNT_SHIFT  1u
#define SRE_FAST_CNT_MASK   0xfu
/*   bit  5       STABLE (tables without COUNT's folded restarts only): the step returns to
 *                the SAME state without an event, and every thread of the state's neutral
 *                set (sre_scan_tables_t.neutral) descends from ITSELF without saving a
 *                capture slot — a thread list "looping in place" (x+ over a run of x).
 *                The capture walker jumps over stretches made of such steps only. */
#define SRE_FAST_STABLE     32u
/*   bit  5       EVT (COUNT tables, which have no STABLE entries): a sub-step recorded a match that is still
 *                pending at its end — the list lives on in a FRESH state (sre_scan_host.cpp) */
#define SRE_FAST_EVT        32u
/*   bit  6       NEXT_FRESH (COUNT tables): the state the entry ends in is FRESH — the scan kernel reads it off the
 *                last entry of a round instead of looking the state's flags up (a dependent LDS access per round) */
#define SRE_FAST_NEXT_FRESH 64u
#define SRE_STATE_FRESH     8u      /* sre_scan_tables_t.state_flags: every way into the state records a match that ends
                                       with the byte just consumed */
#define SRE_FAST_ROW_BYTES  1024u

#ifdef __cplusplus
#include "sre_dfa.h"
#include <vector>

struct sre_scan_fast_t {
    uint32_t              bits, stride;     /* class bits per input byte, input bytes per entry (bits * stride == 8) */
    uint32_t              any_fresh;        /* COUNT: some state is FRESH (entries may carry SRE_FAST_EVT) */
    std::vector<uint8_t>  fresh;            /* [nstates] */
    std::vector<uint32_t> fast;             /* [nstates][256] for `mode` */
    std::vector<uint32_t> fast_plain;       /* same without COUNT's folded restarts (== fast otherwise) */
};

void sre_scan_fast_build(const sre_dfa_t *d, int mode, sre_scan_fast_t *out);
#endif

#endif
