/*
 * sre_dfa.h — the STEP AUTOMATON of a compiled program.
 *
 * Observation behind the scanner: in the reference Pike VM capture vectors
 * never influence control flow.  Which threads exist after a byte, in which
 * priority order, whether a MATCH fires and which threads it cuts off
 * (reference sre_vm_pike.c:235-581, 756-942) is a pure function of
 *      (ordered list of thread pcs, "a match is pending", next byte).
 * The builder therefore runs the VM's own step — same closure order, same
 * generation-tag de-duplication, same SPLIT re-descent (sre_vm_pike.c:774-784),
 * same MATCH cut-off — on capture-free thread lists and memoises the results:
 * a deterministic automaton whose states ARE the reference's thread lists.
 * It is built eagerly at scanner creation (a compile step, independent of any
 * input), with a state cap; programs over the cap are declined and run on the
 * exact VM kernel instead.  Look-ahead assertions ($ \z \b \B) are decided by
 * the next symbol inside the step (sre_dfa.cpp).
 *
 * Each transition additionally records, per surviving thread, which thread of
 * the previous list it descends from and which SAVE slots its closure path
 * wrote.  That is what lets the device reconstruct the winning thread's
 * capture vector afterwards without ever carrying per-thread captures.
 */
#ifndef SRE_DFA_H
#define SRE_DFA_H

#include "sre_program.h"

#ifdef __cplusplus
#include <vector>

enum {
    SRE_DFA_EV_NONE = 0,
    SRE_DFA_EV_DONE = 1,   /* a consuming thread's closure reached MATCH: end = pos + 1 */
    SRE_DFA_EV_POP  = 2    /* a listed MATCH thread was popped:           end = its own */
};

enum {
    SRE_DFA_INIT_START      = 0,  /* stream start:  \A holds, ^ holds        */
    SRE_DFA_INIT_RESTART_NL = 1,  /* re-armed search, ^ holds                */
    SRE_DFA_INIT_RESTART    = 2,  /* re-armed search, ^ fails                */
    SRE_DFA_INIT_RESTART_WORD = 3, /* re-armed search, ^ fails, the byte in front is a word byte (the
                                     context's seen_word, sre_vm_pike.c:472-473, 594); the same list
                                     as RESTART unless the program has \b or \B */
    SRE_DFA_NINIT           = 4
};

#define SRE_DFA_DEAD          0u      /* state 0: empty thread list */
#define SRE_DFA_MAX_THREADS   16u     /* per list, for the lineage tables */
#define SRE_DFA_NO_PARENT     0xffu   /* thread descends from the ".*?" restart */

struct sre_dfa_trans_t {
    uint32_t next;
    uint8_t  ev_kind;
    uint8_t  ev_src;        /* index (old list) of the thread that reached MATCH */
    uint8_t  ev_empty;      /* the match is empty (start == end): the caller's next search starts one
                               byte further (sre_vm_pike.c:179-196) */
    uint16_t ev_regex;
    uint64_t ev_saves;      /* DONE: slots saved on the way to MATCH (value pos + 1) */
    uint64_t ev_early;      /* slots saved by a look-ahead splice in front of the event (value pos) */
    uint32_t lin_off;       /* into lin_parent / lin_saves: one entry per NEW thread */
    uint16_t lin_n;
    uint8_t  skipped;       /* leading-byte skip: list re-seeded, no thread stepped */
};

struct sre_dfa_s {
    uint32_t nstates;               /* state 0 is DEAD */
    uint32_t ncls;                  /* byte classes; symbol ncls is EOF */
    uint8_t  cls_map[256];
    uint32_t init[SRE_DFA_NINIT];
    uint32_t max_threads;           /* longest thread list of any state */
    uint32_t nslots;
    int      has_caret;             /* program contains ^ or \A */
    int      has_lookahead;         /* program contains $ \z \b \B (a re-armed search starts from
                                       init[SRE_DFA_INIT_RESTART_WORD] when the context's seen_word is set) */
    std::vector<sre_dfa_trans_t> trans;      /* [nstates][ncls + 1] */
    std::vector<uint8_t>         lin_parent; /* old-list index or SRE_DFA_NO_PARENT */
    std::vector<uint64_t>        lin_saves;  /* slots saved on the closure path (value pos + 1) */
    std::vector<uint64_t>        lin_early;  /* slots saved by a look-ahead splice before the byte was
                                                consumed (value pos), not saved again afterwards */
    std::vector<uint8_t>         matched;    /* [nstates] a match is pending in this state */
    std::vector<uint8_t>         seen_start; /* [nstates] 0/1, 2 = reached by a leading-byte skip */
    std::vector<uint16_t>        nthreads;   /* [nstates] list length */
    std::vector<uint32_t>        list_off;   /* [nstates + 1] into list_pcs */
    std::vector<uint32_t>        list_pcs;   /* the thread lists themselves (debug / tests) */
    std::vector<uint32_t>        unskip;     /* [nstates] the state a CHUNK BOUNDARY turns this one into: a
                                                leading-byte skip that is still travelling (seen_start == 2) ends
                                                at the end of a chunk, and the next exec() starts with an ordinary
                                                initial-state check (sre_vm_pike.c:304-306, then :256-273 again):
                                                same list, seen_start == 1.  Identity elsewhere. */

    std::vector<uint32_t>        rekind;     /* [nstates][4], only when built with chunk twins: a look-ahead
                                                splice at the FIRST byte of a chunk does not see the byte in
                                                front of it but the context's flags (\b / \B: seen_word,
                                                sre_vm_pike.c:472-473; ^ / \A inside the splice: seen_newline,
                                                :851-860) — the state a chunk boundary turns this one into when
                                                the context says 0: neither, 1: seen_newline, 2: seen_word; 3: the
                                                Thompson VM (no flags: "start of the buffer" at every call).
                                                Same list, same marks; identity for lists without look-ahead. */

    const sre_dfa_trans_t &t(uint32_t s, uint32_t sym) const { return trans[(size_t) s * (ncls + 1) + sym]; }
};
typedef struct sre_dfa_s sre_dfa_t;

extern "C" {
#else
typedef struct sre_dfa_s sre_dfa_t;
#endif

/* Build the automaton, or return NULL when the program is not admitted
 * (more than `max_states` states, too many capture slots).  `why` (optional) receives a static reason string. */
sre_dfa_t *sre_dfa_build(const sre_program_t *prog, uint32_t max_states, const char **why);
/* chunk_twins != 0: also the states a chunk boundary makes of look-ahead lists (`rekind`, and
 * `unskip` for such programs): the automaton of a stream that is fed in chunks */
sre_dfa_t *sre_dfa_build2(const sre_program_t *prog, uint32_t max_states, int chunk_twins, const char **why);
void sre_dfa_free(sre_dfa_t *dfa);

#ifdef __cplusplus
}
#endif
#endif
