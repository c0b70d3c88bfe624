/*
 * sre_nfa.h — the BIT-PARALLEL form of a compiled program: every list-able
 * instruction (CHAR / IN / NOTIN / ANY / MATCH) is one bit of a 64-bit mask, the
 * accounting the reference's Thompson JIT uses for its 64-bit ADDED register
 * (reference src/sregex/sre_vm_thompson_x64.dasc:81-130 test-and-set per thread,
 * :264-290 thread numbering; sre_vm_thompson_jit.c:226-241 "<= 64 threads stay in
 * a register").
 *
 * One step of the reference byte loop (sre_vm_pike.c:235-581,
 * sre_vm_thompson.c:88-258) on SETS of threads is
 *
 *      T  = S & accept[byte]                    which listed threads consume it
 *      S' = OR over i in T of follow[i]         their epsilon closures (:756-942)
 *
 * and `follow` is a union-homomorphism, so it is tabulated per byte slice of T:
 * S' = OR_k slice[k][(T >> 8k) & 255].  This is exact for WHICH threads are
 * listed; it carries no priority order.  It therefore decides Thompson
 * (match / no match) exactly, and for Pike it is exact up to and including the
 * first MATCH event — before any match has cut lower-priority threads
 * (sre_vm_pike.c:535-553) the VM's list, as a set, is S.  The exact VM then runs
 * only over the window from the last CLEAN position (the list consists of the
 * freshly seeded initial closure only) to the end of the search.
 *
 * ^ holds iff the byte just consumed is a newline (sre_vm_pike.c:851-860): a
 * thread whose closure contains ^ and which can consume '\n' gets a TWIN bit —
 * the plain bit accepts every byte but '\n', the twin accepts '\n' only and
 * carries the closure computed with ^ true — so the step needs no case split.
 * \A never holds behind a consumed byte (:841-848); it only shapes the initial
 * sets.
 *
 * Look-ahead assertions ($ \z \b \B) wait in the list like any thread (one bit each,
 * all in ONE byte of the mask, at most eight) and are decided by the NEXT byte
 * (sre_vm_pike.c:450-504): one that holds puts the closure of its continuation into
 * the list at the same position (:506-526).  Whether it holds depends on the kind of
 * the byte in front of the position (word / newline / start of the stream / other) and
 * of the byte at it (word / newline / other / end of input) — the same for every
 * assertion of the list — so per such context the expansion is again a
 * union-homomorphism of the assertion bits, transitively closed, and is tabulated:
 *      S_eff = S | expand[ctx][assertion byte of S];    T = S_eff & accept[byte]; ...
 * A MATCH reached by an expansion is a match event at that position.
 */
#ifndef SRE_NFA_H
#define SRE_NFA_H

#include "sre_program.h"

#ifdef __cplusplus
#include <vector>

#define SRE_NFA_MAX_BITS 64u

/* kinds of a byte for the look-ahead assertions; as the byte IN FRONT of a position
 * START replaces them at the beginning of the stream, as the byte AT it EOF at the end */
enum { SRE_NFA_KIND_OTHER = 0, SRE_NFA_KIND_WORD = 1, SRE_NFA_KIND_NL = 2, SRE_NFA_KIND_EDGE = 3 };
#define SRE_NFA_LEADING 4u          /* the byte can start a match (sre_vm_pike.c:992-1061) */

/*
 * The SHIFT-AND form of the same sets (programs without look-ahead assertions): most compiled
 * programs are CHAINS — the thread behind `x` in `xy` lists `y` and nothing else — so with a
 * thread's successor numbered one bit above it the whole follow relation of a chain is one shift:
 *
 *      t  = S & accept[byte]
 *      S' = ((t & shift_src) << 1) | (t & self) | seed | OR_k lut[k][byte hot[k] of t]
 *
 * `self`: threads that list themselves (x+ x*).  `seed`: the closure of the ".*?" thread, which is
 * alive at every position of an unanchored search and therefore kept IMPLICIT (no bit) when its
 * closure does not depend on ^.  Everything else a thread lists — the other branches of a SPLIT, a
 * JMP back, the exit of an optional chain — comes from a lookup, by the bytes of the mask that hold
 * such threads (`hot`, at most three): one or two lookups per input byte where the plain slices need
 * nbits / 8.  Equivalent threads (same follow set, listed by exactly the same threads: the two arms of
 * `(?:a|b)`) are merged first; all MATCH instructions become one event test.
 * The layout either leaves a HOLE (a bit no byte accepts) above every thread that must not shift
 * (`masked == 0`), or the step masks with shift_src.  MATCH is either sticky bits (`match_bits`, behind
 * a chain that ends in MATCH) or, when that would cost lookups, no bit at all: `evacc`, a step is an
 * event when t & msrc.  (The x86 JIT of the reference keeps the same 64-bit thread mask,
 * sre_vm_thompson_x64.dasc:81-130; its transitions are code, here they are one shift and a table.)
 */
#ifdef __cplusplus
struct sre_nfa_sa_t {
    uint32_t nbits;             /* highest bit in use + 1 */
    uint32_t w64;               /* 0: all in the low 32 bits */
    uint32_t carry;             /* w64: the shift carries bit 31 into bit 32 (else each half shifts alone) */
    uint32_t masked, evacc;
    uint32_t nlut;              /* <= SRE_NFA_SA_MAX_LUT */
    uint32_t hot[4];
    uint64_t init[3];
    uint64_t seed, any_bits, match_bits, msrc, valid, self, shift_src;
    uint64_t accept[256];
    std::vector<uint64_t> lut;  /* [nlut][256] */
    std::vector<int>      bit_of;   /* bit of sre_nfa_s -> bit here; -1: implicit ".*?", -2: MATCH (evacc), >= 0 also for merged threads */
    uint32_t cost;              /* the builder's estimate: instructions per input byte */
    /* look-ahead assertions ($ \z \b \B), as in the plain form: they wait in the list (bits that no byte
     * accepts, all in ONE byte of the mask) and the byte AT the position decides them —
     * S |= expand[prev kind * 4 + cur kind][that byte of S] in front of the step.  Such forms are always
     * `masked` with MATCH bits (an expansion that lists MATCH is an event). */
    uint32_t nassert;
    uint32_t assert_byte;           /* byte of the mask that holds them */
    std::vector<uint64_t> expand;   /* [16][256], indexed by the whole byte (other bits ignored) */
};
#define SRE_NFA_SA_MAX_LUT 3u
/* build options (tests force every kernel variant) */
#define SRE_NFA_SA_FORCE_MASKED  1u
#define SRE_NFA_SA_FORCE_EVACC   2u
#define SRE_NFA_SA_FORCE_W64     4u
#define SRE_NFA_SA_FORCE_CARRY   8u
#define SRE_NFA_SA_NO_MERGE      16u
#define SRE_NFA_SA_EXPLICIT_ANY  32u
#define SRE_NFA_SA_NO_EVACC      64u
#define SRE_NFA_SA_OFF           128u
#endif

struct sre_nfa_s {
    uint32_t nbits;             /* bits in use (<= 64) */
    uint32_t nslices;           /* ceil(nbits / 8) */
    uint64_t init[3];           /* SRE_DFA_INIT_* -> initial set */
    uint64_t any_bits;          /* the ".*?" ANY thread (pc 1): bit 0, and bit 1 when it has a twin */
    uint64_t match_bits;
    uint64_t accept[256];
    std::vector<uint64_t> follow;   /* [nslices][256] */
    std::vector<uint32_t> bit_pc;   /* [nbits] */
    /* look-ahead assertions */
    uint32_t nassert;               /* 0: none */
    uint32_t assert_slice;          /* the byte of the mask that holds their bits */
    uint8_t  kind[256];             /* per input byte: SRE_NFA_KIND_* | SRE_NFA_LEADING */
    std::vector<uint64_t> expand;   /* [4 prev kinds][4 cur kinds][256 values of the assertion byte] */
    sre_nfa_sa_t *sa;               /* the shift-and form, NULL when the program has none */
};
typedef struct sre_nfa_s sre_nfa_t;

extern "C" {
#else
typedef struct sre_nfa_s sre_nfa_t;
#endif

/* NULL + *why when the program has no bit-parallel form (more than 64 bits, more than
 * 8 look-ahead assertions, or a nullable regex: its first event is at offset 0) */
sre_nfa_t *sre_nfa_build(const sre_program_t *prog, const char **why);
/* same with SRE_NFA_SA_* options for the shift-and form (sre_nfa_build: the environment's
 * SRE_HIP_NFA_SA, default 0) */
sre_nfa_t *sre_nfa_build2(const sre_program_t *prog, unsigned sa_options, const char **why);
void sre_nfa_free(sre_nfa_t *nfa);

#ifdef __cplusplus
}
#endif
#endif
