/*
 * sre_program.h — internal data model shared by the host front end, the device
 * blob builder and (read-only) by the test oracle.
 *
 * Unlike the reference's pointer-linked sre_instruction_t (40 B, absolute x/y/
 * ranges pointers and a per-instruction generation tag written by the VMs —
 * reference src/sregex/sre_vm_bytecode.h:45-61), the program here is
 * position-independent: instructions address each other by index and carry no
 * mutable state, so one program can be staged to LDS as is and shared by any
 * number of concurrent streams.
 */
#ifndef SRE_PROGRAM_H
#define SRE_PROGRAM_H

#include <sregex/sregex.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- arena pool (sre_pool.c) ---- */
typedef void (*sre_pool_cleanup_pt)(void *data);

SRE_NOAPI void *sre_palloc(sre_pool_t *pool, size_t size);    /* 16-B aligned */
SRE_NOAPI void *sre_pcalloc(sre_pool_t *pool, size_t size);
/* run `handler(data)` when the pool is destroyed (not on reset), LIFO. */
SRE_NOAPI int sre_pool_add_cleanup(sre_pool_t *pool, sre_pool_cleanup_pt handler,
    void *data);

/* ---- AST (sre_parser.c) ---- */
typedef enum {
    SRE_RE_NIL = 0, SRE_RE_ALT, SRE_RE_CAT, SRE_RE_LIT, SRE_RE_DOT, SRE_RE_PAREN,
    SRE_RE_QUEST, SRE_RE_STAR, SRE_RE_PLUS, SRE_RE_CLASS, SRE_RE_NCLASS,
    SRE_RE_ASSERT, SRE_RE_TOPLEVEL
} sre_re_type_t;

/* assertion bits: same values as the reference (src/sregex/sre_regex.h:35-53)
 * because `assert` operands are visible in program dumps' semantics only by
 * symbol, but tests of the IR compare them numerically */
enum {
    SRE_ASSERT_SMALL_Z = 0x01,   /* \z */
    SRE_ASSERT_DOLLAR  = 0x02,   /* $  */
    SRE_ASSERT_BIG_B   = 0x04,   /* \B */
    SRE_ASSERT_SMALL_B = 0x08,   /* \b */
    SRE_ASSERT_BIG_A   = 0x10,   /* \A */
    SRE_ASSERT_CARET   = 0x20,   /* ^  */
    SRE_ASSERT_LOOKAHEAD = 0x0f  /* evaluated against the NEXT byte */
};

typedef struct { uint8_t from, to; } sre_range_t;

typedef struct {
    sre_range_t *r;
    uint32_t     n, cap;
} sre_rangevec_t;

struct sre_regex_s {
    sre_re_type_t        type;
    struct sre_regex_s  *left, *right;
    /* per type payload */
    uint8_t              ch;        /* LIT */
    uint8_t              greedy;    /* QUEST/STAR/PLUS */
    uint8_t              assertion; /* ASSERT */
    sre_uint_t           group;     /* PAREN */
    sre_int_t            regex_id;  /* TOPLEVEL */
    sre_rangevec_t       ranges;    /* CLASS / NCLASS */
    /* root only */
    sre_uint_t           nregexes;
    sre_uint_t          *multi_ncaps;
};

/* ---- bytecode ---- */
/* opcode numbers match the reference (sre_vm_bytecode.h:18-28) */
enum {
    SRE_OP_CHAR = 1, SRE_OP_MATCH = 2, SRE_OP_JMP = 3, SRE_OP_SPLIT = 4,
    SRE_OP_ANY = 5, SRE_OP_SAVE = 6, SRE_OP_IN = 7, SRE_OP_NOTIN = 8,
    SRE_OP_ASSERT = 9
};

typedef struct {
    uint8_t   opcode;
    uint8_t   ch;          /* CHAR: byte; ASSERT: assertion bit */
    uint16_t  nranges;     /* IN / NOTIN */
    uint32_t  x;           /* SPLIT/JMP target; IN/NOTIN: first range index */
    uint32_t  y;           /* SPLIT second target */
    uint32_t  arg;         /* SAVE: slot; MATCH: regex id */
} sre_insn_t;              /* 16 bytes */

struct sre_hip_program_s;   /* device-side images, owned by sre_hip_*.cpp */

struct sre_program_s {
    sre_pool_t   *pool;
    sre_insn_t   *insns;
    uint32_t      len;
    sre_range_t  *ranges;
    uint32_t      nranges;
    uint32_t      nslots;        /* 2 * sum(ncaps_i + 1): internal capture slots */
    uint32_t      nregexes;
    uint32_t     *multi_ncaps;   /* [nregexes] */
    uint32_t      nthreads;      /* list-able instructions (CHAR/IN/NOTIN/ANY/ASSERT/MATCH) */
    uint32_t      lookahead_asserts; /* count of $ \z \b \B instructions */
    /* leading-byte analysis (reference sre_regex_compiler.c:123-241); the
     * device path uses it as a prefilter hint only, results never depend on it */
    uint8_t       nullable;
    int           leading_byte;     /* -1 if none */
    uint32_t     *leading_insns;    /* indices of leading CHAR/IN/NOTIN, or NULL */
    uint32_t      nleading;
    struct sre_hip_program_s *dev;  /* lazily built by the HIP layer */
};

static inline int sre_isword(unsigned c) {
    return (c >= '0' && c <= '9') || (c >= 'A' && c <= 'Z')
           || (c >= 'a' && c <= 'z') || c == '_';
}

/* range test shared by host tools; inclusive, unsigned (reference
 * sre_vm_pike.c:336-346) */
static inline int sre_in_ranges(const sre_range_t *r, unsigned n, unsigned c) {
    for (unsigned i = 0; i < n; i++) {
        if (c >= r[i].from && c <= r[i].to) return 1;
    }
    return 0;
}

#ifdef __cplusplus
}
#endif
#endif
