/*
 * sre_compiler.c — AST -> position-independent bytecode, plus the program dump.
 *
 * The instruction SEQUENCE is part of the matching semantics: thread priority
 * in the VMs is the order in which SPLIT targets are laid out.  It therefore
 * follows the reference's code shapes exactly (reference
 * src/sregex/sre_regex_compiler.c:288-482) and is pinned by the program dumps
 * of all reference test blocks (tests/golden):
 *
 *   ALT      split L1, L2;  L1: <left>;  jmp END;  L2: <right>;  END:
 *   QUEST    split L1, END; L1: <e>; END:            (non-greedy: targets swapped)
 *   STAR     L0: split L1, END; L1: <e>; jmp L0; END:         (ditto)
 *   PLUS     L1: <e>; split L1, END; END:                      (ditto)
 *   PAREN g  save 2g; <e>; save 2g+1
 *   TOPLEVEL <e>; match id
 */
#include "sre_program.h"
#include <stdio.h>
#include <string.h>

/* Largest program the compiler emits.  The parser shares ONE subtree between all
 * copies of a counted quantifier, so nested {n} makes the instruction count
 * grow exponentially in the pattern length (the reference counts it in 64-bit
 * sre_uint_t, sre_regex_compiler.c:244, and then fails its single allocation).
 * Counting saturates here and stops walking once the cap is passed. */
#define SRE_MAX_PROGRAM_LEN  (1u << 24)
/* ... and the WORK is bounded, not only the result: a nest of shared subtrees that emits nothing
 * — (?:(?:(?:(?:){499}){499}){499}){499} — never reaches the length cap and would be walked
 * 499^4 times (round-2 advisor finding).  Node visits count against a cap of their own; past it
 * the count saturates and sre_regex_compile() fails as for a program that is too long. */
#define SRE_MAX_COMPILE_VISITS  (4ull * SRE_MAX_PROGRAM_LEN)

static void
count_insns(const sre_regex_t *r, uint64_t *n, uint64_t *nranges, uint64_t *visits)
{
    if (*n > SRE_MAX_PROGRAM_LEN) return;       /* saturated: end the walk early */
    if (++*visits > SRE_MAX_COMPILE_VISITS) {
        *n = (uint64_t) SRE_MAX_PROGRAM_LEN + 1;
        return;
    }
    switch (r->type) {
    case SRE_RE_ALT:
        *n += 2;
        count_insns(r->left, n, nranges, visits);
        count_insns(r->right, n, nranges, visits);
        break;
    case SRE_RE_CAT:
        count_insns(r->left, n, nranges, visits);
        count_insns(r->right, n, nranges, visits);
        break;
    case SRE_RE_CLASS:
    case SRE_RE_NCLASS:
        *nranges += r->ranges.n;
        *n += 1;
        break;
    case SRE_RE_LIT:
    case SRE_RE_DOT:
    case SRE_RE_ASSERT:
        *n += 1;
        break;
    case SRE_RE_PAREN:
    case SRE_RE_STAR:
        *n += 2;
        count_insns(r->left, n, nranges, visits);
        break;
    case SRE_RE_QUEST:
    case SRE_RE_PLUS:
    case SRE_RE_TOPLEVEL:
        *n += 1;
        count_insns(r->left, n, nranges, visits);
        break;
    case SRE_RE_NIL:
    default:
        break;
    }
}

typedef struct {
    sre_insn_t  *insns;
    sre_range_t *ranges;
    uint32_t     pc, nranges;
} sre_emit_t;

static void
emit(sre_emit_t *e, const sre_regex_t *r)
{
    uint32_t    split, jmp, body;
    sre_insn_t *in;

    switch (r->type) {
    case SRE_RE_ALT:
        split = e->pc++;
        e->insns[split].opcode = SRE_OP_SPLIT;
        e->insns[split].x = e->pc;
        emit(e, r->left);
        jmp = e->pc++;
        e->insns[jmp].opcode = SRE_OP_JMP;
        e->insns[split].y = e->pc;
        emit(e, r->right);
        e->insns[jmp].x = e->pc;
        break;

    case SRE_RE_CAT:
        emit(e, r->left);
        emit(e, r->right);
        break;

    case SRE_RE_LIT:
        in = &e->insns[e->pc++];
        in->opcode = SRE_OP_CHAR;
        in->ch = r->ch;
        break;

    case SRE_RE_DOT:
        e->insns[e->pc++].opcode = SRE_OP_ANY;
        break;

    case SRE_RE_CLASS:
    case SRE_RE_NCLASS:
        in = &e->insns[e->pc++];
        in->opcode = r->type == SRE_RE_CLASS ? SRE_OP_IN : SRE_OP_NOTIN;
        in->x = e->nranges;
        in->nranges = (uint16_t) r->ranges.n;
        if (r->ranges.n) {
            memcpy(&e->ranges[e->nranges], r->ranges.r, r->ranges.n * sizeof(sre_range_t));
        }
        e->nranges += r->ranges.n;
        break;

    case SRE_RE_ASSERT:
        in = &e->insns[e->pc++];
        in->opcode = SRE_OP_ASSERT;
        in->ch = r->assertion;
        break;

    case SRE_RE_PAREN:
        in = &e->insns[e->pc++];
        in->opcode = SRE_OP_SAVE;
        in->arg = (uint32_t) (2 * r->group);
        emit(e, r->left);
        in = &e->insns[e->pc++];
        in->opcode = SRE_OP_SAVE;
        in->arg = (uint32_t) (2 * r->group + 1);
        break;

    case SRE_RE_QUEST:
        split = e->pc++;
        body = e->pc;
        emit(e, r->left);
        in = &e->insns[split];
        in->opcode = SRE_OP_SPLIT;
        in->x = r->greedy ? body : e->pc;
        in->y = r->greedy ? e->pc : body;
        break;

    case SRE_RE_STAR:
        split = e->pc++;
        body = e->pc;
        emit(e, r->left);
        jmp = e->pc++;
        e->insns[jmp].opcode = SRE_OP_JMP;
        e->insns[jmp].x = split;
        in = &e->insns[split];
        in->opcode = SRE_OP_SPLIT;
        in->x = r->greedy ? body : e->pc;
        in->y = r->greedy ? e->pc : body;
        break;

    case SRE_RE_PLUS:
        body = e->pc;
        emit(e, r->left);
        in = &e->insns[e->pc++];
        in->opcode = SRE_OP_SPLIT;
        in->x = r->greedy ? body : e->pc;
        in->y = r->greedy ? e->pc : body;
        break;

    case SRE_RE_TOPLEVEL:
        emit(e, r->left);
        in = &e->insns[e->pc++];
        in->opcode = SRE_OP_MATCH;
        in->arg = (uint32_t) r->regex_id;
        break;

    case SRE_RE_NIL:
    default:
        break;
    }
}

/*
 * Leading-byte analysis (reference sre_regex_compiler.c:123-241): walk the
 * epsilon closure of the program start, skipping the ".*?" ANY at pc 1, and
 * collect the first consuming instructions.  A reachable MATCH makes the regex
 * nullable, a reachable ANY declines; either way there is no leading set.
 * Matching results never depend on this; the device path may use it as a
 * prefilter hint.
 *   returns 0 ok, 1 declined, 2 done (nullable)
 */
static int
leading_walk(sre_program_t *prog, uint32_t pc, uint8_t *seen, uint32_t *out, uint32_t *n)
{
    int rc;

    for (;;) {
        const sre_insn_t *in;
        if (pc >= prog->len || seen[pc] || pc == 1) return 0;
        seen[pc] = 1;
        in = &prog->insns[pc];
        switch (in->opcode) {
        case SRE_OP_SPLIT:
            rc = leading_walk(prog, in->x, seen, out, n);
            if (rc != 0) return rc;
            pc = in->y;
            continue;
        case SRE_OP_JMP:
            pc = in->x;
            continue;
        case SRE_OP_SAVE:
        case SRE_OP_ASSERT:
            pc++;
            continue;
        case SRE_OP_MATCH:
            prog->nullable = 1;
            return 2;
        case SRE_OP_ANY:
            return 1;
        default:
            if (in->opcode == SRE_OP_CHAR) {
                for (uint32_t i = 0; i < *n; i++) {
                    const sre_insn_t *o = &prog->insns[out[i]];
                    if (o->opcode == SRE_OP_CHAR && o->ch == in->ch) return 0;
                }
            }
            out[(*n)++] = pc;
            return 0;
        }
    }
}

SRE_API sre_program_t *
sre_regex_compile(sre_pool_t *pool, sre_regex_t *re)
{
    sre_program_t *prog;
    sre_emit_t     e;
    uint32_t       n, nranges, i;
    uint64_t       n64 = 0, nranges64 = 0, visits = 0;
    uint8_t       *seen;

    count_insns(re, &n64, &nranges64, &visits);
    if (n64 > SRE_MAX_PROGRAM_LEN || nranges64 > 16ull * SRE_MAX_PROGRAM_LEN) return NULL;
    n = (uint32_t) n64;
    nranges = (uint32_t) nranges64;
    prog = sre_pcalloc(pool, sizeof(sre_program_t));
    if (prog == NULL) return NULL;
    prog->pool = pool;
    prog->insns = sre_pcalloc(pool, (size_t) (n + 1) * sizeof(sre_insn_t));
    prog->ranges = sre_pcalloc(pool, (size_t) (nranges + 1) * sizeof(sre_range_t));
    prog->multi_ncaps = sre_pcalloc(pool, re->nregexes * sizeof(uint32_t));
    prog->leading_insns = sre_pcalloc(pool, (size_t) (n + 1) * sizeof(uint32_t));
    seen = sre_pcalloc(pool, n + 1);
    if (!prog->insns || !prog->ranges || !prog->multi_ncaps || !prog->leading_insns || !seen) {
        return NULL;
    }

    e.insns = prog->insns;
    e.ranges = prog->ranges;
    e.pc = 0;
    e.nranges = 0;
    emit(&e, re);
    if (e.pc != n || e.nranges != nranges) return NULL;

    prog->len = n;
    prog->nranges = nranges;
    prog->nregexes = (uint32_t) re->nregexes;
    for (i = 0; i < prog->nregexes; i++) {
        prog->multi_ncaps[i] = (uint32_t) re->multi_ncaps[i];
        prog->nslots += 2 * (prog->multi_ncaps[i] + 1);
    }
    for (i = 0; i < n; i++) {
        switch (prog->insns[i].opcode) {
        case SRE_OP_ASSERT:
            if (prog->insns[i].ch & SRE_ASSERT_LOOKAHEAD) prog->lookahead_asserts++;
            /* fall through */
        case SRE_OP_CHAR: case SRE_OP_IN: case SRE_OP_NOTIN: case SRE_OP_ANY:
        case SRE_OP_MATCH:
            prog->nthreads++;
            break;
        default:
            break;
        }
    }

    prog->leading_byte = -1;
    if (leading_walk(prog, 0, seen, prog->leading_insns, &prog->nleading) != 0
        || prog->nullable)
    {
        prog->nleading = 0;
    }
    if (prog->nleading == 0) {
        prog->leading_insns = NULL;
    } else if (prog->nleading == 1
               && prog->insns[prog->leading_insns[0]].opcode == SRE_OP_CHAR)
    {
        prog->leading_byte = prog->insns[prog->leading_insns[0]].ch;
    }
    prog->dev = NULL;
    return prog;
}

/* one line per instruction, "%2d. <op> ..." (reference sre_vm_bytecode.c:14-128) */
SRE_API void
sre_program_dump(sre_program_t *prog)
{
    for (uint32_t pc = 0; pc < prog->len; pc++) {
        const sre_insn_t *in = &prog->insns[pc];
        const char       *sym;

        printf("%2d. ", (int) pc);
        switch (in->opcode) {
        case SRE_OP_SPLIT:
            printf("split %d, %d", (int) in->x, (int) in->y);
            break;
        case SRE_OP_JMP:
            printf("jmp %d", (int) in->x);
            break;
        case SRE_OP_CHAR:
            printf("char %d", (int) in->ch);
            break;
        case SRE_OP_IN:
        case SRE_OP_NOTIN:
            printf(in->opcode == SRE_OP_IN ? "in" : "notin");
            for (uint32_t i = 0; i < in->nranges; i++) {
                const sre_range_t *r = &prog->ranges[in->x + i];
                printf("%s %d-%d", i ? "," : "", r->from, r->to);
            }
            break;
        case SRE_OP_ANY:
            printf("any");
            break;
        case SRE_OP_MATCH:
            printf("match %d", (int) in->arg);
            break;
        case SRE_OP_SAVE:
            printf("save %d", (int) in->arg);
            break;
        case SRE_OP_ASSERT:
            switch (in->ch) {
            case SRE_ASSERT_BIG_A:   sym = "\\A"; break;
            case SRE_ASSERT_CARET:   sym = "^";   break;
            case SRE_ASSERT_SMALL_Z: sym = "\\z"; break;
            case SRE_ASSERT_BIG_B:   sym = "\\B"; break;
            case SRE_ASSERT_SMALL_B: sym = "\\b"; break;
            case SRE_ASSERT_DOLLAR:  sym = "$";   break;
            default:                 sym = "?";   break;
            }
            printf("assert %s", sym);
            break;
        default:
            printf("unknown");
            break;
        }
        printf("\n");
    }
}
