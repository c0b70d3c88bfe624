/*
 * sre_oracle_thompson.c — TEST INFRASTRUCTURE (see sre_oracle.h).
 *
 * CPU restatement of the reference Thompson VM (match / no match, no captures):
 * reference src/sregex/sre_vm_thompson.c:25-60 (ctx), :63-270 (exec),
 * :273-345 (closure).  Generation tags live in the context instead of the
 * program; otherwise the control flow is followed statement by statement,
 * including the chunk-local view of \A and ^ (:302-317: "start of input" means
 * start of the CURRENT chunk).
 */
#include "sre_oracle.h"
#include "sre_program.h"
#include <stdlib.h>

typedef struct {
    uint32_t pc;
    uint8_t  seen_word;
} tthr_t;

typedef struct {
    unsigned count;
    tthr_t  *threads;      /* capacity prog->len */
} ttlist_t;

struct sre_oracle_thompson_ctx_s {
    sre_program_t  *prog;
    const sre_char *buffer;
    unsigned       *tags;
    unsigned        tag;
    ttlist_t        lists[2];
    ttlist_t       *clist, *nlist;
    unsigned        first_buf;
    unsigned        overflow;   /* test-harness guard, NOT reference behaviour: on programs whose
                                   assertion splice cycles (a look-ahead assertion inside an empty
                                   loop) the reference appends to its list without bound and writes
                                   past the array (sre_vm_thompson.c:227-231 with :17-27 sizing);
                                   here the exec stops with SRE_ERROR so that tests can skip the case */
};

SRE_API sre_oracle_thompson_ctx_t *
sre_oracle_thompson_create_ctx(sre_program_t *prog)
{
    sre_oracle_thompson_ctx_t *ctx = calloc(1, sizeof(*ctx));
    if (ctx == NULL) return NULL;
    ctx->prog = prog;
    ctx->tags = calloc(prog->len + 1, sizeof(unsigned));
    ctx->lists[0].threads = calloc(prog->len + 1, sizeof(tthr_t));
    ctx->lists[1].threads = calloc(prog->len + 1, sizeof(tthr_t));
    if (!ctx->tags || !ctx->lists[0].threads || !ctx->lists[1].threads) {
        sre_oracle_thompson_free(ctx);
        return NULL;
    }
    ctx->clist = &ctx->lists[0];
    ctx->nlist = &ctx->lists[1];
    ctx->tag = 1;               /* :56 */
    ctx->first_buf = 1;
    return ctx;
}

SRE_API void
sre_oracle_thompson_free(sre_oracle_thompson_ctx_t *ctx)
{
    if (ctx == NULL) return;
    free(ctx->tags);
    free(ctx->lists[0].threads);
    free(ctx->lists[1].threads);
    free(ctx);
}

/* sre_vm_thompson.c:273-345 */
static void
add_thread(sre_oracle_thompson_ctx_t *ctx, ttlist_t *l, uint32_t pc, const sre_char *sp)
{
    const sre_insn_t *in = &ctx->prog->insns[pc];
    uint8_t           seen_word = 0;
    tthr_t           *t;

    if (ctx->tags[pc] == ctx->tag) return;      /* plain de-dup, no re-descent */
    ctx->tags[pc] = ctx->tag;

    switch (in->opcode) {
    case SRE_OP_JMP:
        add_thread(ctx, l, in->x, sp);
        return;
    case SRE_OP_SPLIT:
        add_thread(ctx, l, in->x, sp);
        add_thread(ctx, l, in->y, sp);
        return;
    case SRE_OP_SAVE:
        add_thread(ctx, l, pc + 1, sp);
        return;
    case SRE_OP_ASSERT:
        switch (in->ch) {
        case SRE_ASSERT_BIG_A:
            if (sp != ctx->buffer) return;
            add_thread(ctx, l, pc + 1, sp);
            return;
        case SRE_ASSERT_CARET:
            if (sp != ctx->buffer && sp[-1] != '\n') return;
            add_thread(ctx, l, pc + 1, sp);
            return;
        case SRE_ASSERT_SMALL_B:
        case SRE_ASSERT_BIG_B:
            seen_word = (uint8_t) (sp != ctx->buffer && sre_isword(sp[-1]));
            break;
        default:
            break;      /* $ \z wait in the list for the next byte */
        }
        break;
    default:
        break;
    }

    if (l->count > ctx->prog->len) {
        ctx->overflow = 1;
        return;
    }
    t = &l->threads[l->count++];
    t->pc = pc;
    t->seen_word = seen_word;
}

SRE_API sre_int_t
sre_oracle_thompson_exec(sre_oracle_thompson_ctx_t *ctx, const sre_char *input,
    size_t size, unsigned eof)
{
    sre_program_t  *prog = ctx->prog;
    const sre_char *sp, *last;
    ttlist_t       *clist = ctx->clist, *nlist = ctx->nlist, *tmp;

    ctx->buffer = input;
    if (ctx->first_buf) {                                       /* :81-84 */
        ctx->first_buf = 0;
        add_thread(ctx, clist, 0, input);
    }
    last = input + size;

    for (sp = input; sp < last || (eof && sp == last); sp++) {  /* :88 */
        if (clist->count == 0) break;
        ctx->tag++;

        /* the list may grow while it is walked: a holding look-ahead assertion
         * appends its continuation to the END of the current list (:227-231) */
        for (unsigned i = 0; i < clist->count && !ctx->overflow; i++) {
            tthr_t           *t = &clist->threads[i];
            const sre_insn_t *in = &prog->insns[t->pc];
            unsigned          hold, w;

            switch (in->opcode) {
            case SRE_OP_IN:
            case SRE_OP_NOTIN:
                if (sp == last) break;
                if (sre_in_ranges(&prog->ranges[in->x], in->nranges, *sp)
                    != (in->opcode == SRE_OP_IN))
                {
                    break;
                }
                add_thread(ctx, nlist, t->pc + 1, sp + 1);
                break;
            case SRE_OP_CHAR:
                if (sp == last || *sp != in->ch) break;
                add_thread(ctx, nlist, t->pc + 1, sp + 1);
                break;
            case SRE_OP_ANY:
                if (sp == last) break;
                add_thread(ctx, nlist, t->pc + 1, sp + 1);
                break;
            case SRE_OP_ASSERT:
                hold = 0;
                switch (in->ch) {
                case SRE_ASSERT_SMALL_Z:
                    hold = (sp == last);
                    break;
                case SRE_ASSERT_DOLLAR:
                    hold = (sp == last || *sp == '\n');
                    break;
                case SRE_ASSERT_BIG_B:
                case SRE_ASSERT_SMALL_B:
                    w = (unsigned) (sp != last && sre_isword(*sp));
                    hold = t->seen_word ^ w;
                    if (in->ch == SRE_ASSERT_BIG_B) hold = !hold;
                    break;
                default:
                    break;
                }
                if (hold) {
                    ctx->tag--;
                    add_thread(ctx, clist, t->pc + 1, sp);
                    ctx->tag++;
                }
                break;
            case SRE_OP_MATCH:                                  /* :233-235 */
                return SRE_OK;
            default:
                break;
            }
        }

        if (ctx->overflow) return SRE_ERROR;        /* harness guard, see the struct */
        tmp = clist;
        clist = nlist;
        nlist = tmp;
        nlist->count = 0;
        if (sp == last) break;
    }

    ctx->clist = clist;
    ctx->nlist = nlist;
    return eof ? SRE_DECLINED : SRE_AGAIN;
}
