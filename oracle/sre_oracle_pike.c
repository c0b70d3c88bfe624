/*
 * sre_oracle_pike.c — TEST INFRASTRUCTURE (see sre_oracle.h).
 *
 * CPU restatement of the reference Pike VM.  Each function names the reference
 * lines it follows.  Deliberate differences, none of which can change a result:
 *   - generation tags live in the context, not in the program
 *     (reference keeps them in sre_instruction_t.tag, sre_vm_bytecode.h:51);
 *   - capture vectors are released when a thread is dropped as a duplicate
 *     (the reference leaks that reference, sre_vm_pike.c:770-787 — SURVEY.md
 *     note L; reference counts only steer copy-on-write);
 *   - memory comes from malloc, one context at a time.
 */
#include "sre_oracle.h"
#include "sre_program.h"
#include <stdlib.h>
#include <string.h>

typedef struct cap_s {
    struct cap_s *next;       /* free list */
    unsigned      ref;
    sre_int_t     regex_id;
    sre_int_t     vector[1];  /* nslots */
} cap_t;

typedef struct thr_s {
    struct thr_s *next;
    uint32_t      pc;
    unsigned      seen_word;
    cap_t        *cap;
} thr_t;

typedef struct {
    thr_t   *head, **tail;
    unsigned count;
} tlist_t;

struct sre_oracle_pike_ctx_s {
    sre_program_t *prog;
    unsigned      *tags;          /* [prog->len] generation per instruction */
    unsigned       tag;
    sre_int_t      processed_bytes;
    const sre_char *buffer;
    cap_t         *matched;
    cap_t         *free_caps;
    thr_t         *free_thrs;
    sre_int_t      pending_ovector[2];
    sre_int_t     *ovector;
    size_t         ovecsize;      /* bytes, as given by the caller */
    tlist_t        lists[2];
    tlist_t       *clist, *nlist;
    sre_int_t      last_matched_pos;
    uint32_t      *initial_states;
    unsigned       initial_states_count;
    unsigned       first_buf, seen_start_state, eof, empty_capture,
                   seen_newline, seen_word;
    int            oom;
};

/* ---- captures: reference sre_capture.c:20-85, sre_capture.h:18-22 ---- */

static cap_t *
cap_new(sre_oracle_pike_ctx_t *ctx)
{
    cap_t *c = ctx->free_caps;
    if (c) {
        ctx->free_caps = c->next;
    } else {
        c = malloc(sizeof(cap_t) + ctx->prog->nslots * sizeof(sre_int_t));
        if (c == NULL) {
            ctx->oom = 1;
            return NULL;
        }
    }
    c->next = NULL;
    c->ref = 1;
    c->regex_id = 0;
    return c;
}

static cap_t *
cap_create(sre_oracle_pike_ctx_t *ctx)
{
    cap_t *c = cap_new(ctx);
    if (c) {
        for (uint32_t i = 0; i < ctx->prog->nslots; i++) c->vector[i] = -1;
    }
    return c;
}

static void
cap_release(sre_oracle_pike_ctx_t *ctx, cap_t *c)
{
    if (--c->ref == 0) {
        c->next = ctx->free_caps;
        ctx->free_caps = c;
    }
}

/* copy-on-write store of one slot (sre_capture.c:59-85) */
static cap_t *
cap_update(sre_oracle_pike_ctx_t *ctx, cap_t *c, uint32_t slot, sre_int_t pos)
{
    if (c->ref > 1) {
        cap_t *n = cap_new(ctx);
        if (n == NULL) return NULL;
        memcpy(n->vector, c->vector, ctx->prog->nslots * sizeof(sre_int_t));
        n->regex_id = c->regex_id;
        c->ref--;
        c = n;
    }
    c->vector[slot] = pos;
    return c;
}

/* ---- thread lists: sre_vm_pike.c:738-753, 903-938, 1064-1080 ---- */

static void
list_init(tlist_t *l)
{
    l->head = NULL;
    l->tail = &l->head;
    l->count = 0;
}

static void
thr_free(sre_oracle_pike_ctx_t *ctx, thr_t *t)
{
    t->next = ctx->free_thrs;
    ctx->free_thrs = t;
}

static void
list_clear(sre_oracle_pike_ctx_t *ctx, tlist_t *l)
{
    while (l->head) {
        thr_t *t = l->head;
        l->head = t->next;
        cap_release(ctx, t->cap);
        thr_free(ctx, t);
    }
    list_init(l);
}

SRE_API sre_oracle_pike_ctx_t *
sre_oracle_pike_create_ctx(sre_program_t *prog, sre_int_t *ovector, size_t ovecsize)
{
    /* sre_vm_pike.c:94-145 */
    sre_oracle_pike_ctx_t *ctx = calloc(1, sizeof(*ctx));
    if (ctx == NULL) return NULL;
    ctx->prog = prog;
    ctx->tags = calloc(prog->len + 1, sizeof(unsigned));
    ctx->initial_states = calloc(prog->len + 1, sizeof(uint32_t));
    if (ctx->tags == NULL || ctx->initial_states == NULL) {
        free(ctx->tags);
        free(ctx->initial_states);
        free(ctx);
        return NULL;
    }
    ctx->tag = 1;
    ctx->ovector = ovector;
    ctx->ovecsize = ovecsize;
    list_init(&ctx->lists[0]);
    list_init(&ctx->lists[1]);
    ctx->clist = &ctx->lists[0];
    ctx->nlist = &ctx->lists[1];
    ctx->first_buf = 1;
    ctx->last_matched_pos = -1;
    return ctx;
}

SRE_API void
sre_oracle_pike_free(sre_oracle_pike_ctx_t *ctx)
{
    if (ctx == NULL) return;
    list_clear(ctx, ctx->clist);
    list_clear(ctx, ctx->nlist);
    if (ctx->matched) cap_release(ctx, ctx->matched);
    while (ctx->free_caps) {
        cap_t *c = ctx->free_caps;
        ctx->free_caps = c->next;
        free(c);
    }
    while (ctx->free_thrs) {
        thr_t *t = ctx->free_thrs;
        ctx->free_thrs = t->next;
        free(t);
    }
    free(ctx->tags);
    free(ctx->initial_states);
    free(ctx);
}

/*
 * Recursive epsilon closure: sre_vm_pike.c:756-942.  Consumes one reference of
 * `cap`.  `pcap` non-NULL means "called from the byte loop": reaching MATCH then
 * returns SRE_DONE with the capture handed back through *pcap (:889-899).
 */
static sre_int_t
add_thread(sre_oracle_pike_ctx_t *ctx, tlist_t *l, uint32_t pc, cap_t *cap,
    sre_int_t pos, cap_t **pcap)
{
    const sre_insn_t *in = &ctx->prog->insns[pc];
    unsigned          seen_word = 0;
    sre_int_t         rc;
    thr_t            *t;

    if (ctx->tags[pc] == ctx->tag) {
        /* :770-787 — already visited in this generation; a SPLIT whose second
         * branch has not been visited yet is re-descended into */
        if (in->opcode == SRE_OP_SPLIT && ctx->tags[in->y] != ctx->tag) {
            if (pc == 0) ctx->seen_start_state = 1;
            return add_thread(ctx, l, in->y, cap, pos, pcap);
        }
        cap_release(ctx, cap);
        return SRE_OK;
    }
    ctx->tags[pc] = ctx->tag;

    switch (in->opcode) {
    case SRE_OP_JMP:                                           /* :795 */
        return add_thread(ctx, l, in->x, cap, pos, pcap);

    case SRE_OP_SPLIT:                                         /* :798-812 */
        if (pc == 0) ctx->seen_start_state = 1;
        cap->ref++;
        rc = add_thread(ctx, l, in->x, cap, pos, pcap);
        if (rc != SRE_OK) {
            /* MATCH reached in the first branch: the second is abandoned */
            cap_release(ctx, cap);
            return rc;
        }
        return add_thread(ctx, l, in->y, cap, pos, pcap);

    case SRE_OP_SAVE:                                          /* :814-837 */
        cap = cap_update(ctx, cap, in->arg, ctx->processed_bytes + pos);
        if (cap == NULL) return SRE_ERROR;
        return add_thread(ctx, l, pc + 1, cap, pos, pcap);

    case SRE_OP_ASSERT:                                        /* :839-887 */
        switch (in->ch) {
        case SRE_ASSERT_BIG_A:
            if (pos || ctx->processed_bytes) {
                cap_release(ctx, cap);
                return SRE_OK;
            }
            return add_thread(ctx, l, pc + 1, cap, pos, pcap);

        case SRE_ASSERT_CARET:
            if (pos == 0) {
                if (ctx->processed_bytes && !ctx->seen_newline) {
                    cap_release(ctx, cap);
                    return SRE_OK;
                }
            } else if (ctx->buffer[pos - 1] != '\n') {
                cap_release(ctx, cap);
                return SRE_OK;
            }
            return add_thread(ctx, l, pc + 1, cap, pos, pcap);

        case SRE_ASSERT_SMALL_B:
        case SRE_ASSERT_BIG_B:
            seen_word = pos == 0 ? 0 : (unsigned) sre_isword(ctx->buffer[pos - 1]);
            break;          /* queue the thread: evaluated against the next byte */

        default:
            break;          /* $ and \z: postponed look-ahead */
        }
        break;

    case SRE_OP_MATCH:                                         /* :889-899 */
        ctx->last_matched_pos = cap->vector[1];
        cap->regex_id = (sre_int_t) in->arg;
        if (pcap) {
            *pcap = cap;
            return SRE_DONE;
        }
        break;

    default:
        break;
    }

    /* Test-harness guard, NOT reference behaviour: on programs whose assertion
     * splice cycles (a look-ahead assertion inside an empty loop, e.g.
     * (\n?|^$)+?) the reference lists threads without bound until it runs out
     * of memory or crashes (verified with oracle/_ref/sregex-cli).  The exec
     * stops with SRE_ERROR here so that tests can skip such a case. */
    if (l->count > 64u * (ctx->prog->len + 16u)) {
        ctx->oom = 1;
        cap_release(ctx, cap);
        return SRE_ERROR;
    }

    /* :903-938 append to the tail */
    t = ctx->free_thrs;
    if (t) {
        ctx->free_thrs = t->next;
    } else {
        t = malloc(sizeof(thr_t));
        if (t == NULL) {
            ctx->oom = 1;
            cap_release(ctx, cap);
            return SRE_ERROR;
        }
    }
    t->pc = pc;
    t->cap = cap;
    t->seen_word = seen_word;
    t->next = NULL;
    *l->tail = t;
    l->tail = &t->next;
    l->count++;
    return SRE_OK;
}

/* sre_vm_pike.c:992-1061 */
static const sre_char *
find_first_byte(const sre_program_t *prog, const sre_char *pos, const sre_char *last)
{
    if (prog->leading_byte != -1) {
        const sre_char *p = memchr(pos, prog->leading_byte, (size_t) (last - pos));
        return p ? p : last;
    }
    for (; pos != last; pos++) {
        for (uint32_t i = 0; i < prog->nleading; i++) {
            const sre_insn_t *in = &prog->insns[prog->leading_insns[i]];
            int               hit;
            switch (in->opcode) {
            case SRE_OP_CHAR:
                if (*pos == in->ch) return pos;
                break;
            case SRE_OP_IN:
            case SRE_OP_NOTIN:
                hit = sre_in_ranges(&prog->ranges[in->x], in->nranges, *pos);
                if (hit == (in->opcode == SRE_OP_IN)) return pos;
                break;
            default:
                break;
            }
        }
    }
    return pos;
}

/* sre_vm_pike.c:945-989 */
static sre_int_t
prepare_matched_captures(sre_oracle_pike_ctx_t *ctx, cap_t *matched,
    sre_int_t *ovector, int complete)
{
    const sre_program_t *prog = ctx->prog;
    size_t               ofs = 0, len;
    sre_int_t            i;

    if (matched->regex_id >= (sre_int_t) prog->nregexes) return SRE_ERROR;
    for (i = 0; i < matched->regex_id; i++) ofs += prog->multi_ncaps[i] + 1;
    ofs *= 2;
    len = complete ? 2 * (prog->multi_ncaps[i] + 1) * sizeof(sre_int_t)
                   : 2 * sizeof(sre_int_t);
    memcpy(ovector, &matched->vector[ofs], len);
    if (complete && ctx->ovecsize > len) {
        memset((char *) ovector + len, -1, ctx->ovecsize - len);
    }
    return SRE_OK;
}

/* sre_vm_pike.c:692-735 — note the end offset reads vector[j + 1] WITHOUT the
 * per-regex offset (:721); kept as is */
static void
prepare_temp_captures(sre_oracle_pike_ctx_t *ctx)
{
    const sre_program_t *prog = ctx->prog;

    ctx->ovector[0] = -1;
    ctx->ovector[1] = -1;
    for (thr_t *t = ctx->clist->head; t; t = t->next) {
        size_t ofs = 0;
        for (uint32_t i = 0; i < prog->nregexes; i++) {
            sre_int_t a = ctx->ovector[0], b = t->cap->vector[ofs];
            if (b != -1 && (a == -1 || b < a)) ctx->ovector[0] = b;
            a = ctx->ovector[1];
            b = t->cap->vector[1];
            if (b != -1 && (a == -1 || b > a)) ctx->ovector[1] = b;
            ofs += 2 * (prog->multi_ncaps[i] + 1);
        }
    }
}

/* does `c` let instruction `in` consume a byte?  (:329-448) */
static int
consumes(const sre_program_t *prog, const sre_insn_t *in, unsigned c)
{
    switch (in->opcode) {
    case SRE_OP_CHAR:  return c == in->ch;
    case SRE_OP_ANY:   return 1;
    case SRE_OP_IN:    return sre_in_ranges(&prog->ranges[in->x], in->nranges, c);
    case SRE_OP_NOTIN: return !sre_in_ranges(&prog->ranges[in->x], in->nranges, c);
    default:           return 0;
    }
}

SRE_API sre_int_t
sre_oracle_pike_exec(sre_oracle_pike_ctx_t *ctx, const sre_char *input, size_t size,
    unsigned eof, sre_int_t **pending_matched)
{
    sre_program_t  *prog = ctx->prog;
    const sre_char *sp, *last, *p;
    tlist_t        *clist = ctx->clist, *nlist = ctx->nlist, *tmp, splice;
    cap_t          *cap, *matched = ctx->matched;
    thr_t          *t;
    sre_int_t       rc;
    unsigned        i, hold, seen_word;

    if (ctx->eof) return SRE_ERROR;                            /* :165-168 */

    ctx->buffer = input;
    ctx->last_matched_pos = -1;

    if (ctx->empty_capture) {                                  /* :179-196 */
        ctx->empty_capture = 0;
        if (size == 0) {
            if (eof) {
                ctx->eof = 1;
                return SRE_DECLINED;
            }
            return SRE_AGAIN;
        }
        sp = input + 1;
    } else {
        sp = input;
    }
    last = input + size;

    if (ctx->first_buf) {                                      /* :202-233 */
        ctx->first_buf = 0;
        cap = cap_create(ctx);
        if (cap == NULL) return SRE_ERROR;
        ctx->tag++;
        if (add_thread(ctx, clist, 0, cap, (sre_int_t) (sp - input), NULL) != SRE_OK) {
            return SRE_ERROR;
        }
        ctx->initial_states_count = clist->count;
        /* all but the last thread, which is always the ".*?" ANY (:226-229) */
        for (i = 0, t = clist->head; t && t->next; i++, t = t->next) {
            ctx->initial_states[i] = t->pc;
        }
    }

    for (; sp < last || (eof && sp == last); sp++) {           /* :235 */
        if (clist->head == NULL) break;

        /* :256-309 leading-byte skip while the list equals the initial closure */
        if (prog->leading_insns && ctx->seen_start_state) {
            ctx->seen_start_state = 0;
            if (sp == last || clist->count != ctx->initial_states_count) goto run;
            for (i = 0, t = clist->head; t && t->next; i++, t = t->next) {
                if (t->pc != ctx->initial_states[i]) goto run;
            }
            p = find_first_byte(prog, sp, last);
            if (p > sp) {
                sp = p;
                list_clear(ctx, clist);
                cap = cap_create(ctx);
                if (cap == NULL) return SRE_ERROR;
                ctx->tag++;
                if (add_thread(ctx, clist, 0, cap, (sre_int_t) (sp - input), NULL) != SRE_OK) {
                    return SRE_ERROR;
                }
                if (sp == last) break;
            }
        }
run:
        ctx->tag++;                                            /* :312 */

        while (clist->head) {                                  /* :314 */
            const sre_insn_t *in;

            t = clist->head;
            clist->head = t->next;
            if (clist->head == NULL) clist->tail = &clist->head;
            clist->count--;
            in = &prog->insns[t->pc];
            cap = t->cap;

            switch (in->opcode) {
            case SRE_OP_IN: case SRE_OP_NOTIN: case SRE_OP_CHAR: case SRE_OP_ANY:
                if (sp == last || !consumes(prog, in, *sp)) {  /* :329-448 */
                    cap_release(ctx, cap);
                    break;
                }
                rc = add_thread(ctx, nlist, t->pc + 1, cap, (sre_int_t) (sp - input + 1), &cap);
                if (rc == SRE_DONE) goto matched;
                if (rc != SRE_OK) return SRE_ERROR;
                break;

            case SRE_OP_ASSERT:                                /* :450-528 */
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
                    seen_word = (t->seen_word || (sp == input && ctx->seen_word));
                    hold = seen_word ^ (unsigned) (sp != last && sre_isword(*sp));
                    if (in->ch == SRE_ASSERT_BIG_B) hold = !hold;
                    break;
                default:
                    break;
                }
                if (!hold) {
                    /* the reference drops the thread without releasing (:454-487) */
                    cap_release(ctx, cap);
                    break;
                }
                /* :506-526 — closure at the SAME position, de-duplicated against
                 * the generation of the current list, spliced at its head */
                ctx->tag--;
                list_init(&splice);
                rc = add_thread(ctx, &splice, t->pc + 1, cap, (sre_int_t) (sp - input), NULL);
                if (rc != SRE_OK) return SRE_ERROR;
                if (splice.head) {
                    *splice.tail = clist->head;
                    if (clist->head == NULL) clist->tail = splice.tail;
                    clist->head = splice.head;
                    clist->count += splice.count;
                }
                ctx->tag++;
                break;

            case SRE_OP_MATCH:                                 /* :530-553 */
                ctx->last_matched_pos = cap->vector[1];
                cap->regex_id = (sre_int_t) in->arg;
matched:
                if (matched) cap_release(ctx, matched);
                matched = cap;
                thr_free(ctx, t);
                list_clear(ctx, clist);      /* everything of lower priority */
                goto step_done;

            default:
                break;
            }
            thr_free(ctx, t);
        }

step_done:
        tmp = clist;                                           /* :569-580 */
        clist = nlist;
        nlist = tmp;
        if (nlist->head) list_clear(ctx, nlist);
        if (sp == last) break;
    }

    if (ctx->last_matched_pos >= 0) {                          /* :586-601 */
        p = input + ctx->last_matched_pos - ctx->processed_bytes;
        if (p > input) {
            ctx->seen_newline = (p[-1] == '\n');
            ctx->seen_word = (unsigned) sre_isword(p[-1]);
        }
        ctx->last_matched_pos = -1;
    }

    ctx->clist = clist;
    ctx->nlist = nlist;

    if (matched) {                                             /* :607-658 */
        if (eof || clist->head == NULL) {
            if (prepare_matched_captures(ctx, matched, ctx->ovector, 1) != SRE_OK) {
                return SRE_ERROR;
            }
            if (clist->head) {
                /* the reference recycles the nodes only (:616-622) */
                list_clear(ctx, clist);
                ctx->eof = 1;
            }
            ctx->processed_bytes = ctx->ovector[1];
            ctx->empty_capture = (ctx->ovector[0] == ctx->ovector[1]);
            ctx->matched = NULL;
            ctx->first_buf = 1;
            rc = matched->regex_id;
            cap_release(ctx, matched);
            return rc;
        }
        if (pending_matched) {
            *pending_matched = ctx->pending_ovector;
            if (prepare_matched_captures(ctx, matched, *pending_matched, 0) != SRE_OK) {
                return SRE_ERROR;
            }
        }
    } else {
        if (eof) {                                             /* :660-671 */
            ctx->eof = 1;
            ctx->matched = NULL;
            return SRE_DECLINED;
        }
        if (pending_matched) *pending_matched = NULL;
    }

    ctx->processed_bytes += (sre_int_t) (sp - input);          /* :673-688 */
    ctx->matched = matched;
    prepare_temp_captures(ctx);
    return SRE_AGAIN;
}

SRE_API sre_int_t
sre_oracle_pike_count(sre_program_t *prog, const sre_char *input, size_t size,
    sre_int_t *spans, size_t nov, size_t max_spans, sre_int_t *final_rc)
{
    sre_uint_t  maxcaps = 0;
    sre_int_t  *ov, count = 0, rc;
    size_t      off = 0, n;
    sre_oracle_pike_ctx_t *ctx;

    for (uint32_t i = 0; i < prog->nregexes; i++) {
        if (prog->multi_ncaps[i] > maxcaps) maxcaps = prog->multi_ncaps[i];
    }
    n = 2 * (maxcaps + 1);
    ov = malloc(n * sizeof(sre_int_t));
    ctx = sre_oracle_pike_create_ctx(prog, ov, n * sizeof(sre_int_t));
    if (ov == NULL || ctx == NULL) return SRE_ERROR;
    for (;;) {
        rc = sre_oracle_pike_exec(ctx, input + off, size - off, 1, NULL);
        if (rc < 0) break;
        if (spans && (size_t) count < max_spans) {
            sre_int_t *rec = spans + (size_t) count * (nov + 1);
            rec[0] = rc;
            for (size_t k = 0; k < nov; k++) rec[1 + k] = k < n ? ov[k] : -1;
        }
        count++;
        off = (size_t) ov[1];
    }
    sre_oracle_pike_free(ctx);
    free(ov);
    /* the iteration ends with SRE_DECLINED, or with SRE_ERROR when the previous
     * call returned its match with threads still listed (sre_vm_pike.c:616-622) */
    if (final_rc) *final_rc = rc;
    return count;
}
