/*
 * sre_hip_vm.hip — the EXACT streaming VM kernels (gfx950).
 *
 * One lane owns one stream: it carries that stream's whole VM state (thread
 * lists, per-thread capture vectors, generation tags) in HBM between exec()
 * calls, so the chunked C API (AGAIN / pending matches / byte-at-a-time
 * feeding) behaves exactly like the reference.  These kernels take EVERY
 * program — look-ahead assertions, thousands of instructions, any number of
 * threads — and are the semantic anchor of the device path; the throughput
 * path for large streams is the table-driven scanner in sre_hip_scan.hip.
 *
 * What is mirrored (reference file:line):
 *   pike_exec        sre_vm_pike.c:148-689     byte loop, MATCH cut-off, re-arm
 *   pike_closure     sre_vm_pike.c:756-942     epsilon closure incl. the SPLIT
 *                                              re-descent quirk (:774-784)
 *   thompson_exec    sre_vm_thompson.c:63-270
 *   thompson_closure sre_vm_thompson.c:273-345
 * What is different by design:
 *   - no recursion: the closure runs on an explicit stack whose RESTORE records
 *     undo SAVEs, so capture vectors are plain per-thread values (no
 *     ref-counted copy-on-write objects, sre_capture.c:20-85, and none of the
 *     reference's capture leak);
 *   - generation tags are per stream, the program image is read-only.
 * The leading-byte skip (sre_vm_pike.c:256-309, 992-1061) IS replicated: its
 * "is this the initial state" test ignores the last thread of the list
 * (:266-273), so after a match it can re-seed the search and replace the match
 * — observable behaviour of the reference, pinned by tests/golden/findall.jsonl.
 */
#include <hip/hip_runtime.h>
#include "sre_hip_common.h"
#include "sre_hip_vm.h"

/* Entries of a thread list.  Without look-ahead assertions a list holds every
 * list-able instruction at most once; an assertion splice may list some again
 * (the marks it de-duplicates against have been overwritten meanwhile), so
 * there is generous room — and a hard stop, see Pike::node_new. */
#define SRE_LIST_CAP(n) (8u * ((n) + 1u) + 32u)

namespace {

enum : uint8_t {
    OP_CHAR = 1, OP_MATCH = 2, OP_JMP = 3, OP_SPLIT = 4, OP_ANY = 5, OP_SAVE = 6,
    OP_IN = 7, OP_NOTIN = 8, OP_ASSERT = 9
};
enum : uint8_t {
    AS_SMALL_Z = 0x01, AS_DOLLAR = 0x02, AS_BIG_B = 0x04, AS_SMALL_B = 0x08,
    AS_BIG_A = 0x10, AS_CARET = 0x20
};
enum : int64_t { RC_OK = 0, RC_ERROR = -1, RC_AGAIN = -2, RC_DONE = -4, RC_DECLINED = -5 };

struct Prog {
    const sre_dev_prog_hdr_t *h;
    const sre_dev_insn_t     *insns;
    const uint32_t           *classes;
    const uint32_t           *multi_ncaps;
    const uint32_t           *leading;
};

__device__ inline Prog prog_view(const uint8_t *blob)
{
    Prog p;
    p.h = reinterpret_cast<const sre_dev_prog_hdr_t *>(blob);
    p.insns = reinterpret_cast<const sre_dev_insn_t *>(blob + sre_dev_prog_insns_off());
    p.classes = reinterpret_cast<const uint32_t *>(blob + sre_dev_prog_classes_off(p.h->len));
    p.multi_ncaps = reinterpret_cast<const uint32_t *>(
        blob + sre_dev_prog_ncaps_off(p.h->len, p.h->nclasses));
    p.leading = reinterpret_cast<const uint32_t *>(
        blob + sre_dev_prog_leading_off(p.h->len, p.h->nclasses, p.h->nregexes));
    return p;
}

__device__ inline bool is_word(unsigned c)
{
    return (c - '0' < 10u) || (c - 'A' < 26u) || (c - 'a' < 26u) || c == '_';
}

__device__ inline bool consumes(const Prog &P, const sre_dev_insn_t &in, unsigned c)
{
    switch (in.opcode) {
    case OP_CHAR:  return c == in.ch;
    case OP_ANY:   return true;
    case OP_IN:
    case OP_NOTIN: return (P.classes[in.cls * 8u + (c >> 5)] >> (c & 31u)) & 1u;
    default:       return false;
    }
}

/* chunk bytes: from HBM, or from the kernel argument for tiny chunks */
struct Chunk {
    const uint8_t *p;
    uint64_t       inl;
    __device__ inline unsigned at(int64_t i) const
    {
        return p ? p[i] : (unsigned) ((inl >> (8 * i)) & 0xff);
    }
};

/* ======================================================================= Pike */

constexpr uint32_t PIKE_MAGIC = 0x50494b45u;

struct PikeHdr {
    int64_t  processed_bytes;
    int64_t  last_matched_pos;
    int64_t  matched_regex_id;
    uint32_t magic, tag;
    uint32_t first_buf, eof, empty_capture, seen_newline, seen_word, has_matched;
    uint32_t cur;                 /* which of the two lists is "current" */
    int32_t  head[2], tail[2];
    uint32_t used[2], count[2];
    uint32_t seen_start_state, initial_count;
    uint32_t overflow;            /* a thread list outgrew its nodes: see node_new */
};

struct Node {           /* followed by int64_t cap[nslots] */
    uint32_t pc;
    uint32_t seen_word;
    int32_t  next;
    int32_t  pad;
};

struct StackRec {
    uint32_t a;         /* VISIT: pc   RESTORE: slot */
    uint32_t restore;
    int64_t  old;
};

}  // namespace

__host__ __device__ sre_pike_layout_t
sre_pike_layout(uint32_t len, uint32_t nthreads, uint32_t nslots)
{
    sre_pike_layout_t L;
    size_t            off = SRE_DEV_ALIGN(sizeof(PikeHdr));
    L.node_bytes = (uint32_t) (sizeof(Node) + (size_t) nslots * 8);
    L.tags = off;       off += SRE_DEV_ALIGN((size_t) (len + 1) * 4);
    L.initial = off;    off += SRE_DEV_ALIGN((size_t) (nthreads + 1) * 4);
    L.nodes[0] = off;   off += SRE_DEV_ALIGN((size_t) SRE_LIST_CAP(nthreads) * L.node_bytes);
    L.nodes[1] = off;   off += SRE_DEV_ALIGN((size_t) SRE_LIST_CAP(nthreads) * L.node_bytes);
    L.matched = off;    off += SRE_DEV_ALIGN((size_t) (nslots + 1) * 8);
    L.work = off;       off += SRE_DEV_ALIGN((size_t) (nslots + 1) * 8);
    L.stack = off;      off += SRE_DEV_ALIGN((size_t) (len + 2) * sizeof(StackRec));
    L.total = off;
    return L;
}

__host__ __device__ sre_thompson_layout_t
sre_thompson_layout(uint32_t len)
{
    sre_thompson_layout_t L;
    size_t                off = 64;
    L.tags = off;       off += SRE_DEV_ALIGN((size_t) (len + 1) * 4);
    L.list[0] = off;    off += SRE_DEV_ALIGN((size_t) SRE_LIST_CAP(len) * 4);
    L.list[1] = off;    off += SRE_DEV_ALIGN((size_t) SRE_LIST_CAP(len) * 4);
    L.stack = off;      off += SRE_DEV_ALIGN((size_t) (len + 2) * 4);
    L.total = off;
    return L;
}

namespace {

struct Pike {
    Prog               P;
    PikeHdr           *h;
    uint32_t          *tags, *initial;
    uint8_t           *nodes0;           /* list l's nodes at nodes0 + l * nodes_step (a member ARRAY indexed at run time would
                                          * send the whole struct to scratch memory) */
    size_t             nodes_step;
    int64_t           *matched, *work;
    StackRec          *stack;
    uint32_t           node_bytes, nslots;
    Chunk              in;

    __device__ inline Node *node(int l, int32_t i) const
    {
        return reinterpret_cast<Node *>(nodes0 + (size_t) l * nodes_step + (size_t) i * node_bytes);
    }
    __device__ inline int64_t *cap(Node *n) const { return reinterpret_cast<int64_t *>(n + 1); }

    __device__ inline void list_reset(int l)
    {
        h->head[l] = h->tail[l] = -1;
        h->used[l] = 0;
        h->count[l] = 0;
    }

    /* Append a thread carrying the working capture vector.  A list holds each
     * list-able instruction at most once per generation — except on programs
     * whose assertion splice re-marks and re-lists in a cycle (a look-ahead
     * assertion inside an empty loop, e.g. (\n?|^$)+?): there the reference VM
     * duplicates threads without bound and crashes (sre_vm_pike.c:506-526).
     * Here the list stops growing and the exec ends with SRE_ERROR. */
    __device__ inline int32_t node_new(int l, uint32_t pc, uint32_t seen_word)
    {
        if (h->used[l] >= SRE_LIST_CAP(P.h->nthreads)) {
            h->overflow = 1;
            return -1;
        }
        int32_t  i = (int32_t) h->used[l]++;
        Node    *n = node(l, i);
        h->count[l]++;
        int64_t *c = cap(n);
        n->pc = pc;
        n->seen_word = seen_word;
        n->next = -1;
        for (uint32_t k = 0; k < nslots; k++) c[k] = work[k];
        return i;
    }

    /*
     * Epsilon closure from `pc0` at chunk offset `pos` (sre_vm_pike.c:756-942).
     * The capture vector of the source thread is in `work`; on return `work`
     * is unchanged.  Threads are appended to the (head, tail) chain given by
     * reference in list `l`.  from_loop == the reference's pcap != NULL.
     */
    __device__ int64_t closure(int l, int32_t &head, int32_t &tail, uint32_t pc0,
                               int64_t pos, bool from_loop)
    {
        uint32_t sptr = 0;
        uint32_t tag = h->tag;
        bool     restart = true;
        uint32_t pc = pc0;

        for (;;) {
            if (!restart) {
                /* "return": unwind to the next pending SPLIT branch */
                for (;;) {
                    if (sptr == 0) return RC_OK;
                    StackRec r = stack[--sptr];
                    if (r.restore) {
                        work[r.a] = r.old;
                    } else {
                        pc = r.a;
                        break;
                    }
                }
            }
            restart = false;

            /* follow one chain of tail calls */
            for (;;) {
                const sre_dev_insn_t in = P.insns[pc];
                uint32_t             seen_word = 0;
                bool                 list_it = false;

                if (tags[pc] == tag) {
                    /* :770-787 */
                    if (in.opcode == OP_SPLIT && tags[in.y] != tag) {
                        if (pc == 0) h->seen_start_state = 1;
                        pc = in.y;
                        continue;
                    }
                    break;
                }
                tags[pc] = tag;

                switch (in.opcode) {
                case OP_JMP:
                    pc = in.x;
                    continue;
                case OP_SPLIT:
                    if (pc == 0) h->seen_start_state = 1;      /* :799-802 */
                    stack[sptr++] = StackRec{in.y, 0u, 0};
                    pc = in.x;
                    continue;
                case OP_SAVE:
                    stack[sptr++] = StackRec{in.arg, 1u, work[in.arg]};
                    work[in.arg] = h->processed_bytes + pos;
                    pc = pc + 1;
                    continue;
                case OP_ASSERT:
                    if (in.ch == AS_BIG_A) {
                        if (pos || h->processed_bytes) break;
                        pc = pc + 1;
                        continue;
                    }
                    if (in.ch == AS_CARET) {
                        if (pos == 0) {
                            if (h->processed_bytes && !h->seen_newline) break;
                        } else if (in_at(pos - 1) != '\n') {
                            break;
                        }
                        pc = pc + 1;
                        continue;
                    }
                    if (in.ch == AS_SMALL_B || in.ch == AS_BIG_B) {
                        seen_word = pos == 0 ? 0u : (uint32_t) is_word(in_at(pos - 1));
                    }
                    list_it = true;
                    break;
                case OP_MATCH:
                    h->last_matched_pos = work[1];
                    if (from_loop) {
                        /* :895-898 SRE_DONE: the capture becomes the match */
                        for (uint32_t k = 0; k < nslots; k++) matched[k] = work[k];
                        h->matched_regex_id = in.arg;
                        /* undo pending SAVEs so `work` is the caller's again */
                        while (sptr) {
                            StackRec r = stack[--sptr];
                            if (r.restore) work[r.a] = r.old;
                        }
                        return RC_DONE;
                    }
                    list_it = true;
                    break;
                default:
                    list_it = true;
                    break;
                }

                if (list_it) {
                    int32_t i = node_new(l, pc, seen_word);
                    if (i < 0) break;               /* overflow: flagged, the exec will fail */
                    if (tail >= 0) {
                        node(l, tail)->next = i;
                    } else {
                        head = i;
                    }
                    tail = i;
                }
                break;
            }
        }
    }

    __device__ inline unsigned in_at(int64_t i) const { return in.at(i); }

    /* sre_vm_pike.c:992-1061 */
    __device__ int64_t find_first_byte(int64_t pos, int64_t last) const
    {
        const int32_t lb = P.h->leading_byte;
        for (; pos != last; pos++) {
            const unsigned c = in_at(pos);
            if (lb != -1) {
                if (c == (unsigned) lb) return pos;
                continue;
            }
            for (uint32_t i = 0; i < P.h->nleading; i++) {
                if (consumes(P, P.insns[P.leading[i]], c)) return pos;
            }
        }
        return pos;
    }

    __device__ void load_work(Node *n)
    {
        const int64_t *c = cap(n);
        for (uint32_t k = 0; k < nslots; k++) work[k] = c[k];
    }

    /* sre_vm_pike.c:945-989 */
    __device__ int64_t prepare_matched(int64_t *ov, uint64_t ovec_slots, bool complete)
    {
        int64_t id = h->matched_regex_id;
        if (id >= (int64_t) P.h->nregexes) return RC_ERROR;
        uint64_t ofs = 0;
        for (int64_t i = 0; i < id; i++) ofs += P.multi_ncaps[i] + 1;
        ofs *= 2;
        uint64_t n = complete ? 2ull * (P.multi_ncaps[id] + 1) : 2ull;
        for (uint64_t k = 0; k < n && k < ovec_slots; k++) ov[k] = matched[ofs + k];
        if (complete) {
            for (uint64_t k = n; k < ovec_slots; k++) ov[k] = -1;
        }
        return RC_OK;
    }

    /* sre_vm_pike.c:692-735 (end offset read without the per-regex offset, :721) */
    __device__ void prepare_temp(int64_t *ov, int l)
    {
        int64_t a0 = -1, a1 = -1;
        for (int32_t i = h->head[l]; i >= 0; i = node(l, i)->next) {
            const int64_t *c = cap(node(l, i));
            uint64_t       ofs = 0;
            for (uint32_t r = 0; r < P.h->nregexes; r++) {
                int64_t b = c[ofs];
                if (b != -1 && (a0 == -1 || b < a0)) a0 = b;
                b = c[1];
                if (b != -1 && (a1 == -1 || b > a1)) a1 = b;
                ofs += 2ull * (P.multi_ncaps[r] + 1);
            }
        }
        ov[0] = a0;
        ov[1] = a1;
    }

    /* `start` > 0 (fresh contexts only): the search is picked up at offset `start`
     * of the buffer, where the list is known to be the freshly seeded initial
     * closure and nothing else (a CLEAN position found by the NFA scanner,
     * sre_hip_nfa.hip).  The list the reference holds there is closure(0) at that
     * offset; its initial-state snapshot (:218-229) is still the one taken at
     * offset 0.  start_is_skip_target: the reference is, at `start`, inside a
     * leading-byte skip that began in front of it (:256-309): it re-seeds at the
     * next byte that can start a match and steps that byte WITHOUT another
     * initial-state check, seen_start_state still set. */
    __device__ int64_t exec(uint64_t size, unsigned eof, bool want_pending,
                            sre_dev_result_t *res, int64_t *ov, uint64_t ovec_slots,
                            const sre_dev_req_t *preset = nullptr, int64_t start = 0,
                            bool start_is_skip_target = false)
    {
        const bool fresh = (h->magic != PIKE_MAGIC);
        if (fresh) {
            /* fresh (zero-filled) context: sre_vm_pike.c:94-145 */
            h->magic = PIKE_MAGIC;
            h->processed_bytes = 0;
            h->last_matched_pos = -1;
            h->tag = 1;
            h->first_buf = 1;
            h->eof = h->empty_capture = h->seen_newline = h->seen_word = 0;
            h->has_matched = 0;
            h->cur = 0;
            h->seen_start_state = 0;
            h->initial_count = 0;
            list_reset(0);
            list_reset(1);
            if (preset && preset->preset_valid) {
                /* earlier searches of this context ran on the scanner */
                h->processed_bytes = preset->preset_processed;
                h->empty_capture = (preset->preset_flags & SRE_PRESET_EMPTY_CAPTURE) ? 1 : 0;
                h->seen_newline = (preset->preset_flags & SRE_PRESET_SEEN_NEWLINE) ? 1 : 0;
                h->seen_word = (preset->preset_flags & SRE_PRESET_SEEN_WORD) ? 1 : 0;
                h->eof = (preset->preset_flags & SRE_PRESET_EOF) ? 1 : 0;
            }
        }

        res->has_pending = 0;
        res->consumed = 0;
        if (h->eof) return RC_ERROR;                               /* :165-168 */

        int     cl = (int) h->cur, nl = cl ^ 1;
        int64_t sp = 0, last = (int64_t) size;
        bool    has_matched = h->has_matched != 0;
        bool    no_check_once = false, skip_ran_out = false;

        h->last_matched_pos = -1;
        if (h->empty_capture) {                                    /* :179-196 */
            h->empty_capture = 0;
            if (size == 0) {
                if (eof) {
                    h->eof = 1;
                    return RC_DECLINED;
                }
                return RC_AGAIN;
            }
            sp = 1;
        }

        if (h->first_buf) {                                        /* :202-233 */
            h->first_buf = 0;
            for (uint32_t k = 0; k < nslots; k++) work[k] = -1;
            h->tag++;
            list_reset(cl);
            closure(cl, h->head[cl], h->tail[cl], 0, sp, false);
            /* snapshot of the initial closure, all but its last thread (:218-229) */
            h->initial_count = h->count[cl];
            uint32_t k = 0;
            for (int32_t i = h->head[cl]; i >= 0 && node(cl, i)->next >= 0; i = node(cl, i)->next) {
                initial[k++] = node(cl, i)->pc;
            }
            if (start > sp) {
                /* pick the search up at a clean position */
                sp = start;
                if (P.h->nleading && start_is_skip_target) {
                    sp = find_first_byte(sp, last);
                    no_check_once = true;
                    if (sp == last) skip_ran_out = true;            /* :304-306 */
                }
                h->tag++;
                list_reset(cl);
                closure(cl, h->head[cl], h->tail[cl], 0, sp, false);
            }
        }

        for (; !skip_ran_out && (sp < last || (eof && sp == last)); sp++) {   /* :235 */
            if (h->head[cl] < 0) break;

            if (no_check_once) {
                no_check_once = false;
            } else if (P.h->nleading && h->seen_start_state) {     /* :256-309 */
                h->seen_start_state = 0;
                bool same = (sp != last) && (h->count[cl] == h->initial_count);
                uint32_t k = 0;
                for (int32_t i = h->head[cl]; same && i >= 0 && node(cl, i)->next >= 0;
                     i = node(cl, i)->next)
                {
                    if (node(cl, i)->pc != initial[k++]) same = false;
                }
                if (same) {
                    int64_t p = find_first_byte(sp, last);
                    if (p > sp) {
                        sp = p;
                        list_reset(cl);
                        for (uint32_t s2 = 0; s2 < nslots; s2++) work[s2] = -1;
                        h->tag++;
                        closure(cl, h->head[cl], h->tail[cl], 0, sp, false);
                        if (sp == last) break;
                    }
                }
            }
            h->tag++;                                              /* :312 */
            const bool     at_end = (sp == last);
            const unsigned c = at_end ? 0u : in_at(sp);
            bool           done = false;

            while (h->head[cl] >= 0) {                             /* :314 */
                Node *t = node(cl, h->head[cl]);
                h->head[cl] = t->next;
                if (h->head[cl] < 0) h->tail[cl] = -1;
                h->count[cl]--;
                const uint32_t       pc = t->pc;
                const sre_dev_insn_t in = P.insns[pc];

                if (in.opcode == OP_ASSERT) {                      /* :450-528 */
                    bool hold = false;
                    if (in.ch == AS_SMALL_Z) {
                        hold = at_end;
                    } else if (in.ch == AS_DOLLAR) {
                        hold = at_end || c == '\n';
                    } else {
                        bool sw = t->seen_word || (sp == 0 && h->seen_word);
                        hold = sw != (!at_end && is_word(c));
                        if (in.ch == AS_BIG_B) hold = !hold;
                    }
                    if (!hold) continue;
                    /* closure at the same offset under the CURRENT list's
                     * generation, spliced in front of the rest (:506-526) */
                    load_work(t);
                    int32_t sh = -1, st = -1;
                    h->tag--;
                    closure(cl, sh, st, pc + 1, sp, false);
                    h->tag++;
                    if (sh >= 0) {
                        /* node_new() already counted the spliced threads */
                        node(cl, st)->next = h->head[cl];
                        if (h->head[cl] < 0) h->tail[cl] = st;
                        h->head[cl] = sh;
                    }
                    continue;
                }

                if (in.opcode == OP_MATCH) {                       /* :530-553 */
                    const int64_t *cp = cap(t);
                    h->last_matched_pos = cp[1];
                    for (uint32_t k = 0; k < nslots; k++) matched[k] = cp[k];
                    h->matched_regex_id = in.arg;
                    done = true;
                    break;
                }

                if (at_end || !consumes(P, in, c)) continue;       /* :329-448 */
                load_work(t);
                if (closure(nl, h->head[nl], h->tail[nl], pc + 1, sp + 1, true) == RC_DONE) {
                    done = true;
                    break;
                }
            }

            if (done) {
                /* every thread of lower priority is dropped (:547-553) */
                has_matched = true;
            }
            if (h->overflow) return RC_ERROR;
            /* step_done :569-580 */
            list_reset(cl);
            cl ^= 1;
            nl ^= 1;
            if (at_end) break;
        }

        if (h->last_matched_pos >= 0) {                            /* :586-601 */
            int64_t p = h->last_matched_pos - h->processed_bytes;
            if (p > 0) {
                unsigned b = in_at(p - 1);
                h->seen_newline = (b == '\n');
                h->seen_word = is_word(b);
            }
            h->last_matched_pos = -1;
        }

        h->cur = (uint32_t) cl;
        res->consumed = sp;

        if (has_matched) {                                         /* :607-658 */
            if (eof || h->head[cl] < 0) {
                if (prepare_matched(ov, ovec_slots, true) != RC_OK) return RC_ERROR;
                if (h->head[cl] >= 0) {
                    list_reset(cl);
                    h->eof = 1;
                }
                /* :624-628 re-arm for the next search on the same context */
                uint64_t ofs = 0;
                for (int64_t i = 0; i < h->matched_regex_id; i++) ofs += P.multi_ncaps[i] + 1;
                ofs *= 2;
                h->processed_bytes = matched[ofs + 1];
                h->empty_capture = (matched[ofs] == matched[ofs + 1]);
                h->has_matched = 0;
                h->first_buf = 1;
                return h->matched_regex_id;
            }
            if (want_pending) {
                res->has_pending = 1;
                if (prepare_matched(res->pending, 2, false) != RC_OK) return RC_ERROR;
            }
        } else if (eof) {                                          /* :660-666 */
            h->eof = 1;
            h->has_matched = 0;
            return RC_DECLINED;
        }

        h->processed_bytes += sp;                                  /* :673-688 */
        h->has_matched = has_matched ? 1u : 0u;
        if (ovec_slots >= 2) prepare_temp(ov, cl);
        return RC_AGAIN;
    }
};


/* ============================================================ Thompson, one wavefront per stream */

constexpr uint32_t THOMPSON_WAVE_MAGIC = 0x54485756u;

/*
 * BASELINE.json's north_star layout, for the VM whose answer is a SET property (match / no match,
 * sre_vm_thompson.c:63-270): lane q of the wave is thread q of the program's bit-parallel form,
 * the live set S is a 64-bit lane mask in scalar registers, and one input byte is
 *      T  = S & accept[byte]                       two scalar ANDs
 *      S' = ballot((pred[lane] & T) != 0)          every lane: "does a thread that consumed list me?"
 * The wave fetches 64 input bytes with one coalesced load (lane l byte l) and their accept masks
 * with one gather; inside the block a byte costs two v_readlane, two scalar and four vector
 * instructions and no memory access.  MATCH answers when the position it is listed at is RUN
 * (:233-235): by the next byte, or by the extra iteration at end of input — a match completed by a
 * chunk's last byte waits for the next call.  The context between calls is the mask.
 * Programs with look-ahead assertions (their splices go by generation tags, sre_nfa.cpp) and with
 * more than 64 thread bits keep the scalar VM below.
 */
__device__ inline uint64_t
wave_readlane64(uint64_t v, uint32_t i)
{
    const uint32_t lo = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) v, (int) i);
    const uint32_t hi = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) (v >> 32), (int) i);
    return ((uint64_t) hi << 32) | lo;
}

__device__ inline uint64_t
wave_uniform64(uint64_t v)
{
    const uint32_t lo = (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) v);
    const uint32_t hi = (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) (v >> 32));
    return ((uint64_t) hi << 32) | lo;
}

/* one exec() of sre_vm_thompson_exec on the set S; input == nullptr: the (<= 8) bytes travel in `inl`.
 * STABLE RUNS: the step is a function of (S, byte), so a byte that maps S to itself joins a 256-bit set
 * kept for that S, and from the second such step in a row on the bytes of the set that follow are
 * skipped — inside the block by one ballot, across blocks 512 bytes at a time (`[a-z]+` inside a word,
 * the idle set of a program between candidates). */
__device__ inline int64_t
thompson_wave_run(const sre_dev_wave_t *W, uint64_t &S, const uint8_t *input, uint64_t inl, uint64_t size, bool eof)
{
    __shared__ uint32_t stab[8];
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t pred = W->pred[lane];
    const uint32_t pred_lo = (uint32_t) pred, pred_hi = (uint32_t) (pred >> 32);
    const uint64_t match = wave_uniform64(W->match);
    uint64_t       stabS = 0;
    bool           stab_valid = false, streak = false;
    uint64_t       base = 0;
    while (base < size) {
        if (streak && stab_valid && stabS == S && input != nullptr) {
            /* across blocks: the first byte at or behind `base` that is not in the set */
            bool found = false;
            while (!found && base < size) {
                uint32_t c[8];
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const uint64_t p = base + (uint64_t) k * 64 + lane;
                    c[k] = p < size ? (uint32_t) input[p] : 256u;
                }
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const bool     stop = c[k] > 255u || !((stab[(c[k] >> 5) & 7u] >> (c[k] & 31)) & 1u);
                    const uint64_t m = __builtin_amdgcn_ballot_w64(stop);
                    if (!found && m) {
                        base += (uint64_t) k * 64 + (uint64_t) __builtin_ctzll(m);
                        found = true;
                    }
                }
                if (!found) base += 512;
            }
            if (base >= size) break;
        }
        const uint64_t idx = base + lane;
        uint32_t       byte = 0;
        if (idx < size) byte = input != nullptr ? input[idx] : (uint32_t) ((inl >> (8 * idx)) & 0xffu);
        const uint64_t acc = W->accept[byte];
        const uint32_t nb = size - base < 64 ? (uint32_t) (size - base) : 64u;
        uint32_t       i = 0;
        while (i < nb) {
            if (S == 0) return eof ? RC_DECLINED : RC_AGAIN;            /* :89 the list is empty */
            if (S & match) return RC_OK;                                /* :233-235 */
            const uint64_t T = S & wave_readlane64(acc, i);
            const uint64_t S1 = __builtin_amdgcn_ballot_w64(((pred_lo & (uint32_t) T) | (pred_hi & (uint32_t) (T >> 32))) != 0);
            i++;
            if (S1 != S) {
                S = S1;
                streak = false;
                continue;
            }
            if (streak) {
                const uint32_t b = (uint32_t) __builtin_amdgcn_readlane((int) byte, (int) (i - 1));
                if (!stab_valid || stabS != S) {
                    if (lane < 8) stab[lane] = 0;
                    stabS = S;
                    stab_valid = true;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                if (lane == 0) stab[b >> 5] |= 1u << (b & 31);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                if (i < nb) {
                    /* inside the block: the run of set bytes from lane i on */
                    const bool     mem = lane < nb && ((stab[byte >> 5] >> (byte & 31)) & 1u);
                    const uint64_t stopm = ~__builtin_amdgcn_ballot_w64(mem) >> i;     /* (lanes >= nb stop it) */
                    i += stopm ? (uint32_t) __builtin_ctzll(stopm) : 64u - i;
                }
            }
            streak = true;
        }
        base += nb;
    }
    /* the extra iteration at end of input (:88): a listed MATCH is met, nothing consumes */
    if (eof && (S & match)) return RC_OK;
    return eof ? RC_DECLINED : RC_AGAIN;
}

/* =================================================================== Thompson */

constexpr uint32_t THOMPSON_MAGIC = 0x54484f4du;

struct ThompsonHdr {
    uint32_t magic, tag, first_buf, cur;
    uint32_t count[2];
    uint32_t overflow, pad;       /* see Pike::node_new */
};

struct Thompson {
    Prog         P;
    ThompsonHdr *h;
    uint32_t    *tags;
    uint32_t    *list0;        /* entries: pc | seen_word << 31; list l at list0 + l * list_step (see Pike::nodes0) */
    size_t       list_step;
    uint32_t    *stack;
    Chunk        in;

    /* sre_vm_thompson.c:273-345; `sp` is the chunk offset, chunk start == 0 */
    __device__ void closure(int l, uint32_t pc0, int64_t sp)
    {
        uint32_t sptr = 0, tag = h->tag, pc = pc0;
        bool     first = true;

        for (;;) {
            if (!first) {
                if (sptr == 0) return;
                pc = stack[--sptr];
            }
            first = false;
            for (;;) {
                const sre_dev_insn_t in = P.insns[pc];
                uint32_t             seen_word = 0;

                if (tags[pc] == tag) break;          /* plain de-dup (:280-282) */
                tags[pc] = tag;

                if (in.opcode == OP_JMP) {
                    pc = in.x;
                    continue;
                }
                if (in.opcode == OP_SPLIT) {
                    stack[sptr++] = in.y;
                    pc = in.x;
                    continue;
                }
                if (in.opcode == OP_SAVE) {
                    pc = pc + 1;
                    continue;
                }
                if (in.opcode == OP_ASSERT) {
                    if (in.ch == AS_BIG_A) {
                        if (sp != 0) break;          /* chunk-local (:302-309) */
                        pc = pc + 1;
                        continue;
                    }
                    if (in.ch == AS_CARET) {
                        if (sp != 0 && this->in.at(sp - 1) != '\n') break;
                        pc = pc + 1;
                        continue;
                    }
                    if (in.ch == AS_SMALL_B || in.ch == AS_BIG_B) {
                        seen_word = (sp != 0 && is_word(this->in.at(sp - 1))) ? 1u : 0u;
                    }
                }
                if (h->count[l] >= SRE_LIST_CAP(P.h->len)) {
                    h->overflow = 1;            /* see Pike::node_new */
                    break;
                }
                list0[(size_t) l * list_step + h->count[l]++] = pc | (seen_word << 31);
                break;
            }
        }
    }

    __device__ int64_t exec(uint64_t size, unsigned eof, sre_dev_result_t *res)
    {
        if (h->magic != THOMPSON_MAGIC) {
            h->magic = THOMPSON_MAGIC;
            h->tag = 1;
            h->first_buf = 1;
            h->cur = 0;
            h->count[0] = h->count[1] = 0;
        }
        int cl = (int) h->cur, nl = cl ^ 1;

        if (h->first_buf) {                                        /* :81-84 */
            h->first_buf = 0;
            closure(cl, 0, 0);
        }

        int64_t sp = 0, last = (int64_t) size;
        for (; sp < last || (eof && sp == last); sp++) {           /* :88 */
            if (h->count[cl] == 0) break;
            h->tag++;
            const bool     at_end = (sp == last);
            const unsigned c = at_end ? 0u : in.at(sp);

            for (uint32_t i = 0; i < h->count[cl]; i++) {
                const uint32_t       e = list0[(size_t) cl * list_step + i];
                const uint32_t       pc = e & 0x7fffffffu;
                const sre_dev_insn_t ins = P.insns[pc];

                if (ins.opcode == OP_MATCH) {                      /* :233-235 */
                    res->consumed = sp;
                    return RC_OK;
                }
                if (ins.opcode == OP_ASSERT) {                     /* :174-231 */
                    bool hold;
                    if (ins.ch == AS_SMALL_Z) {
                        hold = at_end;
                    } else if (ins.ch == AS_DOLLAR) {
                        hold = at_end || c == '\n';
                    } else {
                        hold = ((e >> 31) != 0) != (!at_end && is_word(c));
                        if (ins.ch == AS_BIG_B) hold = !hold;
                    }
                    if (hold) {
                        h->tag--;
                        closure(cl, pc + 1, sp);       /* appended to the current list */
                        h->tag++;
                    }
                    continue;
                }
                if (at_end || !consumes(P, ins, c)) continue;
                closure(nl, pc + 1, sp + 1);
            }

            if (h->overflow) return RC_ERROR;
            h->count[cl] = 0;
            cl ^= 1;
            nl ^= 1;
            if (at_end) break;
        }

        h->cur = (uint32_t) cl;
        res->consumed = sp;
        return eof ? RC_DECLINED : RC_AGAIN;
    }
};

}  // namespace

/*
 * One WORKGROUP per request.  The stream context (a few KiB: thread lists, tags,
 * capture vectors) is copied into LDS by the whole wave, lane 0 runs the VM on it
 * there, and the wave copies it back: the VM touches its context several times per
 * byte and thread, and from HBM every one of those is a ~1 us round trip.  A context
 * too large for LDS (thousands of instructions) is worked on in place.
 */
/* copy the program image into LDS (blob_lds_bytes > 0) and return where to read it */
__device__ inline const uint8_t *
stage_blob(const uint8_t *blob, uint8_t *lds, uint32_t blob_lds_bytes)
{
    if (blob_lds_bytes == 0) return blob;
    for (uint32_t i = threadIdx.x * 16; i < blob_lds_bytes; i += blockDim.x * 16) {
        *reinterpret_cast<uint4 *>(lds + i) = *reinterpret_cast<const uint4 *>(blob + i);
    }
    __syncthreads();
    return lds;
}

/* a chunk that came in pinned host memory (sre_dev_req_t.input_pinned): into LDS, every load in flight at once */
__device__ inline const uint8_t *
stage_small_input(const sre_dev_req_t &rq, uint8_t *sh)
{
    if (!rq.input_pinned || rq.input == nullptr) return rq.input;
    for (uint32_t i = threadIdx.x * 16; i < rq.size; i += blockDim.x * 16) {
        *reinterpret_cast<uint4 *>(sh + i) = *reinterpret_cast<const uint4 *>(rq.input + i);
    }
    __syncthreads();
    return sh;
}

__device__ inline void
ctx_fill_zero(uint8_t *dst, size_t bytes)
{
    for (size_t i = (size_t) threadIdx.x * 16; i < bytes; i += (size_t) blockDim.x * 16) {
        *reinterpret_cast<uint4 *>(dst + i) = make_uint4(0, 0, 0, 0);
    }
}

__device__ inline void
ctx_copy(uint8_t *dst, const uint8_t *src, size_t bytes)
{
    for (size_t i = (size_t) threadIdx.x * 16; i < bytes; i += (size_t) blockDim.x * 16) {
        *reinterpret_cast<uint4 *>(dst + i) = *reinterpret_cast<const uint4 *>(src + i);
    }
}

extern "C" __global__ void
sre_k_pike_exec(const uint8_t *__restrict__ blob, const sre_dev_req_t *__restrict__ reqs,
                uint32_t nreqs, uint32_t use_lds, uint32_t blob_lds_bytes)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t ctx_lds[];
    __shared__ __attribute__((aligned(16))) uint8_t sh_small_in[SRE_SMALL_INPUT];
    const uint32_t i = blockIdx.x;
    if (i >= nreqs) return;
    const sre_dev_req_t rq = reqs[i];
    /* the program image too: the VM fetches an instruction per thread and step */
    blob = stage_blob(blob, ctx_lds, blob_lds_bytes);
    const uint8_t *input = stage_small_input(rq, sh_small_in);

    Pike vm;
    vm.P = prog_view(blob);
    const sre_pike_layout_t L = sre_pike_layout(vm.P.h->len, vm.P.h->nthreads, vm.P.h->nslots);
    uint8_t *home = static_cast<uint8_t *>(rq.ctx);
    uint8_t *base = use_lds ? ctx_lds + blob_lds_bytes : home;
    if (rq.fresh) {
        ctx_fill_zero(base, L.total);       /* zero-filled == fresh (tags, lists, the magic word) */
        __syncthreads();
    } else if (use_lds) {
        ctx_copy(base, home, L.total);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        vm.h = reinterpret_cast<PikeHdr *>(base);
        vm.tags = reinterpret_cast<uint32_t *>(base + L.tags);
        vm.initial = reinterpret_cast<uint32_t *>(base + L.initial);
        vm.nodes0 = base + L.nodes[0];
        vm.nodes_step = L.nodes[1] - L.nodes[0];
        vm.matched = reinterpret_cast<int64_t *>(base + L.matched);
        vm.work = reinterpret_cast<int64_t *>(base + L.work);
        vm.stack = reinterpret_cast<StackRec *>(base + L.stack);
        vm.node_bytes = L.node_bytes;
        vm.nslots = vm.P.h->nslots;
        vm.in.p = input;
        vm.in.inl = rq.inline_bytes;

        sre_dev_result_t *res = static_cast<sre_dev_result_t *>(rq.result);
        int64_t          *ov = reinterpret_cast<int64_t *>(res + 1);
        const int64_t     rc = vm.exec(rq.size, rq.eof, rq.want_pending != 0, res, ov, rq.ovec_slots, &rq);
        /* what the context holds between two searches (the host mirrors it after a match, so the
         * next search may run on a throughput scanner again: sre_vm_api.cpp) */
        res->pad[0] = 16 | (vm.h->empty_capture ? SRE_PRESET_EMPTY_CAPTURE : 0) | (vm.h->seen_newline ? SRE_PRESET_SEEN_NEWLINE : 0)
                      | (vm.h->seen_word ? SRE_PRESET_SEEN_WORD : 0) | (vm.h->eof ? SRE_PRESET_EOF : 0);
        res->pad[1] = vm.h->processed_bytes;
        /* the host watches res->rc (device_stream_exec): written last, behind a system-scope fence */
        __threadfence_system();
        *reinterpret_cast<volatile int64_t *>(&res->rc) = rc;
    }
    if (use_lds) {
        __syncthreads();
        ctx_copy(home, base, L.total);
    }
}

extern "C" __global__ void
sre_k_thompson_exec(const uint8_t *__restrict__ blob, const sre_dev_req_t *__restrict__ reqs,
                    uint32_t nreqs, uint32_t use_lds, uint32_t blob_lds_bytes)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t ctx_lds[];
    __shared__ __attribute__((aligned(16))) uint8_t sh_small_in[SRE_SMALL_INPUT];
    const uint32_t i = blockIdx.x;
    if (i >= nreqs) return;
    const sre_dev_req_t rq = reqs[i];
    const uint8_t *input = stage_small_input(rq, sh_small_in);
    {
        const uint32_t wave_off = reinterpret_cast<const sre_dev_prog_hdr_t *>(blob)->wave_off;
        if (wave_off != 0) {
            /* the wave form: the context is the live set (zero-filled or flagged == fresh) */
            const sre_dev_wave_t *W = reinterpret_cast<const sre_dev_wave_t *>(blob + wave_off);
            uint64_t             *cw = static_cast<uint64_t *>(rq.ctx);
            uint64_t              S = wave_uniform64(!rq.fresh && (uint32_t) cw[0] == THOMPSON_WAVE_MAGIC ? cw[1] : W->init0);
            const int64_t         rc = thompson_wave_run(W, S, input, rq.inline_bytes, rq.size, rq.eof != 0);
            if (threadIdx.x == 0) {
                sre_dev_result_t *res = static_cast<sre_dev_result_t *>(rq.result);
                cw[0] = THOMPSON_WAVE_MAGIC;
                cw[1] = S;
                res->has_pending = 0;
                res->consumed = 0;
                __threadfence_system();
                *reinterpret_cast<volatile int64_t *>(&res->rc) = rc;
            }
            return;
        }
    }
    /* the program image too: the VM fetches an instruction per thread and step */
    blob = stage_blob(blob, ctx_lds, blob_lds_bytes);

    Thompson vm;
    vm.P = prog_view(blob);
    const sre_thompson_layout_t L = sre_thompson_layout(vm.P.h->len);
    uint8_t *home = static_cast<uint8_t *>(rq.ctx);
    uint8_t *base = use_lds ? ctx_lds + blob_lds_bytes : home;
    if (rq.fresh) {
        ctx_fill_zero(base, L.total);
        __syncthreads();
    } else if (use_lds) {
        ctx_copy(base, home, L.total);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        vm.h = reinterpret_cast<ThompsonHdr *>(base);
        vm.tags = reinterpret_cast<uint32_t *>(base + L.tags);
        vm.list0 = reinterpret_cast<uint32_t *>(base + L.list[0]);
        vm.list_step = (L.list[1] - L.list[0]) / 4;
        vm.stack = reinterpret_cast<uint32_t *>(base + L.stack);
        vm.in.p = input;
        vm.in.inl = rq.inline_bytes;

        sre_dev_result_t *res = static_cast<sre_dev_result_t *>(rq.result);
        res->has_pending = 0;
        res->consumed = 0;
        const int64_t rc = vm.exec(rq.size, rq.eof, res);
        __threadfence_system();
        *reinterpret_cast<volatile int64_t *>(&res->rc) = rc;
    }
    if (use_lds) {
        __syncthreads();
        ctx_copy(home, base, L.total);
    }
}

/*
 * Whole-stream scan for the batched API: lane i owns stream i for the whole
 * call (fresh context, eof = 1).  mode 1: first match; mode 2: the find-all
 * iteration a caller writes around sre_vm_pike_exec, re-feeding the SAME
 * context from each match end (sre_vm_pike.c:179-196, 624-628).
 * Record per stream: [rc, count, ovector[ovec_slots]].
 */
extern "C" __global__ void
sre_k_pike_scan(const uint8_t *__restrict__ blob, const uint8_t *const *__restrict__ streams,
                const uint64_t *__restrict__ lens, uint32_t nstreams, uint8_t *ctx_base,
                uint64_t ctx_stride, int64_t *__restrict__ records, uint32_t ovec_slots,
                int mode)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nstreams) return;

    Pike vm;
    vm.P = prog_view(blob);
    const sre_pike_layout_t L = sre_pike_layout(vm.P.h->len, vm.P.h->nthreads, vm.P.h->nslots);
    uint8_t *base = ctx_base + (size_t) i * ctx_stride;
    vm.h = reinterpret_cast<PikeHdr *>(base);
    vm.tags = reinterpret_cast<uint32_t *>(base + L.tags);
    vm.initial = reinterpret_cast<uint32_t *>(base + L.initial);
    vm.nodes0 = base + L.nodes[0];
    vm.nodes_step = L.nodes[1] - L.nodes[0];
    vm.matched = reinterpret_cast<int64_t *>(base + L.matched);
    vm.work = reinterpret_cast<int64_t *>(base + L.work);
    vm.stack = reinterpret_cast<StackRec *>(base + L.stack);
    vm.node_bytes = L.node_bytes;
    vm.nslots = vm.P.h->nslots;
    vm.in.inl = 0;

    int64_t         *rec = records + (size_t) i * (2 + ovec_slots);
    int64_t         *ov = rec + 2;
    const uint8_t   *s = streams[i];
    const uint64_t   n = lens[i];
    sre_dev_result_t res;
    uint64_t         off = 0;
    int64_t          count = 0, rc, last_rc = RC_DECLINED;

    for (uint32_t k = 0; k < ovec_slots; k++) ov[k] = -1;
    for (;;) {
        vm.in.p = s + off;
        rc = vm.exec(n - off, 1u, false, &res, ov, ovec_slots);
        if (rc < 0) break;
        count++;
        last_rc = rc;
        if (mode != 2) break;
        off = (uint64_t) vm.h->processed_bytes;      /* == ovector[1] of this match */
    }
    /* the final DECLINED leaves the ovector of the last match in place
     * (sre_vm_pike.c:660-666 writes nothing) */
    rec[0] = (rc == RC_ERROR) ? rc : (count > 0 ? last_rc : rc);
    rec[1] = count;
}

/*
 * The exact Pike VM over a WINDOW of each stream: from the clean position the
 * bit-parallel NFA scanner found in front of the stream's first MATCH event
 * (sre_hip_nfa.hip) to the end of the search.  Streams without an event, or not
 * yet verified, already have their record.
 */
extern "C" __global__ void
sre_k_pike_window(const uint8_t *__restrict__ blob, const uint8_t *const *__restrict__ streams,
                  const uint64_t *__restrict__ lens, uint32_t nstreams, uint8_t *ctx_base,
                  uint64_t ctx_stride, int64_t *__restrict__ records, uint32_t ovec_slots,
                  sre_nfa_window_t *__restrict__ win, const int64_t *__restrict__ lo,
                  const sre_nfa_count_req_t *__restrict__ creq, uint32_t use_lds, uint32_t blob_lds_bytes)
{
    /* one workgroup per stream; the (fresh) context lives in LDS when it fits, see
     * sre_k_pike_exec */
    extern __shared__ __attribute__((aligned(16))) uint8_t ctx_lds[];
    const uint32_t i = blockIdx.x;
    if (i >= nstreams) return;
    if (lo != nullptr && lo[i] < 0) return;     /* settled in an earlier round */
    if (!win[i].done || win[i].ev_pos < 0) return;
    blob = stage_blob(blob, ctx_lds, blob_lds_bytes);

    Pike vm;
    vm.P = prog_view(blob);
    const sre_pike_layout_t L = sre_pike_layout(vm.P.h->len, vm.P.h->nthreads, vm.P.h->nslots);
    uint8_t *base = use_lds ? ctx_lds + blob_lds_bytes : ctx_base + (size_t) i * ctx_stride;
    /* zero-filled == fresh */
    for (size_t b = (size_t) threadIdx.x * 16; b < L.total; b += (size_t) blockDim.x * 16) {
        *reinterpret_cast<uint4 *>(base + b) = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    vm.h = reinterpret_cast<PikeHdr *>(base);
    vm.tags = reinterpret_cast<uint32_t *>(base + L.tags);
    vm.initial = reinterpret_cast<uint32_t *>(base + L.initial);
    vm.nodes0 = base + L.nodes[0];
    vm.nodes_step = L.nodes[1] - L.nodes[0];
    vm.matched = reinterpret_cast<int64_t *>(base + L.matched);
    vm.work = reinterpret_cast<int64_t *>(base + L.work);
    vm.stack = reinterpret_cast<StackRec *>(base + L.stack);
    vm.node_bytes = L.node_bytes;
    vm.nslots = vm.P.h->nslots;
    vm.in.inl = 0;
    vm.in.p = streams[i];

    int64_t         *rec = records + (size_t) i * (2 + ovec_slots);
    sre_dev_result_t res;
    for (uint32_t k = 0; k < ovec_slots; k++) rec[2 + k] = -1;
    uint64_t      len = lens[i];
    int64_t       start = win[i].clean_pos;
    sre_dev_req_t preset;
    preset.preset_valid = 0;
    if (creq != nullptr) {
        /* a search of a find-all iteration: the context is the re-armed one of the reference's caller */
        vm.in.p = creq[i].vptr;
        len = creq[i].vlen;
        start += creq[i].start_add;
        preset.preset_valid = 1;
        preset.preset_processed = creq[i].processed;
        preset.preset_flags = creq[i].preset_flags;
    }
    const int64_t rc = vm.exec(len, 1u, false, &res, rec + 2, ovec_slots, preset.preset_valid ? &preset : nullptr, start,
                               (win[i].clean_mode & 1) != 0);
    rec[0] = rc;
    rec[1] = rc >= 0 ? 1 : 0;
    /* a match returned with threads still listed at end of input: the context is
     * poisoned, its next exec fails (sre_vm_pike.c:616-622) — the compat API asks */
    if (rc >= 0 && vm.h->eof) win[i].clean_mode |= SRE_NFA_WINDOW_POISONED;
    if (rc >= 0 && creq != nullptr && ovec_slots >= 2) {
        /* what the next search's ^ goes by (seen_newline, :586-601) */
        const int64_t e = rec[3] - creq[i].processed;
        if (e > 0 && vm.in.p[e - 1] == '\n') win[i].clean_mode |= SRE_NFA_MATCH_AFTER_NL;
    }
}

/* dynamic LDS a one-request VM kernel may take for the context (and program) copy */
#define SRE_VM_CTX_LDS_LIMIT (96u * 1024u)

/* bytes of LDS for the program image, 0 when it does not fit next to the context */
static uint32_t
vm_blob_lds(size_t blob_bytes, size_t ctx_bytes)
{
    const size_t b = (blob_bytes + 15) & ~(size_t) 15, c = (ctx_bytes + 15) & ~(size_t) 15;
    return (b <= 32 * 1024 && b + c <= SRE_VM_CTX_LDS_LIMIT) ? (uint32_t) b : 0u;
}

extern "C" hipError_t
sre_launch_pike_window(const void *blob, size_t blob_bytes, const void *const *d_streams, const uint64_t *d_lens,
                       uint32_t nstreams, void *d_ctx, uint64_t ctx_stride, int64_t *d_records,
                       uint32_t ovec_slots, sre_nfa_window_t *d_win, const int64_t *d_lo,
                       const sre_nfa_count_req_t *d_creq, hipStream_t stream)
{
    const uint32_t blob_lds = vm_blob_lds(blob_bytes, ctx_stride);
    const size_t   bytes = (((size_t) ctx_stride + 15) & ~(size_t) 15) + blob_lds;
    const uint32_t use_lds = bytes <= SRE_VM_CTX_LDS_LIMIT ? 1u : 0u;
    if (use_lds && bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(sre_k_pike_window),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, SRE_VM_CTX_LDS_LIMIT);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(sre_k_pike_window, dim3(nstreams), dim3(64), use_lds ? bytes : 0, stream,
                       static_cast<const uint8_t *>(blob),
                       reinterpret_cast<const uint8_t *const *>(d_streams), d_lens, nstreams,
                       static_cast<uint8_t *>(d_ctx), ctx_stride, d_records, ovec_slots, d_win, d_lo, d_creq, use_lds,
                       use_lds ? blob_lds : 0u);
    return hipGetLastError();
}

extern "C" __global__ void
sre_k_thompson_scan(const uint8_t *__restrict__ blob, const uint8_t *const *__restrict__ streams,
                    const uint64_t *__restrict__ lens, uint32_t nstreams, uint8_t *ctx_base,
                    uint64_t ctx_stride, int64_t *__restrict__ records, uint32_t ovec_slots)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nstreams) return;

    Thompson vm;
    vm.P = prog_view(blob);
    const sre_thompson_layout_t L = sre_thompson_layout(vm.P.h->len);
    uint8_t *base = ctx_base + (size_t) i * ctx_stride;
    vm.h = reinterpret_cast<ThompsonHdr *>(base);
    vm.tags = reinterpret_cast<uint32_t *>(base + L.tags);
    vm.list0 = reinterpret_cast<uint32_t *>(base + L.list[0]);
    vm.list_step = (L.list[1] - L.list[0]) / 4;
    vm.stack = reinterpret_cast<uint32_t *>(base + L.stack);
    vm.in.p = streams[i];
    vm.in.inl = 0;

    sre_dev_result_t res;
    int64_t         *rec = records + (size_t) i * (2 + ovec_slots);
    int64_t          rc = vm.exec(lens[i], 1u, &res);
    rec[0] = rc;
    rec[1] = rc == RC_OK ? 1 : 0;
    for (uint32_t k = 0; k < ovec_slots; k++) rec[2 + k] = -1;
}

/* whole streams, one wavefront each (the wave form, see thompson_wave_run) */
extern "C" __global__ __launch_bounds__(64) void
sre_k_thompson_wave_scan(const uint8_t *__restrict__ blob, const uint8_t *const *__restrict__ streams,
                         const uint64_t *__restrict__ lens, uint32_t nstreams, int64_t *__restrict__ records,
                         uint32_t ovec_slots)
{
    const uint32_t i = blockIdx.x;
    if (i >= nstreams) return;
    const sre_dev_wave_t *W = reinterpret_cast<const sre_dev_wave_t *>(
        blob + reinterpret_cast<const sre_dev_prog_hdr_t *>(blob)->wave_off);
    uint64_t      S = wave_uniform64(W->init0);
    const int64_t rc = thompson_wave_run(W, S, streams[i], 0, lens[i], true);
    int64_t      *rec = records + (size_t) i * (2 + ovec_slots);
    if (threadIdx.x == 0) {
        rec[0] = rc;
        rec[1] = rc == RC_OK ? 1 : 0;
    }
    for (uint32_t k = threadIdx.x; k < ovec_slots; k += 64) rec[2 + k] = -1;
}

extern "C" hipError_t
sre_launch_vm_scan(const void *blob, int mode, const void *const *d_streams,
                   const uint64_t *d_lens, uint32_t nstreams, void *d_ctx, uint64_t ctx_stride,
                   int64_t *d_records, uint32_t ovec_slots, int has_wave, hipStream_t stream)
{
    uint32_t block = 64, grid = (nstreams + block - 1) / block;
    if (mode == 0 && has_wave) {
        hipLaunchKernelGGL(sre_k_thompson_wave_scan, dim3(nstreams), dim3(64), 0, stream,
                           static_cast<const uint8_t *>(blob),
                           reinterpret_cast<const uint8_t *const *>(d_streams), d_lens, nstreams, d_records, ovec_slots);
    } else if (mode == 0) {
        hipLaunchKernelGGL(sre_k_thompson_scan, dim3(grid), dim3(block), 0, stream,
                           static_cast<const uint8_t *>(blob),
                           reinterpret_cast<const uint8_t *const *>(d_streams), d_lens, nstreams,
                           static_cast<uint8_t *>(d_ctx), ctx_stride, d_records, ovec_slots);
    } else {
        hipLaunchKernelGGL(sre_k_pike_scan, dim3(grid), dim3(block), 0, stream,
                           static_cast<const uint8_t *>(blob),
                           reinterpret_cast<const uint8_t *const *>(d_streams), d_lens, nstreams,
                           static_cast<uint8_t *>(d_ctx), ctx_stride, d_records, ovec_slots, mode);
    }
    return hipGetLastError();
}

template <typename K>
static hipError_t
vm_exec_launch(K kernel, const void *blob, size_t blob_bytes, const sre_dev_req_t *d_reqs, uint32_t nreqs,
               size_t ctx_bytes, hipStream_t stream)
{
    const uint32_t blob_lds = vm_blob_lds(blob_bytes, ctx_bytes);
    const size_t   bytes = ((ctx_bytes + 15) & ~(size_t) 15) + blob_lds;
    const uint32_t use_lds = bytes <= SRE_VM_CTX_LDS_LIMIT ? 1u : 0u;
    if (use_lds && bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, SRE_VM_CTX_LDS_LIMIT);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kernel, dim3(nreqs), dim3(64), use_lds ? bytes : 0, stream,
                       static_cast<const uint8_t *>(blob), d_reqs, nreqs, use_lds, use_lds ? blob_lds : 0u);
    return hipGetLastError();
}

extern "C" hipError_t
sre_launch_pike_exec(const void *blob, size_t blob_bytes, const sre_dev_req_t *d_reqs, uint32_t nreqs,
                     size_t ctx_bytes, hipStream_t stream)
{
    return vm_exec_launch(sre_k_pike_exec, blob, blob_bytes, d_reqs, nreqs, ctx_bytes, stream);
}

extern "C" hipError_t
sre_launch_thompson_exec(const void *blob, size_t blob_bytes, const sre_dev_req_t *d_reqs, uint32_t nreqs,
                         size_t ctx_bytes, hipStream_t stream)
{
    return vm_exec_launch(sre_k_thompson_exec, blob, blob_bytes, d_reqs, nreqs, ctx_bytes, stream);
}
