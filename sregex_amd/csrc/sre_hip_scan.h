/*
 * sre_hip_scan.h — device tables, per-segment summaries and launchers of the
 * table-driven segment-parallel scanner (sre_hip_scan.hip).
 */
#ifndef SRE_HIP_SCAN_H
#define SRE_HIP_SCAN_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#define SRE_CEILING_GRID   2048

/* ---- fast table, staged into LDS: one 32-bit entry per (state, 8-bit index).
 * The index packs the byte CLASSES of `stride` consecutive input bytes
 * (class_bits each, stride * class_bits == 8), so one dependent lookup advances
 * the automaton by 1, 2, 4 or 8 bytes; with more than 16 classes the index is
 * the byte itself (stride 1).
 *   bits 31..10  byte offset of the next state's row (state * 1024)
 *   bit  0       SLOW: a sub-step carries a match event or kills the list
 *   bits 1..4    COUNT mode: matches completed inside this step, each followed
 *                by a restart at the next byte (the entry points at the state
 *                reached after the last restart)
 * (global-memory format; sre_k_scan re-packs an entry for LDS as
 *  [count : 8][flags : 8][LDS address of the next row : 16]) */
#include "sre_scan_fast.h"     /* SRE_FAST_*, SRE_STATE_FRESH */

/* In LDS the fast table has extra rows behind the automaton's own: one TRAP row (every
 * entry points back into it, flagged SLOW; SLOW entries of the other rows point there,
 * so a chain of lookups needs no per-step flag test: the flag is still there at its
 * end) and up to SRE_SCAN_MAX_SHADOWS SHADOW rows: a copy of a state's row whose
 * STABLE entries point to the copy itself, everything else to the ordinary rows.  A
 * lane that starts a stretch in a shadow row and is still in one at its end has seen
 * STABLE steps only. */
#define SRE_SCAN_MAX_SHADOWS  2u
/* rows (states + trap + shadows, 1 KiB each) must end below 64 KiB of LDS, behind the
 * kernel's static LDS: the class-map tables (512 bytes per byte position of an index)
 * and the table header */
#define SRE_SCAN_MAX_ROWS(bits)   ((65536u - 768u - ((bits) == 8 ? 512u : (8u / (bits)) * 512u)) / 1024u)
#define SRE_SCAN_MAX_STATES       62u     /* with 8 class bits; fewer with narrower classes */
#define SRE_SCAN_BLOCK        256u    /* lanes = segments per workgroup */
#define SRE_SCAN_LDS_LIMIT    (128u * 1024u)  /* dynamic LDS a scan workgroup may ask for (160 KiB per CU) */
#define SRE_CAPTURE_MAX_GRID   16384u          /* workgroups of the capture walker: beyond that they take their streams in turn */
#define SRE_CAPTURE_LANE_STREAMS 4096u         /* batches of that many streams are walked by lanes (64 per workgroup) */
#define SRE_CAPTURE_LDS_LIMIT  (144u * 1024u)  /* ... and the capture walker / lineage kernels (one workgroup per CU then) */
#define SRE_SCAN_ROUND        64u     /* bytes a lane consumes per LDS round */
#define SRE_SCAN_LINE         128u    /* staging granule: whole lines, half a wave per stage; also the warm-up */
#define SRE_SCAN_SEG_ALIGN    256u    /* segments are a multiple of the line size */

/* device-only event kind: a DONE (match end = pos + 1) whose match is empty */
#define SRE_SCAN_NINIT 4u            /* = SRE_DFA_NINIT: initial lists, and their pseudo transitions behind the
                                       real ones (sre_dfa.cpp) */
#define SRE_DEV_EV_DONE_EMPTY 3
#define SRE_DEV_EV_POP_FULL   4     /* a popped MATCH thread whose match is NOT empty (behind a look-ahead
                                       assertion: foo$); plain SRE_DFA_EV_POP on the device = empty */

/* full transition record (global memory; slow path and lineage kernels) */
typedef struct {
    uint32_t next;
    uint8_t  kind;          /* SRE_DFA_EV_*, or SRE_DEV_EV_DONE_EMPTY */
    uint8_t  src;           /* matching thread's index in the old list */
    uint16_t regex;
    uint32_t lin_off;       /* per new thread: parent / saves */
    uint16_t lin_n;
    uint8_t  skipped;
    uint8_t  pad;           /* 1: the lineage map at lin_off is idempotent (sre_scan_host.cpp) */
    uint64_t saves;         /* DONE: slots written on the way to MATCH (value pos + 1) */
    uint64_t early;         /* slots written by a look-ahead splice in front of the event (value pos) */
} sre_dev_trans_t;

typedef struct {
    uint32_t nstates, ncls, nslots, max_threads;
    uint32_t init[4];                   /* SRE_DFA_INIT_* -> state */
    uint32_t word_restart;              /* init[RESTART_WORD] differs: a re-armed search looks at the word-ness
                                           of the byte in front of it (restart_variant) */
    int32_t  mode;                      /* SRE_HIP_* */
    uint32_t fast_bytes;                /* nstates * 1024 */
    uint32_t stride, class_bits;        /* input bytes per fast-table step */
    const uint32_t        *fast;        /* [nstates][256] */
    const uint32_t        *fast_plain;  /* same without COUNT's folded restarts (== fast otherwise) */
    const uint8_t         *cls;         /* [256] */
    const sre_dev_trans_t *trans;       /* [nstates][ncls + 1], then 3 pseudo rows for the initial closures */
    const uint16_t        *trans2;      /* [nstates][ncls + 1] next | kind << 8: all the scan kernel's exact path
                                           needs of a transition (2 bytes instead of 40 in its LDS) */
    const uint8_t         *lin_parent;
    const uint64_t        *lin_saves;
    const uint64_t        *lin_early;   /* NULL unless the program has look-ahead assertions: slots a
                                           splice wrote before the byte was consumed (value pos) */
    const uint8_t         *lin_flags;   /* per new thread: bit0 its closure path saved a slot,
                                           bit1 it is the ".*?" ANY thread (pc 1) */
    uint32_t               lin_total, list_total;  /* entries of lin_* / list_pcs */
    const uint8_t         *state_flags; /* [nstates] bit0 matched, bits 1-2 seen_start (2 = reached by a skip) */
    const uint32_t        *list_off;    /* [nstates + 1] */
    const uint32_t        *list_pcs;
    const uint32_t        *multi_ncaps; /* [nregexes] */
    uint32_t nregexes;
    uint32_t any_fresh;                 /* COUNT tables: some state is SRE_STATE_FRESH (entries may carry SRE_FAST_EVT) */
    uint32_t nshadow;                   /* shadow rows in the LDS fast table (FIRST / Thompson tables) */
    uint32_t fast_rows;                 /* rows of the scan kernel's LDS copy: nstates + 1 (trap) + nshadow */
    uint32_t wide;                      /* the staging tile holds 16-bit pre-scaled indices (sre_hip_tile.h): always with
                                           <= 2 class bits; with 4 in COUNT mode when three workgroups per CU still fit */
    uint8_t  shadow_state[SRE_SCAN_MAX_SHADOWS];    /* the state each of them copies */
    const uint8_t         *unskip;      /* [nstates] sre_dfa_t.unskip: what a chunk boundary makes of a state */
    const uint16_t        *neutral;     /* [nstates] bit j: thread j of the state's list descends from itself,
                                           without a save, in every STABLE step of the state (0: none) */
} sre_scan_tables_t;

/* what one lane learnt about its segment */
typedef struct {
    uint32_t s_in;          /* state assumed at the segment start (after warm-up) */
    uint32_t s_out;         /* state at the segment end */
    uint32_t flags;         /* SRE_SUM_* */
    uint32_t pad;
    /* SRE_SUM_PENDING: the search in flight at the segment end holds a match */
    uint32_t pe_state, pe_sym;      /* state before the event's transition, its symbol */
    int64_t  pe_pos, pe_sp;         /* event position; start of its search (-1 unknown) */
    /* SRE_SUM_LASTEV: FIRST: last match event seen here; COUNT: last completed match */
    uint32_t lm_state, lm_sym;
    int64_t  lm_pos, lm_sp;
    /* a known state shortly before that event (start of its 64-byte round), so
     * the capture walker need not replay the segment: -1 if none */
    int64_t  lm_apos, pe_apos;
    uint32_t lm_astate, pe_astate;
    int64_t  term_pos;      /* position where the scan of this stream ended, -1 none */
    int64_t  count;         /* COUNT: searches completed with a match in this segment */
    int64_t  cur_sp;        /* start of the search in flight at the segment end, -1 unknown */
    /* COUNT, SRE_SUM_IN_PENDING: the pending match the lane believed it ENTERED with (from its
     * warm-up, or the exact carry).  Equal entry states do not imply equal pending events, so the
     * chain check compares this with the predecessor's pe_* as well. */
    int64_t  in_pe_pos;
    uint32_t in_pe_state, in_pe_sym;
    /* FIRST: the byte steps [0, stable_until) of the segment, counted from its start, were
     * STABLE steps of the entry state s_in, and so were the steps [stable_from, length) of
     * the exit state s_out (whole 64-byte rounds; stable_until == length: the whole segment,
     * flag SRE_SUM_STABLE). */
    uint32_t stable_until, stable_from;
} sre_seg_summary_t;

/* bits: [1] SRE_SUM_TERM [2] SRE_SUM_LASTEV [3] cur_sp >= 0 [4] not SRE_SUM_STABLE (as verify_one_stream keeps them) */
typedef struct sre_seg_digest_s {
    uint32_t s_in, s_out, bits, count;
} sre_seg_digest_t;

#define SRE_SUM_PENDING   1u
#define SRE_SUM_TERM      2u   /* the scan of this stream ended inside this segment */
#define SRE_SUM_LASTEV    4u
#define SRE_SUM_ERROR     8u   /* COUNT: ... and the iteration ended with SRE_ERROR */
#define SRE_SUM_IN_PENDING 16u /* COUNT: the lane entered its segment holding a pending match (in_pe_*) */
#define SRE_SUM_STABLE    32u  /* every byte step of the segment was a STABLE step of s_in == s_out */

/* COUNT: an empty match ended on the segment's last byte boundary, so the
 * caller's one-byte skip (sre_vm_pike.c:179-196) falls on the first byte of the
 * NEXT segment.  s_out carries this bit, which no assumed entry state has: the
 * chain check fails there and the next lane is re-run with the exact carry. */
#define SRE_STATE_SKIP    0x40000000u

/* lineage of one segment (sre_k_lineage_maps): for the thread at index j of the
 * list at the segment's END, which thread of the list at its START it descends
 * from — valid to jump over the segment only if neither flag bit j is set */
typedef struct {
    uint64_t anc;           /* 16 nibbles */
    uint16_t saved;         /* bit j: the lineage wrote a capture slot inside the segment */
    uint16_t stop;          /* bit j: it passes through the ".*?" thread or a skip re-seed */
    uint32_t pad;
} sre_seg_lineage_t;

#define SRE_LINEAGE_BLOCK  256u     /* segments composed into one block map */
#define SRE_WALK_BUDGET    256      /* positions of plain backward walk (~2 us each) before asking for the maps */

/*
 * State of ONE stream between the chunks of a streaming scan (sre_k_stream_tail): what
 * the reference's context keeps in its thread lists (sre_vm_pike.c:47-76) — the ordered
 * list is the automaton state, and every listed thread's capture vector is carried by
 * value, resolved at the end of each chunk by the backward lineage walk.
 */
#define SRE_STREAM_MAX_THREADS 64u      /* threads of the carried list (BASELINE configs[2]: 37) */
#define SRE_STREAM_MAX_SLOTS   64u
typedef struct {
    uint32_t state;             /* automaton state in front of the next byte; 0: no search under way */
    uint32_t has_pending;       /* a match is pending (the list lives on) */
    int64_t  pending_regex;
    int64_t  pending_vec[SRE_STREAM_MAX_SLOTS];
    int64_t  caps[SRE_STREAM_MAX_THREADS][SRE_STREAM_MAX_SLOTS];
    int64_t  caps_next[SRE_STREAM_MAX_THREADS][SRE_STREAM_MAX_SLOTS];    /* scratch of sre_k_stream_tail */
} sre_stream_ctx_t;

/* sre_stream_result_t.rc: the chunk's lanes were not all verified; run the fix-up rounds */
#define SRE_VERIFY_ONE_SEGS 8192u    /* most segments the tail / capture kernels check themselves (verify != 0) */
#define SRE_STREAM_UNSETTLED (-100)
/* ... set by the host before the launch: the tail kernel has not published its result yet */
#define SRE_STREAM_PENDING   (-101)

/* result of one streaming exec, written to host-visible memory */
typedef struct {
    int64_t  rc;                /* regex id, SRE_AGAIN, SRE_DECLINED, SRE_ERROR */
    int64_t  has_pending, pending[2];
    int64_t  ev_in_chunk;       /* a match event happened inside this chunk (sre_vm_pike.c:586-601) */
    int64_t  poisoned;          /* match returned with threads still listed at eof (:616-622) */
    int64_t  next_state;        /* host shadow of sre_stream_ctx_t.state */
    int64_t  ev_slot1;          /* ev_in_chunk: slot 1 of that match's internal vector (the end of a match of
                                   regex 0, else -1): what last_matched_pos holds when the call returns */
    int64_t  ov[SRE_STREAM_MAX_SLOTS];
} sre_stream_result_t;

/* per-stream outcome of verify + reduce */
typedef struct {
    int64_t  first_bad;     /* first segment whose assumed entry state was wrong, or nseg */
    int64_t  limit;         /* segments [0, limit) are verified and sufficient */
    int64_t  count;
    int64_t  rc;            /* regex id / SRE_DECLINED / SRE_ERROR */
    int64_t  ev_pos, ev_sp;
    uint32_t ev_state, ev_sym;
    int64_t  ev_apos;       /* anchor: a known state shortly before the event, -1 none */
    uint32_t ev_astate;
    uint32_t valid_from;    /* set by sre_k_captures: first segment whose recorded entry state belongs
                               to the event's search (see Tracer::entry_state) */
    int64_t  ev_seg;        /* segment holding the event */
    int64_t  unst_seg;      /* last segment in front of ev_seg that is not SRE_SUM_STABLE, -1 none:
                               the segments between the two are stable, all in one state */
    int64_t  unst_end;      /* ... and the last one in front of the stream's LAST segment */
    int32_t  done;          /* 1: result final, 0: needs a fix-up round from first_bad */
    int32_t  error;         /* COUNT: the iteration ended with SRE_ERROR */
    int32_t  need_maps;     /* set by sre_k_captures: lineage too long for the plain walk */
    int32_t  pad;
} sre_stream_status_t;

typedef struct {
    const uint8_t *const *streams;  /* device array of device pointers */
    const uint64_t *lens;
    const uint64_t *seg_first;      /* [nstreams + 1] prefix of segment counts */
    uint32_t nstreams;
    uint32_t seg_bytes;
    uint64_t nsegs;
    uint32_t init_variant;          /* SRE_DFA_INIT_* of the search that starts at offset 0 */
    uint32_t flags;                 /* SRE_GEOM_* */
    uint32_t entry_state;           /* SRE_GEOM_CONTINUES: the automaton state in front of offset 0 */
    /* SRE_GEOM_ONE: the only stream, in the kernel arguments themselves (the three arrays
     * are not read: a call on one buffer uploads nothing) */
    const uint8_t *one_ptr;
    uint64_t one_len;
    /* NFA tier, find-all rounds (sre_hip_batch.cpp nfa_count_rounds): per stream, instead of
     * init_variant / flags — SRE_SFLAG_*: which initial list the stream's offset 0 starts with, which
     * one the search it belongs to began with (the reference's snapshot, sre_vm_pike.c:218-229), how
     * the reference arrives at offset 0 (bit 0 of sre_nfa_summary_t.last_clean's mode), whether
     * the buffer ends before the stream does.  NULL: init_variant / flags hold for every stream. */
    const uint8_t *sflags;
    /* FIRST / Thompson: what the chain check of a small batch reads of every summary, 16 bytes a segment
     * (sre_seg_digest_t; NULL: none is written).  One workgroup checks up to 8192 segments: out of the
     * 144-byte summaries that was 14 us for the 4096 segments of a 1 MiB chunk. */
    struct sre_seg_digest_s *digest;
} sre_scan_geom_t;
#define SRE_SFLAG_INIT(f)    ((f) & 3u)
#define SRE_SFLAG_SNAP(f)    (((f) >> 2) & 3u)
#define SRE_SFLAG_MODE(f)    (((f) >> 4) & 1u)
#define SRE_SFLAG_NO_EOF     32u

#define SRE_GEOM_CONTINUES 1u       /* the buffers are chunks of streams whose search began earlier */
#define SRE_GEOM_NO_EOF    2u       /* more chunks follow: no EOF step at the end of the buffer */
#define SRE_GEOM_ONE       4u       /* nstreams == 1, described by one_ptr / one_len / nsegs */

#ifdef __HIPCC__
static __device__ inline const uint8_t *
geom_ptr(const sre_scan_geom_t &G, uint32_t s)
{
    return (G.flags & SRE_GEOM_ONE) ? G.one_ptr : G.streams[s];
}
static __device__ inline uint64_t
geom_len(const sre_scan_geom_t &G, uint32_t s)
{
    return (G.flags & SRE_GEOM_ONE) ? G.one_len : G.lens[s];
}
/* first segment of stream i (i == nstreams: the total) */
static __device__ inline uint64_t
geom_first(const sre_scan_geom_t &G, uint32_t i)
{
    return (G.flags & SRE_GEOM_ONE) ? (i ? G.nsegs : 0) : G.seg_first[i];
}
#endif

#ifdef __cplusplus
extern "C" {
#endif
hipError_t sre_launch_gen_data(void *d_dst, uint64_t n, uint64_t tail_len, const void *d_tail,
    hipStream_t stream);
hipError_t sre_launch_read_ceiling(const void *d_src, uint64_t n, uint32_t *d_sink,
    hipStream_t stream);
hipError_t sre_launch_read_pattern(const void *d_src, uint64_t n, uint32_t seg_bytes,
    uint32_t tile, uint32_t lds_bytes, uint32_t *d_sink, hipStream_t stream);

/* dynamic LDS one scan workgroup needs (fast table + class map + tile) */
size_t sre_scan_lds_bytes(const sre_scan_tables_t *h_tab);
int sre_scan_blocks_per_cu(const sre_scan_tables_t *h_tab);

/* control pass over every stream (d_lo == NULL), or a fix-up round: of every stream whose
 * status (d_lo, from the previous round's sre_launch_verify) is not done, the segments from
 * first_bad on; segment first_bad enters with the exact carry of summaries[first_bad - 1]. */
/* d_entry (optional, with lo): exact entry state per segment, 0xff = none (sre_launch_exact_entries) */
hipError_t sre_launch_scan(const sre_scan_tables_t *d_tab, sre_scan_tables_t h_tab,
    sre_scan_geom_t geom, sre_seg_summary_t *d_sum, const sre_stream_status_t *d_lo, const uint8_t *d_entry,
    hipStream_t stream);
hipError_t sre_launch_exact_entries(const sre_scan_tables_t *d_tab, sre_scan_tables_t h_tab,
    sre_scan_geom_t geom, const sre_seg_summary_t *d_sum, const sre_stream_status_t *d_status,
    uint8_t *d_fn, uint8_t *d_comp, uint8_t *d_chunk_entry, uint8_t *d_entry, hipStream_t stream);
/* streaming: finish one chunk of ONE stream (geom.nstreams == 1) after the scan: the match
 * if the search ended, else the carried state for the next chunk and the temporary /
 * pending captures of SRE_AGAIN.  `base` = absolute offset of the chunk.  verify != 0: the
 * chain check runs inside the same kernel (no sre_launch_verify in front) and writes
 * *d_status; else *d_status is what sre_launch_verify left. */
hipError_t sre_launch_stream_tail(const sre_scan_tables_t *d_tab, sre_scan_tables_t h_tab,
    sre_scan_geom_t geom, const sre_seg_summary_t *d_sum, sre_stream_status_t *d_status,
    uint16_t *d_scratch, sre_stream_ctx_t *d_ctx, sre_stream_result_t *result, int64_t base, int eof,
    uint32_t ovec_slots, int verify, hipStream_t stream);
/* copy nwords 64-bit words from pinned, device-mapped host memory to the device with a kernel */
hipError_t sre_launch_upload_words(const uint64_t *h_src_mapped, uint64_t *d_dst, uint32_t nwords, hipStream_t stream);
size_t sre_scan_verify_acc_bytes(uint32_t nstreams);
hipError_t sre_scan_verify_acc_init(void *d_acc, uint32_t nstreams, hipStream_t stream);
hipError_t sre_launch_verify(sre_scan_tables_t h_tab, sre_scan_geom_t geom,
    const sre_seg_summary_t *d_sum, void *d_acc, sre_stream_status_t *d_status,
    hipStream_t stream);
/* captures of each stream's final match -> records [rc, count, ovector].
 * status[s].need_maps is set to 1 when the lineage is too long for the plain walk and the
 * per-segment maps are required (then call sre_launch_lineage and this again
 * with use_maps = 1).  verify != 0 (one stream of at most SRE_VERIFY_ONE_SEGS segments, not
 * COUNT): the chain check runs inside, no sre_launch_verify in front. */
hipError_t sre_launch_captures(const sre_scan_tables_t *d_tab, sre_scan_tables_t h_tab,
    sre_scan_geom_t geom, const sre_seg_summary_t *d_sum, sre_stream_status_t *d_status,
    uint16_t *d_scratch, int64_t *d_records, uint32_t ovec_slots,
    const sre_seg_lineage_t *d_maps, const sre_seg_lineage_t *d_blocks, int use_maps,
    int verify, hipStream_t stream);
/* ancestor maps of every segment in front of a flagged stream's match, and
 * their 256-segment compositions */
hipError_t sre_launch_lineage(const sre_scan_tables_t *d_tab, sre_scan_tables_t h_tab,
    sre_scan_geom_t geom, const sre_seg_summary_t *d_sum, const sre_stream_status_t *d_status,
    sre_seg_lineage_t *d_maps, sre_seg_lineage_t *d_blocks, hipStream_t stream);
#ifdef __cplusplus
}
#endif
#endif
