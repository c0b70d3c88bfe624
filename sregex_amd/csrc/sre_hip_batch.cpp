/*
 * sre_hip_batch.cpp — the additive device-resident batched API (sregex_hip.h).
 *
 * A scanner binds one compiled program, one mode and one engine:
 *   ENGINE_VM    exact bytecode VM kernel, one lane per stream (sre_hip_vm.hip)
 *   ENGINE_SCAN  table-driven segment-parallel scanner        (sre_hip_scan.hip)
 *   ENGINE_NFA   bit-parallel NFA scanner + exact VM window   (sre_hip_nfa.hip)
 * Stream pointers/lengths are staged to the device per call; results come back
 * as fixed-stride records.  Everything enqueues on the caller's hipStream_t so
 * a driver can bracket the scan with its own events.
 */
#include <sregex_hip.h>
#include "sre_hip_runtime.h"
#include "sre_hip_scan.h"
#include "sre_scan_host.h"
#include "sre_dfa.h"
#include "sre_nfa.h"
#include "sre_hip_nfa.h"
#include "sre_pwave.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <chrono>
#include <vector>

/* speculative fix-up rounds before a FIRST / Thompson scan computes exact entry states */
#define SRE_SPECULATIVE_FIXUPS 2

struct sre_hip_scanner_s {
    sre_program_t     *prog;
    sre_hip_program_s *dp;
    int                mode, engine;
    char               kernel_name[64];
    uint32_t           ovec_slots;      /* 2 * (max_ncaps + 1) */
    /* per-call staging, grown on demand */
    size_t             cap_streams;
    const void       **d_ptrs;
    uint64_t          *d_lens;
    int64_t           *d_records;
    void              *d_ctx;           /* ENGINE_VM: per-stream VM state */
    size_t             ctx_stride, ctx_cap;
    uint64_t          *h_lens;          /* pinned staging */
    const void       **h_ptrs;
    size_t             last_n;
    hipStream_t        last_stream;
    /* ENGINE_SCAN */
    sre_dfa_t                *dfa;
    sre_scan_device_tables_t *tab;
    uint32_t                  seg_override;     /* 0 = automatic */
    uint64_t                  seg_cap_env;
    sre_scan_geom_t           geom;
    uint64_t                 *d_seg_first, *h_seg_first;
    sre_seg_summary_t        *d_sum;
    sre_seg_digest_t         *d_digest;                 /* SRE_VERIFY_ONE_SEGS entries, see sre_scan_geom_t.digest */
    size_t                    sum_cap;
    sre_stream_status_t      *d_status, *h_status;
    void                     *d_acc;
    int64_t                  *d_lo, *h_lo;
    uint16_t                 *d_scratch;
    size_t                    scratch_cap;
    sre_seg_lineage_t        *d_maps, *d_blocks;
    size_t                    maps_cap;
    uint8_t                  *d_fn;             /* segment transition functions + compositions + entry states */
    size_t                    fn_cap;
    int                       exact_passes;     /* of the last scan (diagnostics) */
    int                       lineage_passes;   /* of the last scan (diagnostics) */
    uint32_t                  next_init_variant;    /* compat path: a re-armed context's search */
    int                       blocks_per_cu;
    hipStream_t               tail_stream;      /* sre_hip_scanner_set_tail_stream */
    bool                      tail_stream_set;
    uint32_t                  geom_one;         /* SRE_GEOM_ONE when the batch in flight is one stream in the kernel arguments */
    int                       fixup_rounds;     /* of the last scan (diagnostics) */
    hipEvent_t                ev0, ev1;         /* around the dominant scan kernel */
    int                       ev_valid;
    /* results of a call travel to pinned host memory as part of the enqueued
     * work; results() waits for this event only, so a caller can queue the
     * next buffer (on another scanner) before collecting */
    sre_int_t                *h_records;
    hipEvent_t                ev_done;
    /* d_ptrs / d_lens / d_seg_first live in ONE device block (h_* in one pinned
     * block) and d_records / d_status in another, laid out per call, so that a
     * call costs one small copy in and one out */
    uint64_t                 *d_in, *h_in;
    unsigned char            *d_out, *h_out;
    /* ENGINE_NFA */
    sre_nfa_t                *nfa;
    sre_nfa_tables_t          ntab;             /* device pointers inside */
    sre_nfa_sa_tables_t       satab;            /* the shift-and form (sre_nfa.h), when the program has one */
    bool                      use_sa;
    /* find-all counting on the NFA tier (nfa_count_rounds) */
    struct NfaCount          *cnt;
    uint8_t                  *d_sflags, *h_sflags;      /* per stream of a round: SRE_SFLAG_* */
    sre_nfa_count_req_t      *d_creq, *h_creq;
    size_t                    cnt_cap;
    int                       count_rounds;             /* of the last call (diagnostics) */
    sre_pwave_hdr_t          *h_pwave;                  /* the wave form of the exact window's step (sre_pwave.h), or NULL */
    void                     *d_pwave;
    size_t                    layout_n;                 /* streams the staging blocks are laid out for */
    sre_nfa_summary_t        *d_nsum;
    size_t                    nsum_cap;
    uint64_t                 *d_belief;
    uint8_t                  *d_bvalid;
    uint64_t                 *d_nmat;           /* the segments' singleton exit sets (exact entry sets), on demand */
    size_t                    nmat_cap;
    void                     *d_nacc;
    sre_nfa_status_t         *d_nstatus, *h_nstatus;
};

/* one stream of a find-all count on the NFA tier */
struct NfaCountStream {
    const uint8_t *base;
    uint64_t       n;
    uint64_t       cur;         /* where the current search began (the previous match's end) */
    uint64_t       q;           /* a clean position of that search: where the next scanned buffer starts */
    uint64_t       horizon;     /* bytes scanned per round */
    uint32_t       var_cur, var_q, mode_q;
    int64_t        count;
    bool           done, error;
    std::vector<sre_int_t> last;    /* record of the last match */
};
struct NfaCount {
    std::vector<NfaCountStream> st;
    std::vector<size_t>         active;
};

static void
scanner_release(void *data)
{
    sre_hip_scanner_t *sc = static_cast<sre_hip_scanner_t *>(data);
    delete sc->cnt;
    free(sc->h_pwave);
    if (sc->d_pwave) (void) hipFree(sc->d_pwave);
    if (sc->d_sflags) (void) hipFree(sc->d_sflags);
    if (sc->h_sflags) (void) hipHostFree(sc->h_sflags);
    if (sc->d_creq) (void) hipFree(sc->d_creq);
    if (sc->h_creq) (void) hipHostFree(sc->h_creq);
    if (sc->d_in) (void) hipFree(sc->d_in);
    if (sc->h_in) (void) hipHostFree(sc->h_in);
    if (sc->d_out) (void) hipFree(sc->d_out);
    if (sc->h_out) (void) hipHostFree(sc->h_out);
    if (sc->d_ctx) (void) hipFree(sc->d_ctx);
    if (sc->d_sum) (void) hipFree(sc->d_sum);
    if (sc->d_digest) (void) hipFree(sc->d_digest);
    if (sc->d_acc) (void) hipFree(sc->d_acc);
    if (sc->d_lo) (void) hipFree(sc->d_lo);
    if (sc->h_lo) (void) hipHostFree(sc->h_lo);
    if (sc->d_scratch) (void) hipFree(sc->d_scratch);
    if (sc->d_fn) (void) hipFree(sc->d_fn);
    if (sc->d_maps) (void) hipFree(sc->d_maps);
    if (sc->d_blocks) (void) hipFree(sc->d_blocks);
    if (sc->ev0) (void) hipEventDestroy(sc->ev0);
    if (sc->ev1) (void) hipEventDestroy(sc->ev1);
    if (sc->ev_done) (void) hipEventDestroy(sc->ev_done);
    if (sc->d_nsum) (void) hipFree(sc->d_nsum);
    if (sc->d_belief) (void) hipFree(sc->d_belief);
    if (sc->d_bvalid) (void) hipFree(sc->d_bvalid);
    if (sc->d_nmat) (void) hipFree(sc->d_nmat);
    if (sc->d_nacc) (void) hipFree(sc->d_nacc);
    if (sc->ntab.accept) (void) hipFree(const_cast<uint64_t *>(sc->ntab.accept));
    if (sc->ntab.follow) (void) hipFree(const_cast<uint64_t *>(sc->ntab.follow));
    if (sc->ntab.expand) (void) hipFree(const_cast<uint64_t *>(sc->ntab.expand));
    if (sc->ntab.kind) (void) hipFree(const_cast<uint8_t *>(sc->ntab.kind));
    if (sc->satab.accept) (void) hipFree(const_cast<uint64_t *>(sc->satab.accept));
    if (sc->satab.lut) (void) hipFree(const_cast<uint64_t *>(sc->satab.lut));
    if (sc->satab.expand) (void) hipFree(const_cast<uint64_t *>(sc->satab.expand));
    sre_nfa_free(sc->nfa);
    sre_scan_tables_release(sc->tab);
    sre_dfa_free(sc->dfa);
    free(sc);
}

/* device copies of the bit-parallel tables, padded to the slice count the
 * kernel variant is compiled for */
static int
nfa_upload(sre_hip_scanner_t *sc)
{
    const sre_nfa_t *n = sc->nfa;
    const uint32_t   ns = n->nslices;           /* already rounded to a compiled variant */
    std::vector<uint64_t> fol((size_t) ns * 256, 0), exp(16 * 256, 0);
    memcpy(fol.data(), n->follow.data(), n->follow.size() * sizeof(uint64_t));
    if (n->nassert) memcpy(exp.data(), n->expand.data(), exp.size() * sizeof(uint64_t));
    uint64_t *d_acc = NULL, *d_fol = NULL, *d_exp = NULL;
    uint8_t  *d_kind = NULL;
    SRE_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d_acc), 256 * sizeof(uint64_t)));
    sc->ntab.accept = d_acc;
    SRE_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d_fol), fol.size() * sizeof(uint64_t)));
    sc->ntab.follow = d_fol;
    SRE_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d_exp), exp.size() * sizeof(uint64_t)));
    sc->ntab.expand = d_exp;
    SRE_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d_kind), 256));
    sc->ntab.kind = d_kind;
    SRE_HIP_TRY(hipMemcpy(d_acc, n->accept, 256 * sizeof(uint64_t), hipMemcpyHostToDevice));
    SRE_HIP_TRY(hipMemcpy(d_fol, fol.data(), fol.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
    SRE_HIP_TRY(hipMemcpy(d_exp, exp.data(), exp.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
    SRE_HIP_TRY(hipMemcpy(d_kind, n->kind, 256, hipMemcpyHostToDevice));
    sc->ntab.nbits = n->nbits;
    sc->ntab.nslices = ns;
    sc->ntab.nassert = n->nassert;
    for (int v = 0; v < 3; v++) sc->ntab.init[v] = n->init[v];
    sc->ntab.any_bits = n->any_bits;
    sc->ntab.match_bits = n->match_bits;
    if (n->sa) {
        /* the shift-and form: its own accept table and the exception lookups */
        const sre_nfa_sa_t *a = n->sa;
        sre_nfa_sa_tables_t &t = sc->satab;
        uint64_t *d_sacc = NULL, *d_lut = NULL;
        SRE_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d_sacc), 256 * sizeof(uint64_t)));
        t.accept = d_sacc;
        SRE_HIP_TRY(hipMemcpy(d_sacc, a->accept, 256 * sizeof(uint64_t), hipMemcpyHostToDevice));
        if (a->nlut) {
            SRE_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d_lut), a->lut.size() * sizeof(uint64_t)));
            t.lut = d_lut;
            SRE_HIP_TRY(hipMemcpy(d_lut, a->lut.data(), a->lut.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
        }
        if (a->nassert) {
            /* the expansion table, compacted to the assertion bits (bits 0 .. of the mask) */
            const size_t          per = (size_t) 1 << a->nassert;
            std::vector<uint64_t> ex(16 * per);
            for (size_t ctx = 0; ctx < 16; ctx++) {
                for (size_t v = 0; v < per; v++) ex[ctx * per + v] = a->expand[ctx * 256 + v];
            }
            uint64_t *d_ex = NULL;
            SRE_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d_ex), ex.size() * sizeof(uint64_t)));
            t.expand = d_ex;
            SRE_HIP_TRY(hipMemcpy(d_ex, ex.data(), ex.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
            t.kind = sc->ntab.kind;
            t.nassert = a->nassert;
        }
        t.w64 = a->w64;
        t.carry = a->carry;
        t.masked = a->masked;
        t.evacc = a->evacc;
        t.nlut = a->nlut;
        /* v_perm_b32 selector: result byte k = byte hot[k] of the 64-bit mask {hi, lo}; selector
         * values 0..3 pick a byte of the second source (lo), 4..7 of the first (hi) */
        t.perm = 0;
        for (uint32_t k = 0; k < 4; k++) t.perm |= (k < a->nlut ? a->hot[k] : 0u) << (8 * k);
        for (int v = 0; v < 3; v++) t.init[v] = a->init[v];
        t.seed = a->seed;
        t.any_bits = a->any_bits;
        t.match_bits = a->match_bits;
        t.msrc = a->msrc;
        t.valid = a->valid;
        t.self = a->self;
        t.shift_src = a->shift_src;
        sc->use_sa = true;
    }
    return 0;
hip_failed:
    return -1;
}

/* one pass of the set kernel the scanner's program runs on */
static hipError_t
nfa_launch_scan(sre_hip_scanner_t *sc, const int64_t *d_lo, const uint64_t *d_belief, const uint8_t *d_bvalid,
                hipStream_t stream)
{
    if (sc->use_sa) return sre_launch_nfa_sa_scan(sc->satab, sc->geom, sc->d_nsum, d_lo, d_belief, d_bvalid, stream);
    return sre_launch_nfa_scan(sc->mode == SRE_HIP_THOMPSON ? SRE_HIP_THOMPSON : SRE_HIP_PIKE_FIRST, sc->ntab, sc->geom,
                               sc->d_nsum, d_lo, d_belief, d_bvalid, stream);
}

extern "C" SRE_API int
sre_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" SRE_API int
sre_hip_set_device(int ordinal)
{
    hipError_t e = hipSetDevice(ordinal);
    return e == hipSuccess ? 0 : sre_hip_fail("hipSetDevice", e);
}

static sre_hip_scanner_t *scanner_create(sre_pool_t *pool, sre_program_t *prog, int mode, int engine, int chunk_twins);

extern "C" SRE_API sre_hip_scanner_t *
sre_hip_scanner_create(sre_pool_t *pool, sre_program_t *prog, int mode, int engine)
{
    return scanner_create(pool, prog, mode, engine, 0);
}

/* The scanner of a stream that is fed in CHUNKS (sre_vm_api.cpp): table-driven only, its
 * automaton built with the states a chunk boundary makes of look-ahead lists
 * (sre_dfa.h `rekind`).  NULL when the program is not admitted. */
extern "C" sre_hip_scanner_t *
sre_hip_scanner_create_chunked(sre_pool_t *pool, sre_program_t *prog, int mode)
{
    if (sre_hip_ready() != 0) return NULL;
    return scanner_create(pool, prog, mode, SRE_HIP_ENGINE_AUTO, 1);
}

/* the entry state of the next chunk: `state` is what the previous chunk's tail reported,
 * flags = the context's 0: neither, 1: seen_newline, 2: seen_word; 3: sre_vm_thompson_exec */
extern "C" uint32_t
sre_hip_scanner_chunk_entry(sre_hip_scanner_t *sc, uint32_t state, int flags)
{
    if (sc->dfa == NULL || sc->dfa->rekind.empty() || state >= sc->dfa->nstates || flags < 0 || flags > 3) return state;
    return sc->dfa->rekind[4 * (size_t) state + (size_t) flags];
}

static sre_hip_scanner_t *
scanner_create(sre_pool_t *pool, sre_program_t *prog, int mode, int engine, int chunk_twins)
{
    if (mode < SRE_HIP_THOMPSON || mode > SRE_HIP_PIKE_COUNT) return NULL;
    sre_hip_program_s *dp = sre_hip_program_get(prog);
    if (dp == NULL) return NULL;

    sre_hip_scanner_t *sc = static_cast<sre_hip_scanner_t *>(calloc(1, sizeof(*sc)));
    if (sc == NULL) return NULL;
    sc->prog = prog;
    sc->dp = dp;
    sc->mode = mode;
    uint32_t maxcaps = 0;
    for (uint32_t i = 0; i < prog->nregexes; i++) {
        if (prog->multi_ncaps[i] > maxcaps) maxcaps = prog->multi_ncaps[i];
    }
    sc->ovec_slots = 2 * (maxcaps + 1);
    sc->engine = SRE_HIP_ENGINE_VM;

    if (engine == SRE_HIP_ENGINE_AUTO || engine == SRE_HIP_ENGINE_SCAN) {
        /* compile step: step automaton + device tables (independent of any input) */
        const char *why = NULL;
        sc->dfa = sre_dfa_build2(prog, 4 * SRE_SCAN_MAX_STATES, chunk_twins, &why);
        if (sc->dfa) sc->tab = sre_scan_tables_build(prog, sc->dfa, mode, &why);
        if (sc->tab) {
            sc->engine = SRE_HIP_ENGINE_SCAN;
        } else if (engine == SRE_HIP_ENGINE_SCAN) {
            fprintf(stderr, "[sregex-hip] table-driven scanner not available: %s\n",
                    why ? why : "unknown");
            scanner_release(sc);
            return NULL;
        }
    }
    if (chunk_twins && sc->engine != SRE_HIP_ENGINE_SCAN) {
        scanner_release(sc);
        return NULL;
    }
    if (sc->engine == SRE_HIP_ENGINE_VM && (engine == SRE_HIP_ENGINE_AUTO || engine == SRE_HIP_ENGINE_NFA)) {
        /* the ordered-list automaton is too large (or was not asked for): the
         * bit-parallel form, if the program has one */
        const char *why = NULL;
        sc->nfa = sre_nfa_build(prog, &why);
        if (sc->nfa && mode == SRE_HIP_PIKE_COUNT && (sc->nfa->nassert || sc->nfa->init[1] != sc->nfa->init[2])) {
            /* Find-all on this tier restarts searches in the middle of the stream.  A re-armed search
             * that does not start behind a newline holds the bare ".*?" list as its initial-state
             * snapshot, so the reference's leading-byte skip (sre_vm_pike.c:256-309) fires at every
             * idle position and jumps over newlines whose consumption would have listed the threads
             * behind ^: `^b+` over "b x\n\nb" finds one match, not two.  Thread SETS step every byte and
             * cannot know which positions the reference never visits, so programs whose seeded closure
             * depends on ^ keep the exact VM for find-all (found by the tier's own differential test);
             * so do look-ahead programs, whose re-armed context carries seen_word (:472-473). */
            why = sc->nfa->nassert ? "find-all counting of a look-ahead program the step automaton declines"
                                   : "find-all counting of a program whose initial closure depends on ^ (re-armed searches skip newlines)";
            sre_nfa_free(sc->nfa);
            sc->nfa = NULL;
        }
        if (sc->nfa && nfa_upload(sc) == 0) {
            sc->engine = SRE_HIP_ENGINE_NFA;
            if (mode == SRE_HIP_PIKE_COUNT) sc->cnt = new NfaCount();
            if (mode != SRE_HIP_THOMPSON && getenv("SRE_HIP_NO_PWAVE") == NULL) {
                /* the exact window's step by a wavefront, when the program has the form */
                sc->h_pwave = sre_pwave_build(prog);
                if (sc->h_pwave && !sre_pwave_fits(sc->h_pwave)) {
                    free(sc->h_pwave);
                    sc->h_pwave = NULL;
                }
                if (sc->h_pwave
                    && (hipMalloc(&sc->d_pwave, sc->h_pwave->bytes) != hipSuccess
                        || hipMemcpy(sc->d_pwave, sc->h_pwave, sc->h_pwave->bytes, hipMemcpyHostToDevice) != hipSuccess))
                {
                    if (sc->d_pwave) (void) hipFree(sc->d_pwave);
                    sc->d_pwave = NULL;
                    free(sc->h_pwave);
                    sc->h_pwave = NULL;
                }
            }
        } else if (engine == SRE_HIP_ENGINE_NFA) {
            fprintf(stderr, "[sregex-hip] bit-parallel NFA scanner not available: %s\n",
                    why ? why : "device allocation failed");
            scanner_release(sc);
            return NULL;
        }
    }
    sc->ctx_stride = mode == SRE_HIP_THOMPSON ? dp->thompson_layout.total : dp->pike_layout.total;
    if (sc->engine == SRE_HIP_ENGINE_VM && mode != SRE_HIP_THOMPSON && getenv("SRE_HIP_NO_PWAVE") == NULL) {
        /* the exact VM's Pike scans by a wavefront per stream, when the program has the form (sre_pwave.h) */
        sc->h_pwave = sre_pwave_build(prog);
        if (sc->h_pwave && !sre_pwave_fits(sc->h_pwave)) {
            free(sc->h_pwave);
            sc->h_pwave = NULL;
        }
        if (sc->h_pwave
            && (hipMalloc(&sc->d_pwave, sc->h_pwave->bytes) != hipSuccess
                || hipMemcpy(sc->d_pwave, sc->h_pwave, sc->h_pwave->bytes, hipMemcpyHostToDevice) != hipSuccess))
        {
            if (sc->d_pwave) (void) hipFree(sc->d_pwave);
            sc->d_pwave = NULL;
            free(sc->h_pwave);
            sc->h_pwave = NULL;
        }
    }

    if (sre_pool_add_cleanup(pool, scanner_release, sc) != SRE_OK) {
        scanner_release(sc);
        return NULL;
    }
    return sc;
}

extern "C" SRE_API int
sre_hip_scanner_engine(sre_hip_scanner_t *sc)
{
    return sc->engine;
}

extern "C" SRE_API size_t
sre_hip_scanner_result_slots(sre_hip_scanner_t *sc)
{
    return 2 + (size_t) sc->ovec_slots;
}

extern "C" SRE_API int
sre_hip_scanner_set_segment_bytes(sre_hip_scanner_t *sc, size_t bytes)
{
    if (bytes != 0 && (bytes % 64 != 0 || bytes > (1u << 30))) return -1;
    sc->seg_override = (uint32_t) bytes;
    return 0;
}

extern "C" SRE_API int
sre_hip_scanner_last_fixups(sre_hip_scanner_t *sc)
{
    return sc->fixup_rounds;
}

extern "C" SRE_API int
sre_hip_scanner_last_exact_passes(sre_hip_scanner_t *sc)
{
    return sc->exact_passes;
}

extern "C" SRE_API int
sre_hip_scanner_class_bits(sre_hip_scanner_t *sc)
{
    return sc->engine == SRE_HIP_ENGINE_SCAN ? (int) sc->tab->h.class_bits : 0;
}

extern "C" SRE_API const char *
sre_hip_scanner_kernel_name(sre_hip_scanner_t *sc)
{
    if (sc->kernel_name[0] == 0) {
        if (sc->engine == SRE_HIP_ENGINE_SCAN) {
            snprintf(sc->kernel_name, sizeof(sc->kernel_name), "sre_k_scan<%d, %d, %s, %s>",
                     sc->mode == SRE_HIP_PIKE_COUNT ? 2 : 1, (int) sc->tab->h.class_bits,
                     sc->tab->h.wide ? "true" : "false",
                     sc->mode == SRE_HIP_PIKE_COUNT && sc->tab->h.any_fresh ? "true" : "false");
        } else if (sc->engine == SRE_HIP_ENGINE_NFA) {
            if (sc->use_sa) sre_nfa_sa_kernel_name(&sc->satab, sc->kernel_name, sizeof(sc->kernel_name));
            else sre_nfa_kernel_name(sc->mode == SRE_HIP_THOMPSON ? 0 : 1, sc->ntab.nslices, sc->ntab.nassert != 0,
                                     sc->kernel_name, sizeof(sc->kernel_name));
        } else {
            snprintf(sc->kernel_name, sizeof(sc->kernel_name), "%s",
                     sc->mode == SRE_HIP_THOMPSON ? (sc->dp->has_wave ? "sre_k_thompson_wave_scan" : "sre_k_thompson_scan")
                                                  : sc->d_pwave ? "sre_k_pike_scan_wave" : "sre_k_pike_scan");
        }
    }
    return sc->kernel_name;
}

extern "C" SRE_API int
sre_hip_scanner_last_lineage_passes(sre_hip_scanner_t *sc)
{
    return sc->lineage_passes;
}

extern "C" SRE_API double
sre_hip_scanner_last_kernel_ms(sre_hip_scanner_t *sc)
{
    float ms = 0.0f;
    if (!sc->ev_valid) return -1.0;
    if (hipEventSynchronize(sc->ev1) != hipSuccess) return -1.0;
    if (hipEventElapsedTime(&ms, sc->ev0, sc->ev1) != hipSuccess) return -1.0;
    return (double) ms;
}

extern "C" SRE_API int
sre_hip_scanner_order_after_scan(sre_hip_scanner_t *sc, void *hip_stream)
{
    if (!sc->ev_valid) return 0;
    hipError_t e = hipStreamWaitEvent(static_cast<hipStream_t>(hip_stream), sc->ev1, 0);
    return e == hipSuccess ? 0 : sre_hip_fail("hipStreamWaitEvent", e);
}

extern "C" SRE_API int
sre_hip_scanner_set_tail_stream(sre_hip_scanner_t *sc, void *hip_stream)
{
    sc->tail_stream = static_cast<hipStream_t>(hip_stream);
    sc->tail_stream_set = hip_stream != NULL;
    return 0;
}

extern "C" SRE_API size_t
sre_hip_scanner_last_segment_bytes(sre_hip_scanner_t *sc)
{
    return sc->engine != SRE_HIP_ENGINE_VM ? sc->geom.seg_bytes : 0;
}

static size_t
record_bytes(const sre_hip_scanner_t *sc, size_t n)
{
    return n * (2 + (size_t) sc->ovec_slots) * sizeof(int64_t);
}

static int
scanner_reserve(sre_hip_scanner_t *sc, size_t n)
{
    if (n > sc->cap_streams) {
        if (sc->d_in) (void) hipFree(sc->d_in);
        if (sc->h_in) (void) hipHostFree(sc->h_in);
        if (sc->d_out) (void) hipFree(sc->d_out);
        if (sc->h_out) (void) hipHostFree(sc->h_out);
        if (sc->d_acc) (void) hipFree(sc->d_acc);
        if (sc->d_lo) (void) hipFree(sc->d_lo);
        if (sc->h_lo) (void) hipHostFree(sc->h_lo);
        sc->d_in = sc->h_in = NULL;
        sc->d_out = sc->h_out = NULL;
        sc->d_acc = NULL;
        sc->d_lo = sc->h_lo = NULL;
        sc->cap_streams = 0;
        const size_t in_bytes = (3 * n + 1) * sizeof(uint64_t);
        const size_t out_bytes = record_bytes(sc, n) + n * sizeof(sre_stream_status_t);   /* >= sre_nfa_status_t */
        static_assert(sizeof(sre_nfa_status_t) <= sizeof(sre_stream_status_t), "status block");
        static_assert(sizeof(sre_nfa_status_t) == sizeof(sre_nfa_window_t), "window layout");
        SRE_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&sc->d_in), in_bytes));
        SRE_HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&sc->h_in), in_bytes, 0));
        SRE_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&sc->d_out), out_bytes));
        SRE_HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&sc->h_out), out_bytes, 0));
        SRE_HIP_TRY(hipMalloc(&sc->d_acc, sre_scan_verify_acc_bytes((uint32_t) n)));
        SRE_HIP_TRY(sre_scan_verify_acc_init(sc->d_acc, (uint32_t) n, NULL));
        if (sc->engine == SRE_HIP_ENGINE_NFA) {
            if (sc->d_nacc) (void) hipFree(sc->d_nacc);
            sc->d_nacc = NULL;
            SRE_HIP_TRY(hipMalloc(&sc->d_nacc, sre_nfa_verify_acc_bytes((uint32_t) n)));
            SRE_HIP_TRY(sre_nfa_verify_acc_init(sc->d_nacc, (uint32_t) n, NULL));
        }
        SRE_HIP_TRY(hipStreamSynchronize(NULL));
        SRE_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&sc->d_lo), n * sizeof(int64_t)));
        SRE_HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&sc->h_lo), n * sizeof(int64_t), 0));
        sc->cap_streams = n;
    }
    /* this call's layout: [ptrs n][lens n][seg_first n + 1] and [records n][status n] */
    sc->layout_n = n;
    sc->h_ptrs = reinterpret_cast<const void **>(sc->h_in);
    sc->h_lens = sc->h_in + n;
    sc->h_seg_first = sc->h_in + 2 * n;
    sc->d_ptrs = reinterpret_cast<const void **>(sc->d_in);
    sc->d_lens = sc->d_in + n;
    sc->d_seg_first = sc->d_in + 2 * n;
    sc->d_records = reinterpret_cast<sre_int_t *>(sc->d_out);
    sc->h_records = reinterpret_cast<sre_int_t *>(sc->h_out);
    sc->d_status = reinterpret_cast<sre_stream_status_t *>(sc->d_out + record_bytes(sc, n));
    sc->h_status = reinterpret_cast<sre_stream_status_t *>(sc->h_out + record_bytes(sc, n));
    sc->d_nstatus = reinterpret_cast<sre_nfa_status_t *>(sc->d_status);
    sc->h_nstatus = reinterpret_cast<sre_nfa_status_t *>(sc->h_status);
    if ((sc->engine == SRE_HIP_ENGINE_VM || (sc->engine == SRE_HIP_ENGINE_NFA && sc->mode != SRE_HIP_THOMPSON))
        && n * sc->ctx_stride > sc->ctx_cap)
    {
        if (sc->d_ctx) (void) hipFree(sc->d_ctx);
        sc->d_ctx = NULL;
        sc->ctx_cap = 0;
        SRE_HIP_TRY(hipMalloc(&sc->d_ctx, n * sc->ctx_stride));
        sc->ctx_cap = n * sc->ctx_stride;
    }
    return 0;
hip_failed:
    return -1;
}

/* segment geometry: one lane per segment, 256 lanes per workgroup.  The number
 * of workgroups is steered towards a whole multiple of what the chip holds at
 * once (LDS-limited: 160 KiB per CU), so the last round of workgroups is not
 * mostly empty; segments are a multiple of the 64-byte tile and at least 1 KiB
 * so that the speculative warm-up stays a few percent. */
static int
scan_geometry(sre_hip_scanner_t *sc, size_t nstreams)
{
    uint64_t total = 0;
    for (size_t i = 0; i < nstreams; i++) total += sc->h_lens[i];
    uint64_t seg = sc->seg_override;
    {
        const char *e = getenv("SRE_HIP_SEG_BYTES");        /* experiment knobs */
        if (seg == 0 && e && atoi(e) > 0 && atoi(e) % 256 == 0) seg = (uint64_t) atoi(e);
        e = getenv("SRE_HIP_SEG_CAP");
        sc->seg_cap_env = e && atoi(e) > 0 ? (uint64_t) atoi(e) : 0;
    }
    if (seg == 0) {
        /* as few rounds of resident workgroups as keep a segment <= ~40 KiB: longer
         * segments mean fewer summaries to verify and a smaller share of warm-up,
         * shorter ones keep every CU busy on small batches */
        if (sc->blocks_per_cu == 0) {
            sc->blocks_per_cu = sc->engine == SRE_HIP_ENGINE_NFA
                                    ? (sc->use_sa ? sre_nfa_sa_blocks_per_cu(&sc->satab)
                                                  : sre_nfa_blocks_per_cu(sc->mode == SRE_HIP_THOMPSON ? 0 : 1, sc->ntab.nslices,
                                                                          sc->ntab.nassert != 0))
                                                                 : sre_scan_blocks_per_cu(&sc->tab->h);
        }
        const uint64_t resident = (uint64_t) sre_hip_cu_count() * (uint64_t) sc->blocks_per_cu * SRE_SCAN_BLOCK;
        /* (measured, one box, 4 GiB: the COUNT kernel at two workgroups per CU takes 1.33 ms
         * with 16 640-byte segments = two rounds of resident workgroups, 1.25 ms with 33 280 =
         * one round, and 1.67 ms with 21 760 = one and a half: a whole number of rounds
         * matters, and one long round beats two short ones; profiles/r02_experiments.txt) */
        const uint64_t seg_cap = sc->seg_cap_env ? sc->seg_cap_env : 40960;
        uint64_t       rounds = (total + resident * seg_cap - 1) / (resident * seg_cap);
        if (rounds < 1) rounds = 1;
        /* ... and a few workgroup slots are left spare: a grid that needs EVERY slot of its
         * last round waits a whole extra round for the stragglers when anything else (the
         * tail kernels of the previous call) holds a slot at launch — 509 workgroups on 512
         * slots ran 30 % slower than 505 */
        const uint64_t lanes = resident * rounds - (resident * rounds >> 6);
        seg = (total / lanes + SRE_SCAN_SEG_ALIGN) / SRE_SCAN_SEG_ALIGN * SRE_SCAN_SEG_ALIGN;
        /* a small batch does not fill the chip whatever the segment size, and a lane's walk
         * is a serial chain (~1.5 us per 64-byte round): short segments, although half of
         * what such a lane reads is then warm-up (a 1 MiB chunk: 35 us at 1 KiB, 13 us at 256 B) */
        if (seg < 256) seg = 256;
        /* rows that are a multiple of 4 KiB apart land on the same HBM channels */
        if (seg % 4096 == 0) seg += SRE_SCAN_SEG_ALIGN;
    }
    uint64_t nsegs = 0;
    for (size_t i = 0; i < nstreams; i++) {
        sc->h_seg_first[i] = nsegs;
        uint64_t k = (sc->h_lens[i] + seg - 1) / seg;
        nsegs += k ? k : 1;             /* an empty stream still takes its EOF step */
    }
    sc->h_seg_first[nstreams] = nsegs;
    sc->geom.streams = reinterpret_cast<const uint8_t *const *>(sc->d_ptrs);
    sc->geom.lens = sc->d_lens;
    sc->geom.seg_first = sc->d_seg_first;
    sc->geom.nstreams = (uint32_t) nstreams;
    sc->geom.seg_bytes = (uint32_t) seg;
    sc->geom.nsegs = nsegs;
    /* one stream: described in the kernel arguments, nothing to upload (table-driven
     * scanner; the NFA tier's window kernel reads the arrays) */
    sc->geom.one_ptr = static_cast<const uint8_t *>(sc->h_ptrs[0]);
    sc->geom.one_len = sc->h_lens[0];
    sc->geom_one = (nstreams == 1 && sc->engine == SRE_HIP_ENGINE_SCAN) ? SRE_GEOM_ONE : 0u;
    sc->geom.flags = (sc->geom.flags & ~SRE_GEOM_ONE) | sc->geom_one;

    if (sc->engine == SRE_HIP_ENGINE_NFA) {
        if (nsegs > sc->nsum_cap) {
            if (sc->d_nsum) (void) hipFree(sc->d_nsum);
            if (sc->d_belief) (void) hipFree(sc->d_belief);
            if (sc->d_bvalid) (void) hipFree(sc->d_bvalid);
            sc->d_nsum = NULL;
            sc->d_belief = NULL;
            sc->d_bvalid = NULL;
            sc->nsum_cap = 0;
            SRE_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&sc->d_nsum), nsegs * sizeof(sre_nfa_summary_t)));
            SRE_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&sc->d_belief), nsegs * sizeof(uint64_t)));
            SRE_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&sc->d_bvalid), nsegs));
            sc->nsum_cap = nsegs;
        }
        return 0;
    }
    if (nsegs > sc->sum_cap) {
        if (sc->d_sum) (void) hipFree(sc->d_sum);
        sc->d_sum = NULL;
        sc->sum_cap = 0;
        SRE_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&sc->d_sum), nsegs * sizeof(sre_seg_summary_t)));
        sc->sum_cap = nsegs;
    }
    /* the digest for the one-workgroup chain check: small batches only */
    sc->geom.digest = NULL;
    if (nsegs <= SRE_VERIFY_ONE_SEGS && sc->mode != SRE_HIP_PIKE_COUNT) {
        if (sc->d_digest == NULL
            && hipMalloc(reinterpret_cast<void **>(&sc->d_digest), SRE_VERIFY_ONE_SEGS * sizeof(sre_seg_digest_t)) != hipSuccess)
        {
            sc->d_digest = NULL;
        }
        sc->geom.digest = sc->d_digest;
    }
    {
        size_t need = nstreams * ((size_t) seg + 16);
        if (need > sc->scratch_cap) {
            if (sc->d_scratch) (void) hipFree(sc->d_scratch);
            sc->d_scratch = NULL;
            sc->scratch_cap = 0;
            SRE_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&sc->d_scratch), need * sizeof(uint16_t)));
            sc->scratch_cap = need;
        }
    }
    return 0;
hip_failed:
    return -1;
}

/* chain check of the set pass and, for Pike, the exact VM over the window of every
 * stream that is verified and holds an event (d_lo: the streams of this fix-up round) */
static int
nfa_finish(sre_hip_scanner_t *sc, const int64_t *d_lo, hipStream_t stream)
{
    const uint32_t n = sc->geom.nstreams;
    /* (the kernels know two modes: find-all counting is a loop of first-match searches) */
    const int kmode = sc->mode == SRE_HIP_THOMPSON ? SRE_HIP_THOMPSON : SRE_HIP_PIKE_FIRST;
    SRE_HIP_TRY(sre_launch_nfa_verify(kmode, sc->geom, sc->d_nsum, sc->d_nacc, sc->d_nstatus,
                                      sc->d_belief, sc->d_bvalid, sc->d_records, sc->ovec_slots, d_lo, stream));
    if (sc->mode != SRE_HIP_THOMPSON) {
        if (sc->d_pwave) {
            SRE_HIP_TRY(sre_launch_pike_window_wave(sc->d_pwave, sc->h_pwave, sc->d_ptrs, sc->d_lens, n, sc->d_records,
                                                    sc->ovec_slots, reinterpret_cast<sre_nfa_window_t *>(sc->d_nstatus), d_lo,
                                                    sc->geom.sflags ? sc->d_creq : NULL, stream));
        } else {
            /* (the window kernel zero-fills the context it uses) */
            SRE_HIP_TRY(sre_launch_pike_window(sc->dp->d_blob, sc->dp->blob_bytes, sc->d_ptrs, sc->d_lens, n, sc->d_ctx, sc->ctx_stride,
                                               sc->d_records, sc->ovec_slots,
                                               reinterpret_cast<sre_nfa_window_t *>(sc->d_nstatus), d_lo,
                                               sc->geom.sflags ? sc->d_creq : NULL, stream));
        }
    }
    return 0;
hip_failed:
    return -1;
}

/* find-all rounds: the first horizon and the smallest one (SRE_HIP_COUNT_HORIZON: tests force tiny ones) */
static uint64_t
count_horizon(uint64_t dflt)
{
    const char *e = getenv("SRE_HIP_COUNT_HORIZON");
    return e && atoll(e) > 0 ? (uint64_t) atoll(e) : dflt;
}

/* NFA tier: one pass over the batch in h_ptrs / h_lens (geometry, set kernel, chain check, exact
 * windows), queued; with_copy: the records and status words travel to the host behind it */
static int
nfa_enqueue_pass(sre_hip_scanner_t *sc, size_t nstreams, hipStream_t stream, bool with_copy)
{
    if (scan_geometry(sc, nstreams) != 0) return -1;
    /* (a round of a find-all count is a sub-batch: the blocks keep the call's layout) */
    SRE_HIP_TRY(hipMemcpyAsync(sc->d_in, sc->h_in, (3 * sc->layout_n + 1) * sizeof(uint64_t),
                               hipMemcpyHostToDevice, stream));
    if (sc->ev0 == NULL) {
        SRE_HIP_TRY(hipEventCreate(&sc->ev0));
        SRE_HIP_TRY(hipEventCreate(&sc->ev1));
    }
    {
        /* set pass, chain check, and (Pike) the exact VM over each stream's window */
        const bool timed = !sc->ev_valid;       /* find-all: the first round's scan is the one reported */
        if (timed) SRE_HIP_TRY(hipEventRecord(sc->ev0, stream));
        SRE_HIP_TRY(nfa_launch_scan(sc, NULL, NULL, NULL, stream));
        if (timed) SRE_HIP_TRY(hipEventRecord(sc->ev1, stream));
        sc->ev_valid = 1;
    }
    if (sc->tail_stream_set && sc->tail_stream != stream) {
        SRE_HIP_TRY(hipStreamWaitEvent(sc->tail_stream, sc->ev1, 0));
        stream = sc->tail_stream;
    }
    if (nfa_finish(sc, NULL, stream) != 0) return -1;
    if (with_copy) return 0;        /* the caller queues the copy */
    return 0;
hip_failed:
    return -1;
}

/* NFA tier: segments behind a wrong entry set are re-run — the first one from the exact carried
 * set, the ones behind it from what their predecessor's lane ended in last round (sets only grow
 * towards the truth, so corrections travel many segments per round) — until every stream's
 * verified prefix reaches its event or its end.  h_nstatus holds the status of the pass before. */
static int
nfa_settle(sre_hip_scanner_t *sc, size_t n, hipStream_t stream, bool *psettled)
{
    for (bool first = true;; first = false) {
        if (!first) {
            SRE_HIP_TRY(hipMemcpyAsync(sc->h_nstatus, sc->d_nstatus, n * sizeof(sre_nfa_status_t),
                                       hipMemcpyDeviceToHost, stream));
            SRE_HIP_TRY(hipStreamSynchronize(stream));
        }
        size_t pending = 0;
        for (size_t i = 0; i < n; i++) {
            if (sc->h_nstatus[i].done) {
                sc->h_lo[i] = -1;
            } else {
                sc->h_lo[i] = sc->h_nstatus[i].first_bad;
                pending++;
            }
        }
        if (pending == 0) break;
        *psettled = false;
        if (++sc->fixup_rounds > 1000000) {
            fprintf(stderr, "[sregex-hip] NFA scanner fix-up did not converge\n");
            return -1;
        }
        SRE_HIP_TRY(hipMemcpyAsync(sc->d_lo, sc->h_lo, n * sizeof(int64_t), hipMemcpyHostToDevice, stream));
        if (sc->fixup_rounds > SRE_SPECULATIVE_FIXUPS && sc->geom.nsegs <= ((size_t) 1 << 23)       /* (512 bytes a segment) */
            && getenv("SRE_HIP_NO_NFA_EXACT") == NULL) {
            /* speculation does not settle this batch (a program that never forgets): every remaining lane's
             * exact entry set from the segments' singleton exit sets — the pass below is then exact */
            if (sc->geom.nsegs > sc->nmat_cap) {
                if (sc->d_nmat) (void) hipFree(sc->d_nmat);
                sc->d_nmat = NULL;
                sc->nmat_cap = 0;
                SRE_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&sc->d_nmat), sc->geom.nsegs * 64 * sizeof(uint64_t)));
                sc->nmat_cap = sc->geom.nsegs;
            }
            SRE_HIP_TRY(sre_launch_nfa_exact_entries(sc->use_sa ? 1 : 0, sc->ntab, sc->satab, sc->geom, sc->d_nsum, sc->d_lo,
                                                     sc->d_nmat, sc->d_belief, sc->d_bvalid, stream));
            sc->exact_passes++;
        }
        SRE_HIP_TRY(nfa_launch_scan(sc, sc->d_lo, sc->d_belief, sc->d_bvalid, stream));
        if (nfa_finish(sc, sc->d_lo, stream) != 0) return -1;
    }
    return 0;
hip_failed:
    return -1;
}

/*
 * Find-all counting on the NFA tier: the reference's caller iterates
 * exec(input + ovector[1], ...) on one re-armed context (sre_vm_pike.c:179-196, :586-636), every
 * search a first-match search that starts at the previous match's end.  Each ROUND here runs the
 * next piece of every unfinished stream's current search as one batch: the set kernel scans a
 * buffer that starts at a CLEAN position of the search (its start, or the last clean position the
 * previous round found: the list there is the fresh initial closure, so nothing of the stream in
 * front of it matters) and is at most `horizon` bytes long; a MATCH event in it sends the exact VM
 * over the stream from the search's start (it picks the search up at the clean position in front of
 * the event, as for a first match) and the next search starts at the match's end; no event moves
 * the buffer to its last clean position, or lets the horizon grow when it has none.  The horizon
 * follows the distance between matches, so sparse matches cost about one pass over the stream and
 * a round trip per match; dense matches run at the round-trip rate.
 */
static int
nfa_count_rounds(sre_hip_scanner_t *sc, sre_int_t *results)
{
    NfaCount    &c = *sc->cnt;
    const size_t n = c.st.size(), slots = 2 + (size_t) sc->ovec_slots;
    hipStream_t  stream = sc->last_stream;
    const bool   tail_set = sc->tail_stream_set;
    if (n > sc->cnt_cap) {
        if (sc->d_sflags) (void) hipFree(sc->d_sflags);
        if (sc->h_sflags) (void) hipHostFree(sc->h_sflags);
        if (sc->d_creq) (void) hipFree(sc->d_creq);
        if (sc->h_creq) (void) hipHostFree(sc->h_creq);
        sc->d_sflags = sc->h_sflags = NULL;
        sc->d_creq = sc->h_creq = NULL;
        sc->cnt_cap = 0;
        SRE_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&sc->d_sflags), n));
        SRE_HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&sc->h_sflags), n, 0));
        SRE_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&sc->d_creq), n * sizeof(sre_nfa_count_req_t)));
        SRE_HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&sc->h_creq), n * sizeof(sre_nfa_count_req_t), 0));
        sc->cnt_cap = n;
    }
    sc->tail_stream_set = false;            /* every round is read back: one stream */
    for (;;) {
        c.active.clear();
        for (size_t i = 0; i < n; i++) {
            NfaCountStream &t = c.st[i];
            /* a search on an empty remainder: size 0 with eof answers SRE_DECLINED (:179-196) */
            if (!t.done && t.cur >= t.n && t.count > 0) t.done = true;
            if (!t.done) c.active.push_back(i);
        }
        if (c.active.empty()) break;
        const size_t na = c.active.size();
        if (++sc->count_rounds > 100000000) return -1;
        for (size_t j = 0; j < na; j++) {
            NfaCountStream &t = c.st[c.active[j]];
            const uint64_t  left = t.n - t.q, len = left < t.horizon ? left : t.horizon;
            sc->h_ptrs[j] = t.base + t.q;
            sc->h_lens[j] = len;
            sc->h_sflags[j] = (uint8_t) (t.var_q | (t.var_cur << 2) | (t.mode_q << 4) | (len < left ? SRE_SFLAG_NO_EOF : 0u));
            sre_nfa_count_req_t &r = sc->h_creq[j];
            r.vptr = t.base + t.cur;
            r.vlen = t.n - t.cur;
            r.processed = (int64_t) t.cur;
            r.start_add = (int64_t) (t.q - t.cur);
            r.preset_flags = t.var_cur == 1 ? SRE_PRESET_SEEN_NEWLINE : 0u;
            r.pad = 0;
        }
        SRE_HIP_TRY(hipMemcpyAsync(sc->d_sflags, sc->h_sflags, na, hipMemcpyHostToDevice, stream));
        SRE_HIP_TRY(hipMemcpyAsync(sc->d_creq, sc->h_creq, na * sizeof(sre_nfa_count_req_t), hipMemcpyHostToDevice, stream));
        sc->geom.sflags = sc->d_sflags;
        sc->geom.flags = 0;
        if (nfa_enqueue_pass(sc, na, stream, false) != 0) goto hip_failed;
        SRE_HIP_TRY(hipMemcpyAsync(sc->h_out, sc->d_out, record_bytes(sc, sc->layout_n) + sc->layout_n * sizeof(sre_stream_status_t),
                                   hipMemcpyDeviceToHost, stream));
        SRE_HIP_TRY(hipStreamSynchronize(stream));
        {
            bool settled = true;
            if (nfa_settle(sc, na, stream, &settled) != 0) goto hip_failed;
            if (!settled) {
                SRE_HIP_TRY(hipMemcpyAsync(sc->h_out, sc->d_out, record_bytes(sc, sc->layout_n) + sc->layout_n * sizeof(sre_stream_status_t),
                                           hipMemcpyDeviceToHost, stream));
                SRE_HIP_TRY(hipStreamSynchronize(stream));
            }
        }
        for (size_t j = 0; j < na; j++) {
            NfaCountStream         &t = c.st[c.active[j]];
            const sre_nfa_status_t &w = sc->h_nstatus[j];
            const sre_int_t        *rec = sc->h_records + j * slots;
            const bool              truncated = (sc->h_sflags[j] & SRE_SFLAG_NO_EOF) != 0;
            if (getenv("SRE_HIP_DEBUG_COUNT")) {
                fprintf(stderr, "[sregex-hip] count round %d stream %zu: cur %llu q %llu var %u/%u mode %u len %llu%s -> ev %lld clean %lld "
                                "cmode %x rec %lld (%lld, %lld)\n", sc->count_rounds, c.active[j], (unsigned long long) t.cur,
                        (unsigned long long) t.q, t.var_cur, t.var_q, t.mode_q, (unsigned long long) sc->h_lens[j],
                        truncated ? " (truncated)" : "", (long long) w.ev_pos, (long long) w.clean_pos, w.clean_mode,
                        (long long) rec[0], (long long) rec[2], (long long) rec[3]);
            }
            if (w.ev_pos >= 0) {
                if (rec[0] < 0) {
                    /* (an event is a thread reaching MATCH: the exact VM finds that match or an earlier one) */
                    fprintf(stderr, "[sregex-hip] find-all on the NFA tier: the exact window found no match behind an event\n");
                    t.error = t.done = true;
                    continue;
                }
                t.count++;
                t.last.assign(rec, rec + slots);
                const uint64_t end = (uint64_t) rec[3];
                /* the next horizon: a little more than the distance from the previous match to this one */
                const uint64_t gap = end > t.cur ? end - t.cur : 1;
                t.horizon = gap + gap / 4 < count_horizon(1u << 20) ? count_horizon(1u << 20) : gap + gap / 4;
                t.cur = t.q = end;
                t.var_cur = t.var_q = (w.clean_mode & SRE_NFA_MATCH_AFTER_NL) ? 1u : 2u;
                t.mode_q = 0;
                if (w.clean_mode & SRE_NFA_WINDOW_POISONED) t.error = t.done = true;     /* :616-622: the next exec fails */
            } else if (!truncated) {
                t.done = true;                  /* SRE_DECLINED ends the iteration */
            } else if (w.clean_pos > 0) {
                t.q += (uint64_t) w.clean_pos;
                t.var_q = (w.clean_mode & SRE_NFA_CLEAN_AFTER_NL) ? 1u : 2u;
                t.mode_q = (uint32_t) (w.clean_mode & 1);
                t.horizon *= 2;
            } else {
                t.horizon *= 4;                 /* threads alive all along: the same buffer, longer */
            }
        }
    }
    sc->geom.sflags = NULL;
    sc->tail_stream_set = tail_set;
    for (size_t i = 0; i < n; i++) {
        const NfaCountStream &t = c.st[i];
        sre_int_t            *out = results + i * slots;
        if (t.count == 0) {
            out[0] = t.error ? SRE_ERROR : SRE_DECLINED;
            out[1] = 0;
            for (size_t k = 2; k < slots; k++) out[k] = -1;
        } else {
            for (size_t k = 0; k < slots; k++) out[k] = t.last[k];
            if (t.error) out[0] = SRE_ERROR;
            out[1] = (sre_int_t) t.count;
        }
    }
    return 0;
hip_failed:
    sc->geom.sflags = NULL;
    sc->tail_stream_set = tail_set;
    return -1;
}

extern "C" SRE_API int
sre_hip_scanner_last_count_rounds(sre_hip_scanner_t *sc)
{
    return sc->count_rounds;
}

extern "C" SRE_API int
sre_hip_scan_enqueue(sre_hip_scanner_t *sc, const void *const *d_streams, const size_t *lens,
    size_t nstreams, void *hip_stream)
{
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    sc->fixup_rounds = 0;
    sc->exact_passes = 0;
    sc->lineage_passes = 0;
    sc->ev_valid = 0;
    sc->geom.init_variant = sc->next_init_variant;
    sc->geom.flags = 0;
    sc->geom.entry_state = 0;
    sc->next_init_variant = 0;
    if (nstreams == 0) {
        sc->last_n = 0;
        return 0;
    }
    if (scanner_reserve(sc, nstreams) != 0) return -1;
    for (size_t i = 0; i < nstreams; i++) {
        sc->h_ptrs[i] = d_streams[i];
        sc->h_lens[i] = lens[i];
    }
    if (sc->engine == SRE_HIP_ENGINE_VM) {
        SRE_HIP_TRY(hipMemcpyAsync(sc->d_in, sc->h_in, 2 * nstreams * sizeof(uint64_t),
                                   hipMemcpyHostToDevice, stream));
        if (sc->d_pwave && nstreams <= 16384) {
            /* one wavefront per stream: up to ~8000 of them run at once, and a stream's step does not
             * get slower with the length of its thread list (beyond that many streams one LANE per
             * stream keeps more of them in flight) */
            SRE_HIP_TRY(sre_launch_pike_scan_wave(sc->d_pwave, sc->h_pwave, sc->mode, sc->d_ptrs, sc->d_lens,
                                                  (uint32_t) nstreams, sc->d_records, sc->ovec_slots, stream));
        } else {
            /* zero-filled state == fresh context */
            SRE_HIP_TRY(hipMemsetAsync(sc->d_ctx, 0, nstreams * sc->ctx_stride, stream));
            SRE_HIP_TRY(sre_launch_vm_scan(sc->dp->d_blob, sc->mode, sc->d_ptrs, sc->d_lens,
                                           (uint32_t) nstreams, sc->d_ctx, sc->ctx_stride,
                                           sc->d_records, sc->ovec_slots, sc->dp->has_wave, stream));
        }
    } else if (sc->engine == SRE_HIP_ENGINE_NFA && sc->cnt != NULL) {
        /* find-all counting: a loop of first-match searches, run by results() (nfa_count_rounds) */
        NfaCount &c = *sc->cnt;
        c.st.assign(nstreams, NfaCountStream());
        for (size_t i = 0; i < nstreams; i++) {
            NfaCountStream &t = c.st[i];
            t.base = static_cast<const uint8_t *>(d_streams[i]);
            t.n = lens[i];
            t.cur = t.q = 0;
            t.horizon = count_horizon(8u << 20);
            t.var_cur = t.var_q = sc->geom.init_variant;
            t.mode_q = 0;
            t.count = 0;
            t.done = t.error = false;
        }
        sc->count_rounds = 0;
        sc->last_n = nstreams;
        sc->last_stream = stream;
        return 0;
    } else if (sc->engine == SRE_HIP_ENGINE_NFA) {
        if (nfa_enqueue_pass(sc, nstreams, stream, true) != 0) return -1;
        if (sc->tail_stream_set && sc->tail_stream != stream) stream = sc->tail_stream;
    } else {
        if (scan_geometry(sc, nstreams) != 0) return -1;
        if (!sc->geom_one) {
            static const bool dma = getenv("SRE_HIP_DMA_UPLOAD") != NULL;      /* experiment knob: the old way */
            if (dma) {
                SRE_HIP_TRY(hipMemcpyAsync(sc->d_in, sc->h_in, (3 * nstreams + 1) * sizeof(uint64_t),
                                           hipMemcpyHostToDevice, stream));
            } else {
                SRE_HIP_TRY(sre_launch_upload_words(sc->h_in, sc->d_in, (uint32_t) (3 * nstreams + 1), stream));
            }
        }
        /* speculative pass, chain check, captures — all queued; results() only
         * has to look at the status words */
        if (sc->ev0 == NULL) {
            SRE_HIP_TRY(hipEventCreate(&sc->ev0));
            SRE_HIP_TRY(hipEventCreate(&sc->ev1));
        }
        SRE_HIP_TRY(hipEventRecord(sc->ev0, stream));
        SRE_HIP_TRY(sre_launch_scan(sc->tab->d_tab, sc->tab->h, sc->geom, sc->d_sum, NULL, NULL, stream));
        SRE_HIP_TRY(hipEventRecord(sc->ev1, stream));
        sc->ev_valid = 1;
        /* everything behind the scan kernel on the tail stream, if one is set: the caller's
         * next scan follows this one on `stream` with no gap while these small kernels run */
        if (sc->tail_stream_set && sc->tail_stream != stream) {
            SRE_HIP_TRY(hipStreamWaitEvent(sc->tail_stream, sc->ev1, 0));
            stream = sc->tail_stream;
        }
        {
            /* one small buffer: chain check and captures in one workgroup */
            const int fused = sc->geom_one && sc->geom.nsegs <= SRE_VERIFY_ONE_SEGS && sc->mode != SRE_HIP_PIKE_COUNT;
            if (!fused) SRE_HIP_TRY(sre_launch_verify(sc->tab->h, sc->geom, sc->d_sum, sc->d_acc, sc->d_status, stream));
            SRE_HIP_TRY(sre_launch_captures(sc->tab->d_tab, sc->tab->h, sc->geom, sc->d_sum, sc->d_status,
                                            sc->d_scratch, sc->d_records, sc->ovec_slots,
                                            NULL, NULL, 0, fused, stream));
        }
    }
    /* records (and the scanner's status words behind them) in one copy */
    SRE_HIP_TRY(hipMemcpyAsync(sc->h_out, sc->d_out,
                               record_bytes(sc, nstreams)
                                   + (sc->engine != SRE_HIP_ENGINE_VM ? nstreams * sizeof(sre_stream_status_t) : 0),
                               hipMemcpyDeviceToHost, stream));
    if (sc->ev_done == NULL) SRE_HIP_TRY(hipEventCreateWithFlags(&sc->ev_done, hipEventDisableTiming));
    SRE_HIP_TRY(hipEventRecord(sc->ev_done, stream));
    sc->last_n = nstreams;
    sc->last_stream = stream;
    return 0;
hip_failed:
    return -1;
}

/* Segments behind a broken state chain are re-run from the exact carried state until
 * every stream's verified prefix reaches its end (h_status holds the latest status on
 * return).  with_captures: the capture kernel runs behind every round. */
static int
scan_settle(sre_hip_scanner_t *sc, size_t n, hipStream_t stream, bool with_captures, bool *psettled)
{
    bool settled = true;
    int  batch = sc->mode == SRE_HIP_PIKE_COUNT ? 2 : SRE_SPECULATIVE_FIXUPS + 1;
    /* h_status holds the status of the pass before.  Rounds are queued in batches and the
     * status is read once per batch: a round finds its streams and their first wrong segment
     * in the status words on the device, and is a no-op for a stream that has settled.
     * FIRST / Thompson: two speculative rounds, then the exact entry states (all three
     * queued at once); COUNT: 2, 4, 8, ... speculative rounds. */
    for (;;) {
        size_t pending = 0;
        for (size_t i = 0; i < n; i++) pending += sc->h_status[i].done ? 0 : 1;
        if (pending == 0) break;
        settled = false;
        for (int r = 0; r < batch; r++) {
            if (++sc->fixup_rounds > 1000000) {
                fprintf(stderr, "[sregex-hip] scanner fix-up did not converge\n");
                return -1;
            }
            const uint8_t *d_entry = NULL;
            static const bool count_exact = getenv("SRE_HIP_NO_COUNT_EXACT") == NULL;      /* (experiment knob) */
            if ((sc->mode != SRE_HIP_PIKE_COUNT || count_exact) && sc->fixup_rounds > SRE_SPECULATIVE_FIXUPS) {
                /* speculation does not settle this stream (an automaton that never
                 * forgets): compose the segments' transition functions instead — after
                 * this pass every lane enters with the exact state */
                if (sc->geom.nsegs > sc->fn_cap) {
                    if (sc->d_fn) (void) hipFree(sc->d_fn);
                    sc->d_fn = NULL;
                    sc->fn_cap = 0;
                    const size_t nchunks = sc->geom.nsegs / 256 + 1;
                    SRE_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&sc->d_fn),
                                          sc->geom.nsegs * 64 + nchunks * 64 + nchunks + sc->geom.nsegs + 64));
                    sc->fn_cap = sc->geom.nsegs;
                }
                const size_t nchunks = sc->geom.nsegs / 256 + 1;
                uint8_t *d_comp = sc->d_fn + sc->geom.nsegs * 64, *d_chunk = d_comp + nchunks * 64;
                uint8_t *d_ent = d_chunk + nchunks;
                SRE_HIP_TRY(sre_launch_exact_entries(sc->tab->d_tab, sc->tab->h, sc->geom, sc->d_sum, sc->d_status,
                                                     sc->d_fn, d_comp, d_chunk, d_ent, stream));
                d_entry = d_ent;
                sc->exact_passes++;
            }
            SRE_HIP_TRY(sre_launch_scan(sc->tab->d_tab, sc->tab->h, sc->geom, sc->d_sum, sc->d_status, d_entry, stream));
            SRE_HIP_TRY(sre_launch_verify(sc->tab->h, sc->geom, sc->d_sum, sc->d_acc, sc->d_status, stream));
            if (with_captures) {
                SRE_HIP_TRY(sre_launch_captures(sc->tab->d_tab, sc->tab->h, sc->geom, sc->d_sum,
                                                sc->d_status, sc->d_scratch, sc->d_records,
                                                sc->ovec_slots, NULL, NULL, 0, 0, stream));
            }
        }
        SRE_HIP_TRY(hipMemcpyAsync(sc->h_status, sc->d_status, n * sizeof(sre_stream_status_t),
                                   hipMemcpyDeviceToHost, stream));
        SRE_HIP_TRY(hipStreamSynchronize(stream));
        batch = sc->mode == SRE_HIP_PIKE_COUNT ? (batch < 16 ? 2 * batch : 16) : 1;
    }
    if (psettled) *psettled = settled;
    return 0;
hip_failed:
    return -1;
}

extern "C" SRE_API int
sre_hip_scan_results(sre_hip_scanner_t *sc, sre_int_t *results)
{
    if (sc->last_n == 0) return 0;
    const size_t n = sc->last_n;
    hipStream_t  stream = sc->last_stream;

    const size_t bytes = n * (2 + (size_t) sc->ovec_slots) * sizeof(int64_t);
    bool         settled = true;        /* the copies queued by enqueue() are the answer */

    if (sc->engine == SRE_HIP_ENGINE_NFA && sc->cnt != NULL) return nfa_count_rounds(sc, results);
    /* everything enqueue() queued for this call, result copies included */
    SRE_HIP_TRY(hipEventSynchronize(sc->ev_done));
    if (sc->engine == SRE_HIP_ENGINE_NFA && nfa_settle(sc, n, stream, &settled) != 0) return -1;
    if (sc->engine == SRE_HIP_ENGINE_SCAN) {
        if (scan_settle(sc, n, stream, true, &settled) != 0) return -1;
        /* a match whose lineage outran the plain backward walk: build the
         * per-segment ancestor maps in parallel and walk again, jumping */
        if (sc->mode != SRE_HIP_THOMPSON) {
            size_t want = 0;
            for (size_t i = 0; i < n; i++) want += sc->h_status[i].need_maps != 0;
            if (want) {
                if (sc->geom.nsegs > sc->maps_cap) {
                    if (sc->d_maps) (void) hipFree(sc->d_maps);
                    if (sc->d_blocks) (void) hipFree(sc->d_blocks);
                    sc->d_maps = sc->d_blocks = NULL;
                    sc->maps_cap = 0;
                    SRE_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&sc->d_maps),
                                          sc->geom.nsegs * sizeof(sre_seg_lineage_t)));
                    SRE_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&sc->d_blocks),
                                          (sc->geom.nsegs / SRE_LINEAGE_BLOCK + 1) * sizeof(sre_seg_lineage_t)));
                    sc->maps_cap = sc->geom.nsegs;
                }
                sc->lineage_passes++;
                settled = false;
                SRE_HIP_TRY(sre_launch_lineage(sc->tab->d_tab, sc->tab->h, sc->geom, sc->d_sum,
                                               sc->d_status, sc->d_maps, sc->d_blocks, stream));
                SRE_HIP_TRY(sre_launch_captures(sc->tab->d_tab, sc->tab->h, sc->geom, sc->d_sum,
                                                sc->d_status, sc->d_scratch, sc->d_records,
                                                sc->ovec_slots, sc->d_maps, sc->d_blocks,
                                                1, 0, stream));
            }
        }
    }
    if (sc->engine == SRE_HIP_ENGINE_SCAN && getenv("SRE_HIP_DEBUG_STATUS")) {
        /* diagnostics: what the chain check decided per stream */
        for (size_t i = 0; i < n && i < 8; i++) {
            const sre_stream_status_t &t = sc->h_status[i];
            fprintf(stderr, "[sregex-hip] stream %zu: rc %lld count %lld ev_pos %lld ev_sp %lld ev_state %u ev_sym %u "
                            "ev_apos %lld ev_astate %u ev_seg %lld limit %lld need_maps %d seg %u\n",
                    i, (long long) t.rc, (long long) t.count, (long long) t.ev_pos, (long long) t.ev_sp,
                    t.ev_state, t.ev_sym, (long long) t.ev_apos, t.ev_astate, (long long) t.ev_seg,
                    (long long) t.limit, t.need_maps, sc->geom.seg_bytes);
        }
        /* ... and what the lanes of the first segments recorded */
        const size_t ns = sc->geom.nsegs < 16 ? (size_t) sc->geom.nsegs : 16;
        std::vector<sre_seg_summary_t> hs(ns);
        if (hipMemcpy(hs.data(), sc->d_sum, ns * sizeof(sre_seg_summary_t), hipMemcpyDeviceToHost) == hipSuccess) {
            for (size_t k = 0; k < ns; k++) {
                fprintf(stderr, "[sregex-hip]   seg %zu: s_in %x s_out %x flags %x count %lld cur_sp %lld term %lld "
                                "pe_pos %lld lm_pos %lld lm_sp %lld\n",
                        k, hs[k].s_in, hs[k].s_out, hs[k].flags, (long long) hs[k].count,
                        (long long) hs[k].cur_sp, (long long) hs[k].term_pos, (long long) hs[k].pe_pos,
                        (long long) hs[k].lm_pos, (long long) hs[k].lm_sp);
            }
        }
    }
    if (!settled) {
        SRE_HIP_TRY(hipMemcpyAsync(sc->h_records, sc->d_records, bytes, hipMemcpyDeviceToHost, stream));
        if (sc->engine == SRE_HIP_ENGINE_NFA) {
            SRE_HIP_TRY(hipMemcpyAsync(sc->h_nstatus, sc->d_nstatus, n * sizeof(sre_nfa_status_t),
                                       hipMemcpyDeviceToHost, stream));
        }
        SRE_HIP_TRY(hipStreamSynchronize(stream));
    }
    memcpy(results, sc->h_records, bytes);
    return 0;
hip_failed:
    return -1;
}

extern "C" SRE_API int
sre_hip_scan_batch(sre_hip_scanner_t *sc, const void *const *d_streams, const size_t *lens,
    size_t nstreams, sre_int_t *results, void *hip_stream)
{
    if (sre_hip_scan_enqueue(sc, d_streams, lens, nstreams, hip_stream) != 0) return -1;
    return sre_hip_scan_results(sc, results);
}

/* One device-resident buffer through the scanner, for sre_vm_*_exec on large
 * whole-buffer calls.  `init_variant` is the SRE_DFA_INIT_* of a search on a
 * re-armed context; *poisoned reports the "threads still listed at eof" state
 * (sre_vm_pike.c:616-622). */
extern "C" int
sre_hip_scan_one(sre_hip_scanner_t *sc, const void *d_buf, size_t len, int init_variant,
    sre_int_t *rec, int *poisoned, hipStream_t stream)
{
    const void *ptrs[1] = {d_buf};
    size_t      lens[1] = {len};
    sc->next_init_variant = (uint32_t) init_variant;
    if (sre_hip_scan_enqueue(sc, ptrs, lens, 1, stream) != 0) return -1;
    if (sre_hip_scan_results(sc, rec) != 0) return -1;
    if (poisoned) {
        *poisoned = sc->engine == SRE_HIP_ENGINE_SCAN ? sc->h_status[0].error
                  : sc->engine == SRE_HIP_ENGINE_NFA ? (sc->h_nstatus[0].clean_mode & SRE_NFA_WINDOW_POISONED) != 0 : 0;
    }
    return 0;
}

/*
 * One CHUNK of one stream through the table-driven scanner, for the chunked use of
 * sre_vm_pike_exec (sre_vm_api.cpp): scan + chain check (+ fix-up rounds), then
 * sre_k_stream_tail.  `continues`: the search is under way, `entry_state` is the
 * automaton state the previous chunk ended in and *d_ctx holds its threads' capture
 * vectors; otherwise a search starts at the chunk's first byte with `init_variant`.
 * `base`: absolute stream offset of the chunk.  The result lands in *h_res (pinned,
 * device-visible as d_res).  Synchronous.
 */
extern "C" int
sre_hip_scan_stream_chunk(sre_hip_scanner_t *sc, const void *d_buf, size_t len, int init_variant,
    int continues, uint32_t entry_state, int eof, int64_t base, sre_stream_ctx_t *d_ctx,
    sre_stream_result_t *d_res, const sre_stream_result_t *h_res, uint32_t ovec_slots, hipStream_t stream,
    void (*midway)(void *), void *midway_arg)
{
    /* (midway: what the caller still has to do for the chunk to arrive — it runs after the
     * launches, and on every way out) */
    struct Midway {
        void (*fn)(void *);
        void  *arg;
        void   run() { if (fn) fn(arg); fn = nullptr; }
        ~Midway() { run(); }
    } mid{midway, midway_arg};
    if (sc->engine != SRE_HIP_ENGINE_SCAN || sc->mode == SRE_HIP_PIKE_COUNT) return -1;
    sc->fixup_rounds = 0;
    sc->exact_passes = 0;
    sc->lineage_passes = 0;
    sc->ev_valid = 0;
    if (scanner_reserve(sc, 1) != 0) return -1;
    sc->h_ptrs[0] = d_buf;
    sc->h_lens[0] = len;
    sc->geom.init_variant = (uint32_t) init_variant;
    if (scan_geometry(sc, 1) != 0) return -1;
    sc->geom.flags = (continues ? SRE_GEOM_CONTINUES : 0u) | (eof ? 0u : SRE_GEOM_NO_EOF) | sc->geom_one;
    sc->geom.entry_state = entry_state;
    /* two launches per chunk: the scan, and the chain check + tail in one workgroup; should
     * the speculative entry states of the chunk's lanes have been wrong (rare), the tail
     * says so and the rounds are run first */
    SRE_HIP_TRY(sre_launch_scan(sc->tab->d_tab, sc->tab->h, sc->geom, sc->d_sum, NULL, NULL, stream));
    {
        /* (a large chunk has too many segments for one workgroup: the chain check's own kernels) */
        const int fused = sc->geom.nsegs <= SRE_VERIFY_ONE_SEGS;
        if (!fused) SRE_HIP_TRY(sre_launch_verify(sc->tab->h, sc->geom, sc->d_sum, sc->d_acc, sc->d_status, stream));
        SRE_HIP_TRY(sre_launch_stream_tail(sc->tab->d_tab, sc->tab->h, sc->geom, sc->d_sum, sc->d_status,
                                           sc->d_scratch, d_ctx, d_res, base, eof, ovec_slots, fused, stream));
    }
    mid.run();
    /* the result lands in host-visible memory, rc last: watching that word costs less than
     * the runtime's wait (an interrupt and a wake-up) — for as long as a chunk of this size
     * can reasonably take, then the ordinary wait */
    {
        const volatile int64_t *prc = &h_res->rc;
        const auto              t0 = std::chrono::steady_clock::now();
        const auto              limit = std::chrono::microseconds(200 + (int64_t) (len >> 12));
        while (*prc == SRE_STREAM_PENDING) {
            if (std::chrono::steady_clock::now() - t0 > limit) {
                SRE_HIP_TRY(hipStreamSynchronize(stream));
                break;
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
    }
    if (h_res->rc == SRE_STREAM_UNSETTLED) {
        SRE_HIP_TRY(hipMemcpyAsync(sc->h_status, sc->d_status, sizeof(sre_stream_status_t),
                                   hipMemcpyDeviceToHost, stream));
        SRE_HIP_TRY(hipStreamSynchronize(stream));
        if (scan_settle(sc, 1, stream, false, NULL) != 0) goto hip_failed;
        SRE_HIP_TRY(sre_launch_stream_tail(sc->tab->d_tab, sc->tab->h, sc->geom, sc->d_sum, sc->d_status,
                                           sc->d_scratch, d_ctx, d_res, base, eof, ovec_slots, 0, stream));
        SRE_HIP_TRY(hipStreamSynchronize(stream));
    }
    sc->geom.flags = 0;
    sc->last_n = 0;
    return 0;
hip_failed:
    sc->geom.flags = 0;
    return -1;
}

/* can chunks of a stream go through this scanner?  (the carried state has room for 64
 * threads of 64 capture slots) */
extern "C" int
sre_hip_scanner_streams(sre_hip_scanner_t *sc)
{
    if (sc->engine != SRE_HIP_ENGINE_SCAN) return 0;
    if (sc->mode == SRE_HIP_THOMPSON) return 1;         /* the state alone */
    return sc->mode == SRE_HIP_PIKE_FIRST && sc->tab->h.max_threads <= SRE_STREAM_MAX_THREADS
           && sc->tab->h.nslots <= SRE_STREAM_MAX_SLOTS;
}

/* ------------------------------------------------------------------ helpers */

extern "C" SRE_API void *
sre_hip_alloc(size_t bytes)
{
    void *p = NULL;
    if (sre_hip_ready() != 0) return NULL;
    hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
    if (e != hipSuccess) {
        sre_hip_fail("hipMalloc", e);
        return NULL;
    }
    return p;
}

extern "C" SRE_API void
sre_hip_free(void *d_ptr)
{
    if (d_ptr) (void) hipFree(d_ptr);
}

extern "C" SRE_API int
sre_hip_upload(void *d_dst, const void *h_src, size_t bytes)
{
    hipError_t e = hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice);
    return e == hipSuccess ? 0 : sre_hip_fail("hipMemcpy H2D", e);
}

extern "C" SRE_API int
sre_hip_download(void *h_dst, const void *d_src, size_t bytes)
{
    hipError_t e = hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost);
    return e == hipSuccess ? 0 : sre_hip_fail("hipMemcpy D2H", e);
}

extern "C" SRE_API int
sre_hip_synchronize(void *hip_stream)
{
    hipError_t e = hipStreamSynchronize(static_cast<hipStream_t>(hip_stream));
    return e == hipSuccess ? 0 : sre_hip_fail("hipStreamSynchronize", e);
}

extern "C" SRE_API int
sre_hip_gen_data(void *d_dst, size_t n, const void *h_tail, size_t tail_len, void *hip_stream)
{
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    void       *d_tail = NULL;
    int         rc = -1;
    if (sre_hip_ready() != 0 || tail_len > n) return -1;
    SRE_HIP_TRY(hipMalloc(&d_tail, tail_len ? tail_len : 1));
    if (tail_len) {
        SRE_HIP_TRY(hipMemcpyAsync(d_tail, h_tail, tail_len, hipMemcpyHostToDevice, stream));
    }
    SRE_HIP_TRY(sre_launch_gen_data(d_dst, n, tail_len, d_tail, stream));
    SRE_HIP_TRY(hipStreamSynchronize(stream));
    rc = 0;
hip_failed:
    if (d_tail) (void) hipFree(d_tail);
    return rc;
}

static uint32_t *
ceiling_sink(void)
{
    static uint32_t *d_sink = NULL;
    if (d_sink == NULL
        && hipMalloc(reinterpret_cast<void **>(&d_sink), SRE_CEILING_GRID * sizeof(uint32_t)) != hipSuccess)
    {
        d_sink = NULL;
    }
    return d_sink;
}

extern "C" SRE_API int
sre_hip_read_pattern(const void *d_src, size_t n, unsigned seg_bytes, unsigned tile,
                     unsigned lds_bytes, void *hip_stream)
{
    if (sre_hip_ready() != 0) return -1;
    uint32_t *d_sink = ceiling_sink();
    if (d_sink == NULL) return -1;
    SRE_HIP_TRY(sre_launch_read_pattern(d_src, n, seg_bytes, tile, lds_bytes, d_sink,
                                        static_cast<hipStream_t>(hip_stream)));
    return 0;
hip_failed:
    return -1;
}

extern "C" SRE_API int
sre_hip_read_ceiling(const void *d_src, size_t n, void *hip_stream)
{
    if (sre_hip_ready() != 0) return -1;
    uint32_t *d_sink = ceiling_sink();
    if (d_sink == NULL) return -1;
    SRE_HIP_TRY(sre_launch_read_ceiling(d_src, n, d_sink, static_cast<hipStream_t>(hip_stream)));
    return 0;
hip_failed:
    return -1;
}
