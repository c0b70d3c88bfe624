/*
 * sre_hip_common.h — layouts shared by the host HIP layer and the kernels.
 *
 * Device program image ("blob"): the sre_program_t flattened for the GPU.
 *   - instructions are 16-byte records addressed by index (no pointers, no
 *     mutable tags: the reference's per-instruction generation tag,
 *     sre_vm_bytecode.h:51, becomes per-stream state),
 *   - every IN/NOTIN range list (sre_vm_pike.c:336-346 walks it per byte) is
 *     pre-expanded into a 256-bit membership bitmap, NOTIN already negated,
 *     identical bitmaps shared.
 */
#ifndef SRE_HIP_COMMON_H
#define SRE_HIP_COMMON_H

#include <stdint.h>
#include <stddef.h>

#define SRE_DEV_ALIGN(n)  (((n) + 15) & ~(size_t) 15)

#ifdef __HIPCC__
#   define SRE_HD __host__ __device__
#else
#   define SRE_HD
#endif

typedef struct {
    uint8_t   opcode;      /* SRE_OP_* */
    uint8_t   ch;          /* CHAR byte / ASSERT bit */
    uint16_t  cls;         /* IN/NOTIN: class bitmap index */
    uint32_t  x;           /* SPLIT/JMP */
    uint32_t  y;           /* SPLIT */
    uint32_t  arg;         /* SAVE slot / MATCH regex id */
} sre_dev_insn_t;

typedef struct {
    uint32_t  len;         /* instructions */
    uint32_t  nslots;      /* capture slots of the internal vector */
    uint32_t  nregexes;
    uint32_t  nthreads;    /* list-able instructions */
    uint32_t  nclasses;
    uint32_t  nleading;    /* leading instructions (0: no leading-byte skip) */
    int32_t   leading_byte;/* single leading CHAR, or -1 */
    uint32_t  wave_off;    /* byte offset of the sre_dev_wave_t behind the arrays below, 0: the program has none */
    /* followed, 16-B aligned, by:
     *   sre_dev_insn_t insns[len]
     *   uint32_t       classes[nclasses][8]
     *   uint32_t       multi_ncaps[nregexes]
     *   uint32_t       leading[nleading]      instruction indices
     */
} sre_dev_prog_hdr_t;

SRE_HD static inline size_t sre_dev_prog_insns_off(void) { return SRE_DEV_ALIGN(sizeof(sre_dev_prog_hdr_t)); }
SRE_HD static inline size_t sre_dev_prog_classes_off(uint32_t len) {
    return sre_dev_prog_insns_off() + SRE_DEV_ALIGN((size_t) len * sizeof(sre_dev_insn_t));
}
SRE_HD static inline size_t sre_dev_prog_ncaps_off(uint32_t len, uint32_t nclasses) {
    return sre_dev_prog_classes_off(len) + SRE_DEV_ALIGN((size_t) nclasses * 32);
}
SRE_HD static inline size_t sre_dev_prog_leading_off(uint32_t len, uint32_t nclasses, uint32_t nregexes) {
    return sre_dev_prog_ncaps_off(len, nclasses) + SRE_DEV_ALIGN((size_t) nregexes * 4);
}
SRE_HD static inline size_t sre_dev_prog_bytes(uint32_t len, uint32_t nclasses, uint32_t nregexes,
                                               uint32_t nleading) {
    return sre_dev_prog_leading_off(len, nclasses, nregexes) + SRE_DEV_ALIGN((size_t) (nleading + 1) * 4);
}

/*
 * The WAVE form of a program (Thompson semantics, programs without look-ahead assertions whose
 * list-able threads fit 64 bits): one wavefront walks one stream, lane q IS thread q of the
 * bit-parallel form (sre_nfa.h), the live set is a 64-bit lane mask in scalar registers, and one
 * input byte is  T = S & accept[byte];  S' = ballot((pred[lane] & T) != 0)  — BASELINE.json's
 * north_star layout (reference loop: sre_vm_thompson.c:88-258; the x86 JIT keeps the same mask,
 * sre_vm_thompson_x64.dasc:81-130).  pred[q] = the threads whose closure lists q.
 */
typedef struct {
    uint64_t init0;         /* the list of the first buffer: \A and ^ hold */
    uint64_t match;         /* MATCH threads */
    uint64_t accept[256];
    uint64_t pred[64];
} sre_dev_wave_t;

/* ---- per-stream request / result of one exec() on the exact VM kernels ---- */

typedef struct {
    const uint8_t *input;       /* device pointer, or NULL when size <= 8 and   */
    uint64_t       inline_bytes;/* the chunk travels in the kernel argument    */
    uint64_t       size;
    uint32_t       eof;
    uint32_t       want_pending;/* caller passed a pending_matched pointer     */
    void          *ctx;         /* device stream state, see sre_hip_vm.hip     */
    void          *result;      /* sre_dev_result_t + ovector, host-visible    */
    uint64_t       ovec_slots;  /* caller ovector length in slots              */
    /* state of a context whose previous searches ran on the scanner: applied
     * when the VM kernel first touches the (still zero-filled) device context */
    int64_t        preset_processed;
    uint32_t       preset_valid;
    uint32_t       preset_flags;    /* SRE_PRESET_* */
    /* the context has not been used since it was opened / handed back: whatever its memory holds
     * is ignored (a flag instead of a fill in front of every first call) */
    uint32_t       fresh;
    /* `input` lies in pinned host memory (chunks up to SRE_SMALL_INPUT bytes: written next to the
     * request instead of a copy of their own): the kernel stages it into LDS before it runs */
    uint32_t       input_pinned;
} sre_dev_req_t;

#define SRE_SMALL_INPUT 1024u

#define SRE_PRESET_EMPTY_CAPTURE 1u
#define SRE_PRESET_SEEN_NEWLINE  2u
#define SRE_PRESET_SEEN_WORD     4u
#define SRE_PRESET_EOF           8u

typedef struct {
    int64_t   rc;               /* regex id >= 0, SRE_AGAIN, SRE_DECLINED, SRE_ERROR */
    int64_t   has_pending;
    int64_t   pending[2];
    int64_t   consumed;         /* bytes of this chunk the VM looked at */
    int64_t   pad[3];
    /* int64_t ovector[ovec_slots] follows */
} sre_dev_result_t;

#endif
