/*
 * sre_oracle.h — TEST INFRASTRUCTURE.  Not part of the product.
 *
 * CPU restatement (plain C, single thread) of the reference's two executors,
 *   sre_vm_pike_exec      reference src/sregex/sre_vm_pike.c:148-689
 *   sre_vm_thompson_exec  reference src/sregex/sre_vm_thompson.c:63-270
 * over this repo's position-independent sre_program_t.  It exists so that the
 * HIP path can be checked on a GPU box where /root/reference is absent.
 *
 * PINNED: tests/test_oracle.py checks it (a) against the reference's own CLI
 * output for all 1999 blocks of t/ files in whole-buffer AND byte-at-a-time
 * ("splitted") modes, in single and forced-multi-regex variants
 * (tests/golden/t_blocks.jsonl.gz), (b) against reference runs on gen-data
 * streams and iterated find-all traces (tests/golden/{gen_data,findall}.jsonl),
 * and, in the build container, (c) live against oracle/_ref/libsregex_ref.so.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * this library.  The product (sregex_amd/) never links or loads it.
 */
#ifndef SRE_ORACLE_H
#define SRE_ORACLE_H

#include <sregex/sregex.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sre_oracle_pike_ctx_s      sre_oracle_pike_ctx_t;
typedef struct sre_oracle_thompson_ctx_s  sre_oracle_thompson_ctx_t;

/* same argument meaning as sre_vm_pike_create_ctx / sre_vm_pike_exec */
SRE_API sre_oracle_pike_ctx_t *sre_oracle_pike_create_ctx(sre_program_t *prog,
    sre_int_t *ovector, size_t ovecsize);
SRE_API sre_int_t sre_oracle_pike_exec(sre_oracle_pike_ctx_t *ctx,
    const sre_char *input, size_t size, unsigned eof, sre_int_t **pending_matched);
SRE_API void sre_oracle_pike_free(sre_oracle_pike_ctx_t *ctx);

/* same argument meaning as sre_vm_thompson_create_ctx / sre_vm_thompson_exec */
SRE_API sre_oracle_thompson_ctx_t *sre_oracle_thompson_create_ctx(sre_program_t *prog);
SRE_API sre_int_t sre_oracle_thompson_exec(sre_oracle_thompson_ctx_t *ctx,
    const sre_char *input, size_t size, unsigned eof);
SRE_API void sre_oracle_thompson_free(sre_oracle_thompson_ctx_t *ctx);

/* iterate exec over one buffer on one ctx, re-feeding from each match end
 * (SURVEY.md 8b "stream contract"); returns the number of matches and, if
 * `spans` is non-NULL, writes up to max_spans records of (regex_id, ovector[nov]);
 * *final_rc receives the rc that ended the iteration. */
SRE_API sre_int_t sre_oracle_pike_count(sre_program_t *prog, const sre_char *input,
    size_t size, sre_int_t *spans, size_t nov, size_t max_spans, sre_int_t *final_rc);

#ifdef __cplusplus
}
#endif
#endif
