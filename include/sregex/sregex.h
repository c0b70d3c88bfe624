/*
 * sregex.h — public C API of the MI355X-native streaming regex matcher.
 *
 * Source-compatible with the reference's installed header
 * (reference src/sregex/sregex.h:46-171): same type names, status codes, flag
 * values and the same 15 entry points, so that the reference's own clients
 * (src/sre_cli.c, bench/sregex.c) compile and link against this library
 * unchanged.  The byte loops behind sre_vm_pike_exec / sre_vm_thompson_exec run
 * as HIP kernels on gfx950; there is no CPU matcher in this library.
 */
#ifndef SREGEX_AMD_SREGEX_H
#define SREGEX_AMD_SREGEX_H

#include <stdint.h>
#include <stdlib.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__) && __GNUC__ >= 4
#   define SRE_API    __attribute__ ((visibility ("default")))
#   define SRE_NOAPI  __attribute__ ((visibility ("hidden")))
#else
#   define SRE_API
#   define SRE_NOAPI
#endif

/* reference sregex.h:46-61 */
#ifndef sre_char
#define sre_char  sre_char
typedef uint8_t  sre_char;
#endif

#ifndef sre_int_t
#define sre_int_t sre_int_t
typedef intptr_t  sre_int_t;
#endif

#ifndef sre_uint_t
#define sre_uint_t sre_uint_t
typedef uintptr_t  sre_uint_t;
#endif

/* status codes, reference sregex.h:65-72 */
enum {
    SRE_OK       = 0,
    SRE_ERROR    = -1,
    SRE_AGAIN    = -2,
    SRE_BUSY     = -3,
    SRE_DONE     = -4,
    SRE_DECLINED = -5
};

/* ---- memory pool (reference sregex.h:78-84) ---- */
typedef struct sre_pool_s  sre_pool_t;

SRE_API sre_pool_t *sre_create_pool(size_t size);
SRE_API void sre_reset_pool(sre_pool_t *pool);
SRE_API void sre_destroy_pool(sre_pool_t *pool);

/* ---- regex parser (reference sregex.h:87-108) ---- */
enum {
    SRE_REGEX_CASELESS = 1,
    SRE_REGEX_NEWLINE  = 2
};

typedef struct sre_regex_s  sre_regex_t;

SRE_API sre_regex_t *sre_regex_parse(sre_pool_t *pool, sre_char *src,
    sre_uint_t *ncaps, int flags, sre_int_t *err_offset);

SRE_API void sre_regex_dump(sre_regex_t *re);

SRE_API sre_regex_t *sre_regex_parse_multi(sre_pool_t *pool,
    sre_char **regexes, sre_int_t nregexes, sre_uint_t *max_ncaps,
    int *multi_flags, sre_int_t *err_offset, sre_int_t *err_regex_id);

/* ---- regex compiler (reference sregex.h:111-120) ---- */
typedef struct sre_program_s  sre_program_t;

SRE_API void sre_program_dump(sre_program_t *prog);
SRE_API sre_program_t *sre_regex_compile(sre_pool_t *pool, sre_regex_t *re);

/* ---- Pike VM: sub-match captures + regex id (reference sregex.h:123-134) ---- */
typedef struct sre_vm_pike_ctx_s  sre_vm_pike_ctx_t;

SRE_API sre_vm_pike_ctx_t *sre_vm_pike_create_ctx(sre_pool_t *pool,
    sre_program_t *prog, sre_int_t *ovector, size_t ovecsize);

SRE_API sre_int_t sre_vm_pike_exec(sre_vm_pike_ctx_t *ctx, sre_char *input,
    size_t len, unsigned eof, sre_int_t **pending_matched);

/* ---- Thompson VM: match / no match (reference sregex.h:137-148) ---- */
typedef struct sre_vm_thompson_ctx_s  sre_vm_thompson_ctx_t;

SRE_API sre_vm_thompson_ctx_t *sre_vm_thompson_create_ctx(sre_pool_t *pool,
    sre_program_t *prog);

SRE_API sre_int_t sre_vm_thompson_exec(sre_vm_thompson_ctx_t *ctx,
    sre_char *input, size_t len, unsigned eof);

/* ---- Thompson JIT (reference sregex.h:151-171).  The x86-64 DynASM JIT is
 * dropped; the symbols stay so clients link, and jit_compile() answers
 * SRE_DECLINED, which both reference clients already handle
 * (src/sre_cli.c:419-424, bench/sregex.c:260-263). ---- */
typedef struct sre_vm_thompson_code_s  sre_vm_thompson_code_t;

typedef sre_int_t (*sre_vm_thompson_exec_pt)(sre_vm_thompson_ctx_t *ctx,
    sre_char *input, size_t size, unsigned eof);

SRE_API sre_int_t sre_vm_thompson_jit_compile(sre_pool_t *pool,
    sre_program_t *prog, sre_vm_thompson_code_t **pcode);
SRE_API sre_vm_thompson_ctx_t *sre_vm_thompson_jit_create_ctx(sre_pool_t *pool,
    sre_program_t *prog);
SRE_API sre_vm_thompson_exec_pt
    sre_vm_thompson_jit_get_handler(sre_vm_thompson_code_t *code);
SRE_API sre_int_t sre_vm_thompson_jit_free(sre_vm_thompson_code_t *code);

#ifdef __cplusplus
}
#endif

#endif /* SREGEX_AMD_SREGEX_H */
