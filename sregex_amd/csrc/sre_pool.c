/*
 * sre_pool.c — region allocator behind sre_create_pool / sre_reset_pool /
 * sre_destroy_pool (public contract: reference src/sregex/sregex.h:82-84).
 *
 * Same contract as the reference's nginx-style pool (sre_palloc.c:26-148):
 * everything allocated from a pool lives until the pool is reset or destroyed,
 * there is no per-object free, and a pool carries a list of cleanup handlers
 * that run on destroy only — the HIP layer hangs its device buffers there.
 * The implementation is a plain chunk list: bump allocation inside 16-B
 * aligned chunks, oversized requests get a chunk of their own.
 */
#include "sre_program.h"
#include <string.h>

typedef struct sre_chunk_s {
    struct sre_chunk_s *next;
    size_t              size;   /* payload bytes */
    size_t              used;
    /* payload follows, 16-B aligned */
} sre_chunk_t;

typedef struct sre_cleanup_s {
    struct sre_cleanup_s *next;
    sre_pool_cleanup_pt   handler;
    void                 *data;
} sre_cleanup_t;

struct sre_pool_s {
    sre_chunk_t   *head;       /* current chunk (newest first) */
    size_t         chunk_size;
    sre_cleanup_t *cleanups;
};

#define SRE_ALIGN16(n)  (((n) + 15u) & ~(size_t) 15u)
#define SRE_CHUNK_HDR   SRE_ALIGN16(sizeof(sre_chunk_t))

static sre_chunk_t *
sre_chunk_new(size_t payload)
{
    sre_chunk_t *c = malloc(SRE_CHUNK_HDR + payload);
    if (c == NULL) {
        return NULL;
    }
    c->next = NULL;
    c->size = payload;
    c->used = 0;
    return c;
}

SRE_API sre_pool_t *
sre_create_pool(size_t size)
{
    sre_pool_t *pool = malloc(sizeof(sre_pool_t));
    if (pool == NULL) {
        return NULL;
    }
    if (size < 256) {
        size = 256;
    }
    pool->chunk_size = SRE_ALIGN16(size);
    pool->cleanups = NULL;
    pool->head = sre_chunk_new(pool->chunk_size);
    if (pool->head == NULL) {
        free(pool);
        return NULL;
    }
    return pool;
}

static void
sre_pool_free_chunks(sre_chunk_t *c)
{
    while (c) {
        sre_chunk_t *n = c->next;
        free(c);
        c = n;
    }
}

SRE_API void
sre_reset_pool(sre_pool_t *pool)
{
    /* keep one regular chunk, drop the rest; cleanup handlers stay armed */
    sre_chunk_t *keep = NULL, *c = pool->head;
    while (c) {
        sre_chunk_t *n = c->next;
        if (keep == NULL && c->size == pool->chunk_size) {
            keep = c;
        } else {
            free(c);
        }
        c = n;
    }
    if (keep == NULL) {
        keep = sre_chunk_new(pool->chunk_size);
    }
    if (keep) {
        keep->next = NULL;
        keep->used = 0;
    }
    pool->head = keep;
}

SRE_API void
sre_destroy_pool(sre_pool_t *pool)
{
    sre_cleanup_t *cl = pool->cleanups;
    while (cl) {
        /* handler records live in malloc memory of their own so that a
         * preceding reset cannot have recycled them */
        sre_cleanup_t *n = cl->next;
        if (cl->handler) {
            cl->handler(cl->data);
        }
        free(cl);
        cl = n;
    }
    sre_pool_free_chunks(pool->head);
    free(pool);
}

SRE_NOAPI void *
sre_palloc(sre_pool_t *pool, size_t size)
{
    sre_chunk_t *c = pool->head;
    size = SRE_ALIGN16(size ? size : 1);
    if (c == NULL || c->size - c->used < size) {
        size_t payload = size > pool->chunk_size ? size : pool->chunk_size;
        sre_chunk_t *n = sre_chunk_new(payload);
        if (n == NULL) {
            return NULL;
        }
        if (size > pool->chunk_size && c != NULL) {
            /* oversized: park it behind the current chunk, keep bumping there */
            n->next = c->next;
            c->next = n;
        } else {
            n->next = c;
            pool->head = n;
        }
        c = n;
    }
    void *p = (char *) c + SRE_CHUNK_HDR + c->used;
    c->used += size;
    return p;
}

SRE_NOAPI void *
sre_pcalloc(sre_pool_t *pool, size_t size)
{
    void *p = sre_palloc(pool, size);
    if (p) {
        memset(p, 0, size);
    }
    return p;
}

SRE_NOAPI int
sre_pool_add_cleanup(sre_pool_t *pool, sre_pool_cleanup_pt handler, void *data)
{
    sre_cleanup_t *cl = malloc(sizeof(sre_cleanup_t));
    if (cl == NULL) {
        return SRE_ERROR;
    }
    cl->handler = handler;
    cl->data = data;
    cl->next = pool->cleanups;
    pool->cleanups = cl;
    return SRE_OK;
}
