/*
 * ddebug.h — compile-time debug printf, installed next to sregex.h because the
 * reference's clients include it (reference src/sregex/ddebug.h:12-24,
 * src/sre_cli.c:13).  Enable per translation unit with `#define DDEBUG 1`
 * before the include.
 */
#ifndef SREGEX_AMD_DDEBUG_H
#define SREGEX_AMD_DDEBUG_H

#include <stdio.h>

#if defined(DDEBUG) && (DDEBUG)
#   define dd(...)                                                           \
        do {                                                                 \
            fprintf(stderr, "sregex *** ");                                  \
            fprintf(stderr, __VA_ARGS__);                                    \
            fprintf(stderr, " at %s line %d.\n", __FILE__, __LINE__);        \
        } while (0)
#else
#   define dd(...)  do { } while (0)
#endif

#endif /* SREGEX_AMD_DDEBUG_H */
