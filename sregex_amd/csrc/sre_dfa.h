/*
 * sre_dfa.h — step automaton of a compiled program (host-side construction).
 * See sre_dfa.cpp.
 */
#ifndef SRE_DFA_H
#define SRE_DFA_H

#include "sre_program.h"

struct sre_dfa_s;
typedef struct sre_dfa_s sre_dfa_t;

#ifdef __cplusplus
extern "C" {
#endif
void sre_dfa_free(sre_dfa_t *dfa);
#ifdef __cplusplus
}
#endif
#endif
