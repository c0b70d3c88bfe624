#include "sre_dfa.h"
#include <stdlib.h>
struct sre_dfa_s { int dummy; };
extern "C" void sre_dfa_free(sre_dfa_t *dfa) { free(dfa); }
