/*
 * sre_scan_host.h — host handle of the scanner's device tables.
 */
#ifndef SRE_SCAN_HOST_H
#define SRE_SCAN_HOST_H

#include "sre_dfa.h"
#include "sre_hip_scan.h"
#include <vector>

struct sre_scan_device_tables_t {
    sre_scan_tables_t   h;          /* host copy (device pointers inside) */
    sre_scan_tables_t  *d_tab;      /* device copy */
    std::vector<void *> owned;
};

/* NULL + *why when the automaton does not fit the scanner (see sre_hip_scan.h) */
sre_scan_device_tables_t *sre_scan_tables_build(const sre_program_t *prog, const sre_dfa_t *d,
    int mode, const char **why);
void sre_scan_tables_release(sre_scan_device_tables_t *t);

#endif
