/*
 * sre_hip_scan.h — launchers of sre_hip_scan.hip.
 */
#ifndef SRE_HIP_SCAN_H
#define SRE_HIP_SCAN_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#define SRE_CEILING_GRID 2048

#ifdef __cplusplus
extern "C" {
#endif
hipError_t sre_launch_gen_data(void *d_dst, uint64_t n, uint64_t tail_len, const void *d_tail,
    hipStream_t stream);
hipError_t sre_launch_read_ceiling(const void *d_src, uint64_t n, uint32_t *d_sink,
    hipStream_t stream);
#ifdef __cplusplus
}
#endif
#endif
