/*
 * sre_hip_runtime.cpp — device discovery, loud failure, and the device image
 * of a compiled program.
 */
#include "sre_hip_runtime.h"
#include "sre_pwave.h"
#include "sre_dfa.h"
#include "sre_nfa.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

extern "C" int
sre_hip_fail(const char *what, hipError_t err)
{
    fprintf(stderr, "[sregex-hip] %s failed: %s\n", what, hipGetErrorString(err));
    return -1;
}

extern "C" int
sre_hip_ready(void)
{
    static int state = 0;   /* 0 unknown, 1 ok, -1 unusable */
    if (state == 0) {
        int        n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        if (e != hipSuccess || n <= 0) {
            fprintf(stderr,
                    "[sregex-hip] no usable HIP device (%s): the matcher runs on "
                    "gfx950 only, there is no CPU fallback\n",
                    e != hipSuccess ? hipGetErrorString(e) : "0 devices");
            state = -1;
        } else {
            state = 1;
        }
    }
    return state == 1 ? 0 : -1;
}

/* compute units of the current device (hipDeviceProp_t.multiProcessorCount; MI355X: 256) */
extern "C" int
sre_hip_cu_count(void)
{
    static int cached_dev = -1, cached_cus = 0;
    int        dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (dev != cached_dev) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached_cus = n;
        cached_dev = dev;
    }
    return cached_cus;
}

static void
free_program_image(void *data)
{
    sre_hip_program_s *dp = static_cast<sre_hip_program_s *>(data);
    if (dp->d_blob) {
        (void) hipFree(dp->d_blob);
    }
    sre_dfa_free(dp->dfa_pike);
    sre_dfa_free(dp->dfa_thompson);
    if (dp->d_pwave) (void) hipFree(dp->d_pwave);
    free(dp->h_pwave);
    free(dp);
}

extern "C" void *
sre_hip_program_pwave(struct sre_hip_program_s *dp, sre_program_t *prog)
{
    if (dp->pwave_tried) return dp->d_pwave;
    dp->pwave_tried = 1;
    if (getenv("SRE_HIP_NO_PWAVE") != NULL) return NULL;
    sre_pwave_hdr_t *h = sre_pwave_build(prog);
    if (h == NULL) return NULL;
    void *d = NULL;
    if (!sre_pwave_fits(h) || hipMalloc(&d, h->bytes) != hipSuccess
        || hipMemcpy(d, h, h->bytes, hipMemcpyHostToDevice) != hipSuccess)
    {
        if (d) (void) hipFree(d);
        free(h);
        return NULL;
    }
    dp->h_pwave = h;
    dp->d_pwave = d;
    return d;
}

extern "C" struct sre_hip_program_s *
sre_hip_program_get(sre_program_t *prog)
{
    if (prog->dev) {
        /* one process drives one GPU (multi-GPU): the image, the scanners
         * and the stream contexts built from this program live on the device that
         * was current when it was first used */
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess || cur != prog->dev->device) {
            fprintf(stderr, "[sregex-hip] program was bound to device %d but device %d is current: "
                            "compile one program per device\n", prog->dev->device, cur);
            return NULL;
        }
        return prog->dev;
    }
    if (sre_hip_ready() != 0) {
        return NULL;
    }

    /* distinct 256-bit membership bitmaps, NOTIN stored negated */
    std::vector<uint32_t> classes;
    std::vector<uint16_t> cls_of(prog->len, 0);
    for (uint32_t pc = 0; pc < prog->len; pc++) {
        const sre_insn_t &in = prog->insns[pc];
        if (in.opcode != SRE_OP_IN && in.opcode != SRE_OP_NOTIN) continue;
        uint32_t bm[8] = {0};
        for (unsigned c = 0; c < 256; c++) {
            int hit = sre_in_ranges(&prog->ranges[in.x], in.nranges, c);
            if (hit == (in.opcode == SRE_OP_IN)) bm[c >> 5] |= 1u << (c & 31);
        }
        size_t n = classes.size() / 8, k;
        for (k = 0; k < n; k++) {
            if (memcmp(&classes[k * 8], bm, 32) == 0) break;
        }
        if (k == n) classes.insert(classes.end(), bm, bm + 8);
        if (k > 0xffff) {
            fprintf(stderr, "[sregex-hip] too many distinct character classes\n");
            return NULL;
        }
        cls_of[pc] = (uint16_t) k;
    }
    uint32_t nclasses = (uint32_t) (classes.size() / 8);

    size_t               bytes = sre_dev_prog_bytes(prog->len, nclasses, prog->nregexes, prog->nleading);
    /* the wave form (sre_hip_common.h), when the program has one */
    sre_dev_wave_t wave;
    bool           has_wave = false;
    {
        sre_nfa_t *n = sre_nfa_build2(prog, SRE_NFA_SA_OFF, NULL);
        if (n != NULL && n->nassert == 0) {
            memset(&wave, 0, sizeof(wave));
            wave.init0 = n->init[0];
            wave.match = n->match_bits;
            memcpy(wave.accept, n->accept, sizeof(wave.accept));
            for (uint32_t i = 0; i < n->nbits; i++) {
                /* follow of bit i = its slice's entry for the value with only that bit set */
                const uint64_t f = n->follow[(size_t) (i / 8) * 256 + (1u << (i % 8))];
                for (uint32_t q = 0; q < n->nbits; q++) {
                    if ((f >> q) & 1) wave.pred[q] |= 1ull << i;
                }
            }
            has_wave = true;
        }
        sre_nfa_free(n);
    }
    const size_t wave_off = bytes;
    if (has_wave) bytes += SRE_DEV_ALIGN(sizeof(sre_dev_wave_t));
    std::vector<uint8_t> img(bytes, 0);
    sre_dev_prog_hdr_t  *h = reinterpret_cast<sre_dev_prog_hdr_t *>(img.data());
    if (has_wave) {
        h->wave_off = (uint32_t) wave_off;
        memcpy(img.data() + wave_off, &wave, sizeof(wave));
    }
    h->len = prog->len;
    h->nslots = prog->nslots;
    h->nregexes = prog->nregexes;
    h->nthreads = prog->nthreads;
    h->nclasses = nclasses;
    h->nleading = prog->nleading;
    h->leading_byte = prog->leading_byte;
    sre_dev_insn_t *di = reinterpret_cast<sre_dev_insn_t *>(img.data() + sre_dev_prog_insns_off());
    for (uint32_t pc = 0; pc < prog->len; pc++) {
        const sre_insn_t &in = prog->insns[pc];
        di[pc].opcode = in.opcode;
        di[pc].ch = in.ch;
        di[pc].cls = cls_of[pc];
        di[pc].x = in.x;
        di[pc].y = in.y;
        di[pc].arg = in.arg;
    }
    if (nclasses) {
        memcpy(img.data() + sre_dev_prog_classes_off(prog->len), classes.data(),
               (size_t) nclasses * 32);
    }
    memcpy(img.data() + sre_dev_prog_ncaps_off(prog->len, nclasses), prog->multi_ncaps,
           (size_t) prog->nregexes * 4);
    if (prog->nleading) {
        memcpy(img.data() + sre_dev_prog_leading_off(prog->len, nclasses, prog->nregexes),
               prog->leading_insns, (size_t) prog->nleading * 4);
    }

    sre_hip_program_s *dp = static_cast<sre_hip_program_s *>(calloc(1, sizeof(*dp)));
    if (dp == NULL) return NULL;
    SRE_HIP_TRY(hipGetDevice(&dp->device));
    SRE_HIP_TRY(hipMalloc(&dp->d_blob, bytes));
    SRE_HIP_TRY(hipMemcpy(dp->d_blob, img.data(), bytes, hipMemcpyHostToDevice));
    dp->blob_bytes = bytes;
    dp->has_wave = has_wave ? 1 : 0;
    dp->nclasses = nclasses;
    dp->pike_layout = sre_pike_layout(prog->len, prog->nthreads, prog->nslots);
    dp->thompson_layout = sre_thompson_layout(prog->len);
    if (sre_pool_add_cleanup(prog->pool, free_program_image, dp) != SRE_OK) goto hip_failed;
    prog->dev = dp;
    return dp;

hip_failed:
    if (dp->d_blob) (void) hipFree(dp->d_blob);
    free(dp);
    return NULL;
}
