#!/usr/bin/env python3
"""Measure what the scanner's staging access pattern alone reaches on this GPU.

Rows of seg_bytes per lane, TILE bytes of each row per round (see
sre_hip_read_pattern in include/sregex_hip.h), at several workgroups-per-CU
settings, next to the plain streaming read.  Prints one JSON object.
"""
import ctypes
import json
import sys

import torch

import sregex_amd as S


def main():
    nbytes = int(sys.argv[1]) if len(sys.argv) > 1 else (4 << 30)
    lib = S.load_library()
    torch.cuda.set_device(0)
    lib.sre_hip_set_device(0)
    buf = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    buf.fill_(97)
    stream = torch.cuda.current_stream()
    hs = ctypes.c_void_p(stream.cuda_stream)
    ptr = ctypes.c_void_p(buf.data_ptr())

    def timed(fn, reps=5):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            fn()
        e1.record(stream)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    out = {"bytes": nbytes}
    ms = timed(lambda: lib.sre_hip_read_ceiling(ptr, nbytes, hs))
    out["plain"] = {"ms": ms, "GBps": nbytes / ms / 1e6}
    for seg in (16640, 16384, 66560):
        for tile in (64, 128, 256):
            for lds in (16384, 40000, 53000, 80000):   # 4 / 3(4) / 3 / 2(1) workgroups per CU
                if lds > 65536:
                    continue
                rc = []
                ms = timed(lambda: rc.append(lib.sre_hip_read_pattern(ptr, nbytes, seg, tile, lds, hs)))
                assert all(r == 0 for r in rc), rc
                used = nbytes // seg * seg
                out[f"seg{seg}_tile{tile}_lds{lds}"] = {"ms": round(ms, 4), "GBps": round(used / ms / 1e6, 1)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
