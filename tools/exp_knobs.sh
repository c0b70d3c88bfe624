#!/bin/bash
# A/B runs on ONE box (boxes differ by a few percent): experiment knobs of the scan kernel.
#   SRE_HIP_LDS_PAD   extra dynamic LDS per workgroup (fewer workgroups per CU)
#   SRE_HIP_SEG_BYTES segment size
#   SRE_HIP_NO_SHADOW no shadow rows (stable-stretch tracking off)
#   SREGEX_AMD_LIB    another build of the library
one() {
  local name=$1; shift
  local out
  out=$(env "$@" python bench.py --no-variants --no-cpu-baseline --steps 10 --config $CFG $EXTRA 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('step_gpu_ms %.4f kernel_ms %.4f frac %.4f kfrac %.4f seg %s' % (r['step_gpu_ms'], r['kernel_ms'], r['frac'], r['kernel_frac'], d['config']['segment_bytes']))")
  echo "$CFG $EXTRA | $name | $out"
}
CFG=cfg3 EXTRA=
one "default (padded to 2 per CU)" A=1
for seg in 8448 12544 16640 24832 33024; do one "default seg=$seg" SRE_HIP_SEG_BYTES=$seg; done
CFG=cfg2
for pad in 0 10000 20000 45000; do one "pad=$pad" SRE_HIP_LDS_PAD=$pad; done
CFG=cfg4
for pad in 0 8000 20000; do one "pad=$pad" SRE_HIP_LDS_PAD=$pad; done
CFG=nfa
one "default" A=1
