#!/bin/bash
# A/B runs on ONE box (boxes differ by several percent): experiment knobs.
#   SRE_HIP_LDS_PAD      extra dynamic LDS per workgroup (fewer workgroups per CU)
#   SRE_HIP_SEG_BYTES    segment size;  SRE_HIP_SEG_CAP  largest segment the automatic choice makes
#   SRE_HIP_NO_SHADOW    no shadow rows (stable-stretch tracking off)
#   SRE_BENCH_ONE_STREAM bench.py: both scanners on one HIP stream
#   SREGEX_AMD_LIB       another build of the library (an older commit, another tile layout)
one() {
  local name=$1; shift
  local out
  out=$(env "$@" python bench.py --no-variants --no-cpu-baseline --steps 10 --config $CFG $EXTRA 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('step_ms %.4f kernel_ms %.4f frac %.4f kfrac %.4f seg %s' % (r['step_ms'], r['kernel_ms'], r['frac'], r['kernel_frac'], d['config']['segment_bytes']))")
  echo "$CFG $EXTRA | $name | $out"
}
C4=$PWD/sregex_amd/lib_c4/libsregex.so
for CFG in cfg3 dense; do
EXTRA=
one "wide, 2 per CU (default)" A=1
one "narrow, 3 per CU" SRE_HIP_NO_WIDE4=1 SRE_HIP_LDS_PAD=0
one "narrow, 2 per CU" SRE_HIP_NO_WIDE4=1
one "narrow, 4 per CU (66 spills)" SRE_HIP_NO_WIDE4=1 SRE_HIP_LDS_PAD=0 SREGEX_AMD_LIB=$C4
one "wide, 2 per CU again" A=1
done
