#!/bin/bash
# A/B runs on ONE box (boxes differ by several percent): experiment knobs.
#   SRE_HIP_LDS_PAD      extra dynamic LDS per workgroup (fewer workgroups per CU)
#   SRE_HIP_SEG_BYTES    segment size;  SRE_HIP_SEG_CAP  largest segment the automatic choice makes
#   SRE_HIP_NO_SHADOW    no shadow rows (stable-stretch tracking off)
#   SRE_BENCH_STREAMS    bench.py: two (default for one stream) | one (default for many) | tail, see measure()
#   SREGEX_AMD_LIB       another build of the library (an older commit, another tile layout)
#   SRE_BENCH_DEPTH      bench.py: scanners (steps) in flight, default 2
#   SRE_BENCH_TRACE      bench.py: per-step wall times of every run to stderr
#   SRE_HIP_DMA_UPLOAD   stream descriptors of a batch by hipMemcpyAsync instead of a kernel
#   SRE_HIP_NO_PULL      compat API: pinned staging buffer copied by the DMA engine, not pulled by a kernel
#   SRE_HIP_NO_WIDE4     COUNT with 4 class bits: 8-bit tile indices
one() {
  local name=$1; shift
  local out
  out=$(env "$@" python bench.py --no-variants --no-cpu-baseline --steps 10 --config $CFG $EXTRA 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('step_ms %.4f kernel_ms %.4f frac %.4f kfrac %.4f seg %s' % (r['step_ms'], r['kernel_ms'], r['frac'], r['kernel_frac'], d['config']['segment_bytes']))")
  echo "$CFG $EXTRA | $name | $out"
}
for CFG in cfg2 cfg2m cfg4 cfg3; do
EXTRA=
one "one stream per scanner, scans chained by an event (default)" A=1
one "scans on one stream, tails on a second" SRE_BENCH_STREAMS=tail
one "... and three steps in flight" SRE_BENCH_STREAMS=tail SRE_BENCH_DEPTH=3
one "default again" A=1
done
