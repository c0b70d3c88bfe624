#!/bin/bash
# differential fuzz on the GPU box with other seeds than the fixed suite's: every engine
# (scanner, NFA tier forced, exact VM), the compat API's CLI call sequence, chunked streams
# usage (repo root, through gpurun): bash tools/fuzz_campaign.sh SEED...
for seed in "$@"; do
  SRE_FUZZ_SEED=$seed SRE_FUZZ_BIG=1 python -m pytest tests/test_gpu_parity.py -x -q -m gpu \
      -k "random_patterns or chunked_streams_take or count_on_the_nfa" > gpurun_out/fuzz_$seed.log 2>&1
  echo "seed $seed: $(tail -1 gpurun_out/fuzz_$seed.log)"
done
