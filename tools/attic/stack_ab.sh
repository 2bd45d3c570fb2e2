#!/bin/bash
# Same-box A/B of LocalFeatureTransformer.stack_self at small batch: bash tools/stack_ab.sh (GPU box)
for p in 1 4; do
  for i in 1 2; do
    for mode in stacked separate; do
      if [ $mode = separate ]; then export FAR_NO_STACK=1; else unset FAR_NO_STACK; fi
      python bench.py --pairs $p --steps 20 --no-cpu-baseline --no-other-modes 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pairs $p $mode', d['value'], d['ms_per_step'])"
    done
  done
done
