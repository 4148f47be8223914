#!/bin/bash
# same-box sweep of the pipeline shape: bench.py --head-group g --encoder-streams e  (ms per step, roofline.frac), interleaved rounds
# usage: tools/sweep_pipeline.sh "--head-group 2" "--head-group 4" ...
for i in 1 2 3; do
  for cfg in "$@"; do
    echo "== [$cfg] round $i"
    python bench.py --no-cpu-baseline --steps 20 --warmup 5 $cfg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), round(d['roofline']['frac'],4))"
  done
done
