#!/bin/bash
# same-box sweep of the pipeline shape: bench.py --head-group g --encoder-streams e  (ms per step, roofline.frac), interleaved rounds
# usage: tools/sweep_pipeline.sh "--head-group 2" "--head-group 4" ...     (no arguments: the round-5 set)
if [ $# -eq 0 ]; then
  set -- "--encoder-streams 2 --head-group 4" "--encoder-streams 1 --head-group 4" "--encoder-streams 3 --head-group 4" "--encoder-streams 2 --head-group 5" "--encoder-streams 2 --head-group 8" "--encoder-streams 3 --head-group 6"
fi
for i in 1 2; do
  for cfg in "$@"; do
    echo "== [$cfg] round $i"
    python bench.py --no-cpu-baseline --steps 20 --warmup 5 $cfg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), round(d['roofline']['frac'],4))"
  done
done
