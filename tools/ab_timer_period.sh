for i in 1 2 3; do
  for cfg in "--timer-period 5" "--timer-period 1" "--timer-period 20"; do
    echo "== cfg [$cfg] round $i"
    python bench.py --no-cpu-baseline --steps 20 --warmup 3 $cfg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['ms_per_step'],2), round(r['frac'],4), r['timed_launches'], round(r['launches_per_step'],1), round(r['avg_launch_ms'],4))"
  done
done
