#!/bin/bash
# Run GPU steps in order, logging each to gpurun_out/<name>.log; stop at the first step that times out / is killed
# (a hung or killed GPU step must not be followed by another one), carry on after ordinary failures.
#   tools/gpu_steps.sh "name1|timeout_s|command ..." "name2|..."
mkdir -p gpurun_out
rc_all=0
for spec in "$@"; do
  name="${spec%%|*}"; rest="${spec#*|}"; tmo="${rest%%|*}"; cmd="${rest#*|}"
  echo "=== step $name (timeout ${tmo}s): $cmd"
  timeout -k 10 "$tmo" bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "=== step $name rc=$rc"; tail -n 6 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name timed out / killed: stopping"; exit $rc; fi
  [ $rc -ne 0 ] && rc_all=$rc
done
exit $rc_all
