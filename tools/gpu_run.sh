#!/bin/bash
# Runs a list of GPU steps, each under its own time limit; stops at the first step that dies by
# signal / timeout (a test failure with rc=1 does not stop the list).  Usage:
#   tools/gpu_run.sh "<seconds>|<name>|<command>" ...
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-.}"
for spec in "$@"; do
  secs="${spec%%|*}"; rest="${spec#*|}"; name="${rest%%|*}"; cmd="${rest#*|}"
  echo "=== [$name] (limit ${secs}s): $cmd"
  start=$(date +%s)
  timeout -k 10 "$secs" bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "=== [$name] rc=$rc in $(( $(date +%s) - start ))s"
  tail -n 25 "gpurun_out/$name.log"
  if [ $rc -ge 124 ] || [ $rc -lt 0 ]; then
    echo "=== stopping: step $name ended with rc=$rc"
    exit $rc
  fi
done
exit 0
