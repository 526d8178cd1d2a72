#!/bin/bash
# runs GPU steps one after the other on the gpurun box; a step that is killed at its time limit ends the call
# (no further GPU step after a hang), a step that merely fails does not.  usage: step <seconds> <logfile> <command...>
mkdir -p gpurun_out
step() {
  local limit=$1 log=$2; shift 2
  echo "== $(date +%T) $* (limit ${limit}s) -> $log"
  timeout -k 10 "$limit" "$@" > "gpurun_out/$log" 2>&1
  local rc=$?
  echo "   rc=$rc"; tail -3 "gpurun_out/$log" | cut -c1-300
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step hit its time limit: stopping"; exit $rc; fi
  return 0
}
