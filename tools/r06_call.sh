#!/bin/bash
# One GPU call of round 6: a list of steps, each under its own timeout, everything logged to gpurun_out/r06_<tag>.log.
#   tools/r06_call.sh <tag> <step> [<step> ...]       step = "test:<pytest node ids / -k args>" | "ab:<ab_bench spec>" | "sh:<command>"
cd "$(dirname "$0")/.."
TAG=$1; shift
mkdir -p gpurun_out
LOG=gpurun_out/r06_${TAG}.log
: > $LOG
for step in "$@"; do
  kind=${step%%:*}; arg=${step#*:}
  echo "=== $step" >> $LOG
  t0=$(date +%s)
  case $kind in
    test) timeout 900 python -m pytest -m gpu -x -q -p no:cacheprovider $arg >> $LOG 2>&1 ;;
    ab)   timeout 900 tools/ab_bench.sh "$arg" >> $LOG 2>&1 ;;
    sh)   timeout 1200 bash -c "$arg" >> $LOG 2>&1 ;;
  esac
  echo "=== rc $? in $(( $(date +%s) - t0 )) s" >> $LOG
done
grep -E "^===|passed|failed|\[ab\]|error|Error" $LOG | tail -60
