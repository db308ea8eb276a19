#!/bin/bash
# One GPU call's worth of checks: each program runs under its own time limit, in order, and the script stops at the
# first one that was killed, aborted or crashed (exit 124/137/134/139) -- a failing test alone (exit 1) does not stop it.
#   tools/gpu_step.sh <outdir> '<seconds> <command>' ...
OUT=$1; shift
mkdir -p "$OUT"
i=0
for item in "$@"; do
  secs=${item%% *}; cmd=${item#* }
  i=$((i+1))
  echo "== [$i] $cmd"
  timeout -k 10 "$secs" bash -c "$cmd" > "$OUT/step$i.out" 2> "$OUT/step$i.err"
  rc=$?
  echo "rc $rc" >> "$OUT/step$i.out"
  tail -c 1500 "$OUT/step$i.out"
  case $rc in 124|137|134|139) echo "stopping: step $i ended with $rc"; tail -c 2000 "$OUT/step$i.err"; exit $rc;; esac
done
