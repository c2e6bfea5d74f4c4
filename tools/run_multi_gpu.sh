#!/bin/bash
# tools/run_multi_gpu.sh <world> <dpx_main args...> -- one dpx_main process per GPU, each on its own shard of the pairs
# file (-rank r -world N -device r); the result blocks are concatenated in rank order (= input order) on stdout,
# each rank's header / statistics go to stderr.  DPX_SHARE_GPU=1 puts every rank on device 0 (rehearsal on a 1-GPU box).
# Exit status: non-zero if ANY rank failed (nothing is printed to stdout then -- no silently truncated output).
W=$1; shift
BIN="$(dirname "$0")/../dpx_gpu_genomics_project_amd/hostcpp/dpx_main"
TMP=$(mktemp -d)
PIDS=()
for r in $(seq 0 $((W-1))); do
  DEV=$r; [ -n "$DPX_SHARE_GPU" ] && DEV=0
  "$BIN" "$@" -rank $r -world $W -device $DEV > "$TMP/out.$r" &
  PIDS+=($!)
done
FAILED=0
for r in $(seq 0 $((W-1))); do
  wait "${PIDS[$r]}"; rc=$?
  if [ $rc -ne 0 ]; then echo "[rank $r] FAILED (exit status $rc)" >&2; cat "$TMP/out.$r" >&2; FAILED=1; fi
done
if [ $FAILED -ne 0 ]; then rm -rf "$TMP"; exit 1; fi
for r in $(seq 0 $((W-1))); do
  awk -v r=$r '/^Pair # \| Score$/ {p=1; next} /^Elapsed time/ {p=0} { if (p) print; else print "[rank " r "] " $0 > "/dev/stderr" }' "$TMP/out.$r"
done
rm -rf "$TMP"
