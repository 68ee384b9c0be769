#!/bin/bash
# gpurun with retries while the pod's GPU slots are busy (exit 3 / "transient"): gpurun_retry.sh <timeout> <command...>
T=$1; shift
for i in 1 2 3 4 5 6 7 8 9 10; do
  out=$(timeout $((T + 900)) gpurun --timeout $T -- "$@" 2>&1)
  echo "$out" | tail -${TAILN:-15}
  echo "$out" | grep -q "status=transient" || exit 0
  sleep 90
done
