#!/bin/bash
# cfg4 without Python: N processes of tests/harness/scan_node (C++ over the C ABI), one per GPU of this node.
#   tools/run_scan_node.sh <n_gpus> [streams_total=4*n_gpus] [epochs_per_stream=4096] [steps=50]
N=${1:-1}; S=${2:-$((4 * N))}; E=${3:-4096}; K=${4:-50}
ID=$(mktemp -u /tmp/crn_rccl_id.XXXXXX)
pids=()
for r in $(seq 0 $((N - 1))); do
  RANK=$r WORLD_SIZE=$N LOCAL_RANK=$r "$(dirname "$0")/../tests/harness/scan_node" "$S" "$E" "$K" "$ID" &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait "$p" || rc=1; done
rm -f "$ID"
exit $rc
