#!/bin/bash
# cfg4 without Python: N processes of tests/harness/scan_node (C++ over the C ABI), one per GPU of this node.
#   tools/run_scan_node.sh <n_gpus> [streams_total=4*n_gpus] [epochs_per_stream=4096] [steps=50]
#   ONE_GPU=1: every rank on device 0 (a one-GPU box; needs a stand-in for RCCL: CRN_RCCL_LIB=tests/harness/libfake_rccl_mp.so)
N=${1:-1}; S=${2:-$((4 * N))}; E=${3:-4096}; K=${4:-50}
ID=${CRN_SCAN_ID_FILE:-$(mktemp -u /tmp/crn_rccl_id.XXXXXX)}
# no leftovers of a killed run under the same name may vouch for ranks of this one (scan_node trusts "<id>.rank<r>" files)
rm -f "$ID" "$ID".rank* "$ID".rank*.tmp
trap 'rm -f "$ID" "$ID".rank* "$ID".rank*.tmp' EXIT
pids=()
for r in $(seq 0 $((N - 1))); do
  RANK=$r WORLD_SIZE=$N LOCAL_RANK=$([ -n "$ONE_GPU" ] && echo 0 || echo $r) "$(dirname "$0")/../tests/harness/scan_node" "$S" "$E" "$K" "$ID" &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait "$p" || rc=1; done
exit $rc
