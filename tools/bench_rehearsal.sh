#!/bin/bash
# Rehearse `bench.py --gpus N` (the N>1 code path the driver launches with torch.distributed.run) on ONE GPU: N processes on
# device 0, collectives over the file transport, halo over the IPC peer-store transport.  usage: tools/bench_rehearsal.sh <N> <log> [bench args]
N=${1:-2}; LOG=${2:-gpurun_out/bench_rehearsal.log}; shift 2
export MASTER_ADDR=127.0.0.1 MASTER_PORT=$((20000 + $$ % 20000)) WORLD_SIZE=$N QUDA_AMD_FORCE_DEVICE=0
export QUDA_AMD_TRANSPORT=shm QUDA_AMD_SHM_DIR=$(mktemp -d /dev/shm/quda_amd_XXXXXX)
pids=()
for r in $(seq 0 $((N-1))); do
  RANK=$r LOCAL_RANK=$r timeout -k 5 ${REH_TIMEOUT:-200} python3 bench.py --gpus $N --no-cpu --no-extra "$@" > ${LOG}.rank$r 2>&1 &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=$?; done
cat ${LOG}.rank* > $LOG
echo "rehearsal rc=$rc" >> $LOG
rm -rf $QUDA_AMD_SHM_DIR
exit $rc
