#!/bin/bash
# Rehearse the N-rank RCCL path on a single GPU (all ranks forced onto device 0).  RCCL may refuse duplicate GPUs; in that
# case the log says so and nothing else is learnt.  usage: tools/mgpu_rehearsal.sh <nranks> <logfile>
N=${1:-2}; LOG=${2:-gpurun_out/mgpu_rehearsal.log}
export MASTER_ADDR=127.0.0.1 MASTER_PORT=$((20000 + $$ % 20000)) WORLD_SIZE=$N QUDA_AMD_FORCE_DEVICE=0
# RCCL refuses N ranks on one device ("Duplicate GPU detected"): use the file-based rehearsal transport instead
export QUDA_AMD_TRANSPORT=shm QUDA_AMD_SHM_DIR=$(mktemp -d /dev/shm/quda_amd_XXXXXX)
pids=()
for r in $(seq 0 $((N-1))); do
  RANK=$r LOCAL_RANK=$r timeout -k 5 150 python3 -u tools/mgpu_check.py $N > ${LOG}.rank$r 2>&1 &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=$?; done
for r in $(seq 0 $((N-1))); do echo "--- rank $r ---"; cat ${LOG}.rank$r; done > $LOG
echo "rehearsal rc=$rc" >> $LOG
rm -rf $QUDA_AMD_SHM_DIR
exit $rc
