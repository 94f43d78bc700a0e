#!/bin/bash
# PMC passes over tools/fine_block_timing.py (GPU box, repo root): where do the waves of fine_block_kernel spend their cycles?
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmc_fb; rm -rf $out; mkdir -p $out
rocprofv3 -L > $out/counters.txt 2>&1 || true
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM_WR SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $out/sq -o s -- python3 tools/fine_block_timing.py 48,48,48,48 > $out/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --output-format csv -d $out/sq2 -o s -- python3 tools/fine_block_timing.py 48,48,48,48 > $out/sq2.log 2>&1
rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum --output-format csv -d $out/tcp -o t -- python3 tools/fine_block_timing.py 48,48,48,48 > $out/tcp.log 2>&1
python3 - <<'PY'
import csv, glob, collections, os
out=os.environ.get('PWD')+'/gpurun_out/pmc_fb'
for sub in ('sq','sq2','tcp'):
    for f in glob.glob(out+'/'+sub+'/**/*counter_collection.csv', recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'][:60]
            if 'fine_block' in k: acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in acc.items():
            print(sub,k)
            for c,vals in v.items(): print('   %-34s n=%d mean=%.4g' % (c,len(vals),sum(vals)/len(vals)))
PY
