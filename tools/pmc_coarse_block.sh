#!/bin/bash
# PMC passes over tools/coarse_block_timing.py: where do the waves of coarse_block_kernel<48, 24> spend their cycles?
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmc_cb; rm -rf $out; mkdir -p $out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES --output-format csv -d $out/sq -o s -- python3 tools/coarse_block_timing.py 48 96 24 > $out/sq.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --output-format csv -d $out/sq2 -o s -- python3 tools/coarse_block_timing.py 48 96 24 > $out/sq2.log 2>&1
rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr GRBM_GUI_ACTIVE --output-format csv -d $out/tcp -o t -- python3 tools/coarse_block_timing.py 48 96 24 > $out/tcp.log 2>&1
python3 - <<'PY'
import csv, glob, collections, os
out=os.environ.get('PWD')+'/gpurun_out/pmc_cb'
for sub in ('sq','sq2','tcp'):
    for f in glob.glob(out+'/'+sub+'/**/*counter_collection.csv', recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'][:60]
            if 'coarse_block' in k: acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in acc.items():
            print(sub,k)
            for c,vals in v.items(): print('   %-34s n=%d mean=%.4g' % (c,len(vals),sum(vals)/len(vals)))
PY
