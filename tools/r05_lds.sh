#!/bin/bash
# LDS bank conflicts of uniformly random accesses (tools/micro/lds_random.hip) -> gpurun_out/r05_lds_random.txt
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ldsr
timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d /tmp/ldsr -o p -- $R/tools/micro/lds_random > /dev/null 2>&1
python3 - <<'PY' | tee $GRAFT_REPO_ROOT/gpurun_out/r05_lds_random.txt
import csv, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open('/tmp/ldsr/p_counter_collection.csv')):
    acc[r['Kernel_Name'].split('(')[0]][r['Counter_Name']] += float(r['Counter_Value'])
print('tools/micro/lds_random.hip on MI355X: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE for 64 lanes of a wave on')
names = {'k_lds<0, 4096>': 'consecutive words, returning atomic OR', 'k_lds<1, 512>': 'uniformly random words of a 2^14-bit bitmap, returning atomic OR',
         'k_lds<1, 4096>': 'uniformly random words of a 2^17-bit bitmap, returning atomic OR', 'k_lds<2, 4096>': 'uniformly random words, 4-byte reads',
         'k_lds<3, 8192>': 'uniformly random 16-byte buckets (ds_read_b128)'}
for k, v in sorted(acc.items()):
    kk = k.replace('void ', '')
    if v.get('SQ_LDS_IDX_ACTIVE'):
        print(f"  {names.get(kk, kk):75s} {v['SQ_LDS_BANK_CONFLICT'] / v['SQ_LDS_IDX_ACTIVE']:.3f}   ({v['SQ_INSTS_LDS'] / 3:.0f} LDS instructions per launch)")
PY
