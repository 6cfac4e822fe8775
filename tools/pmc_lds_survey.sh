#!/bin/bash
# Runs ON THE GPU BOX: LDS bank-conflict survey of every kernel family -- one rocprofv3 pass (--kernel-trace + --pmc, the
# program itself after `--`) per workload, then per kernel SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (share of the LDS-array
# cycles that are conflict cycles) and the LDS-active share of the wave cycles.   bash tools/pmc_lds_survey.sh
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export GRAFT_REPO_ROOT=$ROOT
OUT=$ROOT/gpurun_out/r03lds
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CTR="SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_ADDR_CONFLICT"
rocprofv3 --kernel-trace --pmc $CTR -d $OUT/cfg4 -o pmc --output-format csv -- python3 $ROOT/bench.py --config 4 --scaling weak --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/cfg4.log 2>&1 || { tail -5 $OUT/cfg4.log; exit 1; }
rocprofv3 --kernel-trace --pmc $CTR -d $OUT/train -o pmc --output-format csv -- python3 $ROOT/tools/train_perf.py 2048 10 2 > $OUT/train.log 2>&1 || { tail -5 $OUT/train.log; exit 1; }
rocprofv3 --kernel-trace --pmc $CTR -d $OUT/trainc -o pmc --output-format csv -- python3 $ROOT/tools/train_perf.py 2048 10 2 conv3D > $OUT/trainc.log 2>&1 || { tail -5 $OUT/trainc.log; exit 1; }
rocprofv3 --kernel-trace --pmc $CTR -d $OUT/toy -o pmc --output-format csv -- python3 $ROOT/bench.py --config 2 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/toy.log 2>&1 || { tail -5 $OUT/toy.log; exit 1; }
rocprofv3 --kernel-trace --pmc $CTR -d $OUT/cfg3 -o pmc --output-format csv -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-configs-table --no-trained-ess > $OUT/cfg3.log 2>&1 || { tail -5 $OUT/cfg3.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections, os
base=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r03lds'
for d in ('cfg4','train','trainc','toy','cfg3'):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(base+'/'+d+'/*counter_collection.csv'):
        for row in csv.DictReader(open(f)):
            n=row['Kernel_Name']
            if 'l2hmc' not in n: continue
            acc[n.split('(')[0][-56:]][row['Counter_Name']].append(float(row['Counter_Value']))
    print('==',d)
    for k,v in sorted(acc.items()):
        m={c:sorted(x)[len(x)//2] for c,x in v.items()}
        act=m.get('SQ_LDS_IDX_ACTIVE',0) or 1
        if m.get('SQ_INSTS_LDS',0) < 1e4: continue
        print('%-58s n=%4d lds_insts %.2e  conflict/active %.3f  lds_active/wave_cycles %.3f' % (k, len(v['SQ_INSTS_LDS']), m['SQ_INSTS_LDS'], m.get('SQ_LDS_BANK_CONFLICT',0)/act, act/(4*m.get('SQ_WAVE_CYCLES',1) or 1)))
PY
find $OUT -name "*kernel_trace.csv" -size +10M -delete
find $OUT -name "*counter_collection.csv" -size +30M -delete
