cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=${OUT:-gpurun_out/pmc_fused}; SIMD=${SIMD:-0}; rm -rf $OUT; mkdir -p $OUT
i=0
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rocprofv3 --pmc $c --kernel-include-regex 'k_splat_fused' --output-format csv -d $OUT/p$i -o run -- python bench.py --main-only --steps 1 --warmup 1 --simd $SIMD > $OUT/p$i.log 2>&1
done
python - $OUT <<'PY'
import csv,glob,collections,sys
for f in sorted(glob.glob(sys.argv[1] + '/p*/**/run_counter_collection.csv', recursive=True)):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:40]; acc[k][r['Counter_Name']]+=float(r['Counter_Value']); 
        n[(k,r['Counter_Name'])]+=1
    for k,v in acc.items():
        print(k, {c:"%.4g"%(x/n[(k,c)]) for c,x in v.items()})
PY
