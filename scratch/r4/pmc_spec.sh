# counters of k_stream_spec / k_path_fused on cbox 1080p x 128 spp, reference-order streams (separate --pmc passes, kernel-trace only)
O=gpurun_out/r4pmc; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS" "SQ_INSTS_FLAT SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SENDMSG SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  RL_SPEC_ONLY=1 timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/$O/p$i -o p -- python $R/scratch/r4/dev_spec.py 1920 1080 128 cbox > $R/$O/p$i.log 2>&1
done
cd $R
python - <<PY
import csv, glob, collections, json
out = {}
for d in sorted(glob.glob("$O/p*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split('(')[0][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, v in acc.items():
            if "spec" in k or "fused" in k or "chain" in k: out.setdefault(k, {}).update({a: b for a, b in v.items()})
json.dump(out, open("$O/summary.json", "w"), indent=1)
for k, v in out.items(): print(k, {a: f"{b:.4g}" for a, b in v.items()})
PY
find $O -name '*.csv' -size +2M -delete
