# round 3, final tree: what changes with the last kernel-source edits — suite, default bench line + its kernel-trace stats, the reference-order lines, PMC passes.
#   usage: TAG=r3x RL_COMMIT=<sha> bash scratch/round3_refresh.sh
R=$GRAFT_REPO_ROOT; cd $R
T=${TAG:-r3x}; O=$R/gpurun_out/$T; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" > $O/pytest.log; cat $O/pytest.log
( time python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; tail -n 3 $O/bench_default.err
python bench.py --stream-mode reference --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_reference.json 2> /dev/null
python bench.py --stream-mode reference --numerics fast --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_reference_fast.json 2> /dev/null
python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline 2> $O/bench_2rank.err | grep "^{" > $O/bench_2rank.json
python scratch/mc_ref_bench.py 16 > $O/mc_ref.log 2>&1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_default -o p -- python $R/bench.py --no-cpu-baseline > $O/stats_default.log 2>&1
cd $R
bash scratch/pmc_collect.sh $T/pmc_cbox k_path_fused > $O/pmc_cbox.log 2>&1
bash scratch/pmc_collect.sh $T/pmc_reference k_stream_chain --stream-mode reference > $O/pmc_reference.log 2>&1
bash scratch/pmc_collect.sh $T/pmc_living k_path_fused --scene living_room > $O/pmc_living.log 2>&1
bash scratch/pmc_collect.sh $T/pmc_living_fast k_path_fused --scene living_room --numerics fast > $O/pmc_living_fast.log 2>&1
bash scratch/pmc_collect.sh $T/pmc_medium k_path_fused --scene cbox_medium > $O/pmc_medium.log 2>&1
( time timeout 1500 python tests/parity_fuzz.py ${FUZZ_SECONDS:-300} ${FUZZ_SEED:-6006} ) > $O/fuzz.log 2>&1; tail -n 4 $O/fuzz.log
for f in bench_default bench_reference bench_reference_fast bench_2rank; do grep "^{" $O/$f.json | python -c "import sys,json; o=json.loads(sys.stdin.read()); print('$f', round(o['value'],1), round(o['ms_per_step'],2), {k: round(v['avg_launch_ms'],2) for k,v in o['roofline']['kernels'].items()}, o['n_gpus'], o['distributed']['crc_match'], o['distributed']['image_crc32'], o.get('reference_order_value'))"; done
find $O -name '*kernel_trace.csv' -size +2M -delete; find $O -name '*agent_info.csv' -delete; find $O -name '*domain_stats.csv' -delete; find $O -name '*counter_collection.csv' -size +2M -delete
