# round 3 evidence run: suite, the default bench line (with `also`), every other bench line, kernel-trace stats of the same commands,
# PMC passes (-> profiles/pmc_live.json), fast-mode parity at full size, a fuzz run.    usage: TAG=r3z RL_COMMIT=<sha> bash scratch/round3_final.sh
R=$GRAFT_REPO_ROOT; cd $R
T=${TAG:-r3z}; O=$R/gpurun_out/$T; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" > $O/pytest.log; cat $O/pytest.log
( time python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; tail -3 $O/bench_default.err
python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline 2> $O/bench_2rank.err | grep "^{" > $O/bench_2rank.json
python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --scaling strong 2> $O/bench_2rank_strong.err | grep "^{" > $O/bench_2rank_strong.json
python bench.py --numerics fast --no-cpu-baseline > $O/bench_fast.json 2> /dev/null
python bench.py --stream-mode reference --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_reference.json 2> /dev/null
RL_REF_SINGLE_PASS=1 python bench.py --stream-mode reference --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_reference_single_pass.json 2> /dev/null
python bench.py --stream-mode reference --numerics fast --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_reference_fast.json 2> /dev/null
python bench.py --pipeline wavefront --no-cpu-baseline > $O/bench_wavefront.json 2> /dev/null
python bench.py --scene living_room --steps 3 --warmup 1 > $O/bench_living.json 2> /dev/null
python bench.py --scene living_room --steps 3 --warmup 1 --no-cpu-baseline --numerics fast > $O/bench_living_fast.json 2> /dev/null
python bench.py --scene living_room --spp 16 --stream-mode reference --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_living_reference_16spp.json 2> /dev/null
python bench.py --scene cbox_medium --steps 3 --warmup 1 > $O/bench_medium.json 2> /dev/null
python bench.py --scene cbox_medium --steps 2 --warmup 1 --no-cpu-baseline --numerics fast > $O/bench_medium_fast.json 2> /dev/null
python bench.py --scene cbox_medium --spp 16 --stream-mode reference --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_medium_reference_16spp.json 2> /dev/null
python bench.py --scene living_room --tris 4000000 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_living4m.json 2> /dev/null
python bench.py --scene living_room --tris 4000000 --steps 2 --warmup 1 --no-cpu-baseline --numerics fast > $O/bench_living4m_fast.json 2> /dev/null
python scratch/fast_parity.py > $O/fast_parity.json 2> $O/fast_parity.err
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_default -o p -- python $R/bench.py --no-cpu-baseline > $O/stats_default.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_living_fast -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --scene living_room --numerics fast > $O/stats_living_fast.log 2>&1
cd $R
bash scratch/pmc_collect.sh $T/pmc_cbox k_path_fused > $O/pmc_cbox.log 2>&1
bash scratch/pmc_collect.sh $T/pmc_reference k_stream_chain --stream-mode reference > $O/pmc_reference.log 2>&1
bash scratch/pmc_collect.sh $T/pmc_living k_path_fused --scene living_room > $O/pmc_living.log 2>&1
bash scratch/pmc_collect.sh $T/pmc_living_fast k_path_fused --scene living_room --numerics fast > $O/pmc_living_fast.log 2>&1
bash scratch/pmc_collect.sh $T/pmc_medium k_path_fused --scene cbox_medium > $O/pmc_medium.log 2>&1
( time timeout 1500 python tests/parity_fuzz.py ${FUZZ_SECONDS:-480} ${FUZZ_SEED:-3003} ) > $O/fuzz.log 2>&1; tail -4 $O/fuzz.log
( time timeout 1500 python tests/parity_fuzz.py ${FUZZ_SECONDS:-480} $((${FUZZ_SEED:-3003} + 1)) fast ) > $O/fuzz_fast.log 2>&1; tail -4 $O/fuzz_fast.log
for f in bench_default bench_2rank bench_2rank_strong bench_fast bench_reference bench_reference_single_pass bench_reference_fast bench_wavefront bench_living bench_living_fast bench_living_reference_16spp bench_living4m bench_living4m_fast bench_medium bench_medium_fast bench_medium_reference_16spp; do grep "^{" $O/$f.json | python -c "import sys,json; o=json.loads(sys.stdin.read()); print('$f', round(o['value'],1), round(o['ms_per_step'],2), {k: round(v['avg_launch_ms'],2) for k,v in o['roofline']['kernels'].items()}, o['n_gpus'], o['distributed']['crc_match'], o['distributed']['image_crc32'])"; done
find $O -name '*kernel_trace.csv' -size +2M -delete; find $O -name '*agent_info.csv' -delete; find $O -name '*domain_stats.csv' -delete; find $O -name '*counter_collection.csv' -size +2M -delete
