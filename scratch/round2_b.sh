# round 2, GPU call B: the split / LIGHTS-specialised build — suite, bench, kernel-trace stats, PMC passes for cbox and the living-room stand-in
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r2b; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err
RL_GENERIC_LIGHTS=1 python bench.py --no-cpu-baseline > $O/bench_generic_lights.json 2> /dev/null
timeout 900 python bench.py --stream-mode reference --pipeline fused --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_reference_fused.json 2> $O/bench_reference_fused.err
python bench.py --scene living_room --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_living.json 2> $O/bench_living.err
python bench.py --scene cbox_medium --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_medium.json 2> $O/bench_medium.err
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cbox -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/stats_cbox.log 2>&1
cd $R
bash scratch/pmc_collect.sh r2b/pmc_cbox k_path_fused > $O/pmc_cbox.log 2>&1
bash scratch/pmc_collect.sh r2b/pmc_living k_path_fused --scene living_room --spp 32 > $O/pmc_living.log 2>&1
tail -3 $O/pytest.log; for f in bench_default bench_generic_lights bench_reference_fused bench_living bench_medium; do grep "^{" $O/$f.json | python -c "import sys,json; o=json.loads(sys.stdin.read()); print('$f', o['value'], o['ms_per_step'], o['roofline']['avg_launch_ms'])"; done
