# round 2 evidence run: suite, bench lines (all configs + modes), kernel-trace stats, PMC passes (-> profiles/pmc_live.json), fast-mode parity at full size
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/${TAG:-r2z}; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --gpus 2 --steps 2 --warmup 1 2> $O/bench_2rank.err | grep "^{" > $O/bench_2rank.json
python bench.py --width 1080 --height 1080 --no-cpu-baseline > $O/bench_square.json 2> /dev/null
python bench.py --numerics fast --no-cpu-baseline > $O/bench_fast.json 2> /dev/null
timeout 900 python bench.py --stream-mode reference --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_reference.json 2> /dev/null
python bench.py --pipeline wavefront --no-cpu-baseline > $O/bench_wavefront.json 2> /dev/null
python bench.py --scene living_room --steps 2 --warmup 1 > $O/bench_living.json 2> /dev/null
python bench.py --scene living_room --steps 2 --warmup 1 --no-cpu-baseline --numerics fast > $O/bench_living_fast.json 2> /dev/null
python bench.py --scene cbox_medium --steps 2 --warmup 1 > $O/bench_medium.json 2> /dev/null
python bench.py --scene cbox_medium --steps 2 --warmup 1 --no-cpu-baseline --numerics fast > $O/bench_medium_fast.json 2> /dev/null
python scratch/fast_parity.py > $O/fast_parity.json 2> $O/fast_parity.err
python bench.py --scene living_room --tris 4000000 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_living4m.json 2> /dev/null
mkdir -p scratch/variants; rm -f scratch/variants/*.so; cp rustlight_amd/lib/librustlight_amd.so scratch/variants/libdefault.so
python scratch/mc_bench.py > $O/mc_bench.log 2>&1; rm -f scratch/variants/libdefault.so
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cbox -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/stats_cbox.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_living -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --scene living_room > $O/stats_living.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_medium -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --scene cbox_medium > $O/stats_medium.log 2>&1
cd $R
bash scratch/pmc_collect.sh ${TAG:-r2z}/pmc_cbox k_path_fused > $O/pmc_cbox.log 2>&1
bash scratch/pmc_collect.sh ${TAG:-r2z}/pmc_living k_path_fused --scene living_room > $O/pmc_living.log 2>&1
bash scratch/pmc_collect.sh ${TAG:-r2z}/pmc_living4m k_path_fused --scene living_room --tris 4000000 > $O/pmc_living4m.log 2>&1
cat $O/mc_bench.log
for f in bench_default bench_2rank bench_square bench_fast bench_reference bench_wavefront bench_living bench_living_fast bench_living4m bench_medium bench_medium_fast; do grep "^{" $O/$f.json | python -c "import sys,json; o=json.loads(sys.stdin.read()); print('$f', round(o['value'],1), round(o['ms_per_step'],2), round(o['roofline']['avg_launch_ms'],2), o['n_gpus'], o['distributed']['crc_match'])"; done
find $O -name '*kernel_trace.csv' -size +2M -delete; find $O -name '*agent_info.csv' -delete; find $O -name '*domain_stats.csv' -delete
