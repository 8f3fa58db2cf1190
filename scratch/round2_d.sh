R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r2d; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L > $O/counters_list.txt 2>&1
grep -oE "(TA|TCP|TD|TCC|SQ|SQC|GRBM|SPI|CPC|TCA)_[A-Za-z0-9_]+" $O/counters_list.txt | sort -u > $O/counter_names.txt
wc -l $O/counter_names.txt
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
python bench.py --no-cpu-baseline --numerics fast > $O/bench_fast.json 2> $O/bench_fast.err
timeout 600 python bench.py --stream-mode reference --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_reference.json 2> $O/bench_reference.err
for s in 0 1 2 3 4 5 6; do RL_ITEM_SHIFT=$s timeout 600 python bench.py --stream-mode reference --steps 1 --warmup 0 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read()); print('item_shift $s', o['value'], o['ms_per_step'])"; done > $O/item_shift.txt 2>&1
python bench.py --scene living_room --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_living.json 2> $O/bench_living.err
python bench.py --scene living_room --steps 2 --warmup 1 --no-cpu-baseline --numerics fast > $O/bench_living_fast.json 2> $O/bench_living_fast.err
cat $O/item_shift.txt
for f in bench_default bench_fast bench_reference bench_living bench_living_fast; do grep "^{" $O/$f.json | python -c "import sys,json; o=json.loads(sys.stdin.read()); print('$f', o['value'], o['ms_per_step'], o['roofline']['avg_launch_ms'], o['config']['image_mean'])"; done
