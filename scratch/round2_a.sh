# round 2, GPU call A: VALU calibration (+ its PMC passes), the -m gpu suite, bench lines (default, 2 ranks on one GPU, reference-order, square frame)
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r2a; mkdir -p $O
$R/scratch/bin/valu_calib 20000 > $O/valu_calib.jsonl 2> $O/valu_calib.err
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/calib_pmc -o p -- $R/scratch/bin/valu_calib 20000 0 1 2 7 14 25 > $O/calib_pmc.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $O/calib_grbm -o p -- $R/scratch/bin/valu_calib 20000 0 1 2 7 14 25 > $O/calib_grbm.log 2>&1
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --gpus 2 --steps 2 --warmup 1 > $O/bench_2rank.json 2> $O/bench_2rank.err
python bench.py --width 1080 --height 1080 --no-cpu-baseline > $O/bench_square.json 2> $O/bench_square.err
timeout 600 python bench.py --stream-mode reference --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_reference.json 2> $O/bench_reference.err
tail -3 $O/pytest.log; head -c 600 $O/bench_default.json; echo; head -c 400 $O/bench_2rank.json; echo; tail -5 $O/bench_2rank.err
