# round 6, GPU call 9 (final evidence on the final sources): quick check of the Cornell box in reference-order streams, smoke, the whole -m gpu suite, scratch/r6/profile_round6.sh
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6g9; mkdir -p $O
for rep in 1 2; do REPS=4 timeout 300 python scratch/ref_bench.py cbox 128 2>&1 | tail -1 | cut -c1-175; done | tee $O/cbox_check.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; grep -n "passed\|failed" $O/pytest_gpu.log | tail -2
bash scratch/r6/profile_round6.sh > $O/profile.log 2>&1; tail -c 300 $O/profile.log
