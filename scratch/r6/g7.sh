# round 6, GPU call 7: smoke, the whole -m gpu suite, the round's evidence (scratch/r6/profile_round6.sh)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6g7; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
bash scratch/r6/profile_round6.sh > $O/profile.log 2>&1; tail -c 1500 $O/profile.log
