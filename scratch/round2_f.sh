R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r2f; mkdir -p $O
timeout 600 python scratch/variants.py run living_room 2 32 > $O/variants_living.txt 2>&1
cat $O/variants_living.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mixed or full_size or medium or fast or degenerate or hostile or randomized or light_tree" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -5 $O/pytest.log
