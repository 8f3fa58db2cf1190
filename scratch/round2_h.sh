R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r2h; mkdir -p $O
python scratch/variants.py run cbox 2 128 > $O/variants_cbox.txt 2>&1
python scratch/variants.py run living_room 2 32 > $O/variants_living.txt 2>&1
SPLITS=1,4,16,64 python scratch/variants.py one scratch/variants/libcur.so living_room 2 32 > $O/splits_living.txt 2>&1
cat $O/variants_cbox.txt $O/variants_living.txt $O/splits_living.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -4 $O/pytest.log
