R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r2g; mkdir -p $O
python scratch/variants.py run cbox 2 128 > $O/variants_cbox.txt 2>&1
python scratch/variants.py run cbox_medium 2 32 > $O/variants_medium.txt 2>&1
python scratch/variants.py run living_room 2 32 > $O/variants_living.txt 2>&1
NUMERICS=1 python scratch/variants.py run cbox 2 128 > $O/variants_cbox_fast.txt 2>&1
cat $O/variants_*.txt
