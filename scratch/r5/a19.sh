cd $GRAFT_REPO_ROOT
for i in 1 2; do python scratch/variants.py run cbox 2 128; done
python scratch/variants.py run cbox_medium 2 32
python scratch/variants.py run living_room 2 32
