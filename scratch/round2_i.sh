R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r2i; mkdir -p $O
python -c "
from rustlight_amd import api, scenes
print('cbox', api.Context(api.Scene(scenes.cbox(64,64)),0).debug_sizes())
"
for v in w4 w5; do python scratch/variants.py one scratch/variants/lib$v.so cbox 2 128; python scratch/variants.py one scratch/variants/lib$v.so cbox_medium 2 32; done 2>&1 | tee $O/variants_cbox.txt
for v in w4 s4 s5 s8; do python scratch/variants.py one scratch/variants/lib$v.so living_room 2 32; done 2>&1 | tee $O/variants_living.txt
