# round 3: root treelet in LDS for the streaming kernel: sweep of treelet size x LDS stack levels (508 k-triangle scene, 32 spp)
O=gpurun_out/r3e; mkdir -p $O
L=rustlight_amd/lib/librustlight_amd.so
for lv in 6 4 2; do for t in 0 32 64 128 256 512; do
  echo -n "levels $lv top $t: "; RL_LDS_LEVELS=$lv RL_TOP_NODES=$t python scratch/variants.py one $L living_room 2 32 | tail -1
done; done 2>&1 | tee $O/sweep.log
