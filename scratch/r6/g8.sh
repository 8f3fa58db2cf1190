# round 6, GPU call 8: (1) where do the waves of k_stream_chain run (HW_ID / XCC_ID per wave) at 4 / 8 / 2 chains per wave, overlap on and off;
# (2) same-box regression check of the Cornell box in reference-order streams: the round-5 tree (3dddd68) against this one
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6g8; mkdir -p $O
{
T=scratch/variants/libtimers.so
for k in 4 5 3; do for ov in 0 1; do
  echo "-- timers build, cbox + medium, 128 spp, RL_ITEM_SHIFT=$k RL_NO_OVERLAP=$ov"
  if [ $ov = 1 ]; then export RL_NO_OVERLAP=1; else unset RL_NO_OVERLAP; fi
  RL_ITEM_SHIFT=$k RL_CHAIN_WAVE_TIMES=$O/chain_waves_shift${k}_noov$ov.txt REPS=1 timeout 600 python scratch/ref_bench.py cbox_medium 128 $T 2>&1 | grep -v "^\[stage\]" | tail -5 | cut -c1-230
done; done
unset RL_NO_OVERLAP
echo "== shipped library, cbox + medium, overlap on / off"
for rep in 1 2; do
REPS=2 timeout 300 python scratch/ref_bench.py cbox_medium 128 2>&1 | tail -1 | cut -c1-175
RL_NO_OVERLAP=1 REPS=2 timeout 300 python scratch/ref_bench.py cbox_medium 128 2>&1 | tail -1 | cut -c1-175
done
echo "== Cornell box, reference-order, 1080p x 128 spp: round-5 tree vs this tree, interleaved"
for rep in 1 2 3; do
(cd scratch/r5tree && REPS=4 timeout 300 python scratch/ref_bench.py cbox 128 2>&1 | tail -1 | cut -c1-175 | sed 's/^/r5  /')
REPS=4 timeout 300 python scratch/ref_bench.py cbox 128 2>&1 | tail -1 | cut -c1-175 | sed 's/^/now /'
done
(cd scratch/r5tree && REPS=2 timeout 300 python scratch/ref_bench.py cbox_medium 128 2>&1 | tail -1 | cut -c1-175 | sed 's/^/r5  /')
(cd scratch/r5tree && REPS=2 timeout 300 python scratch/ref_bench.py living_room 128 2>&1 | tail -1 | cut -c1-175 | sed 's/^/r5  /')
REPS=2 timeout 300 python scratch/ref_bench.py living_room 128 2>&1 | tail -1 | cut -c1-175 | sed 's/^/now /'
} > $O/log.txt 2>&1
cat $O/log.txt
