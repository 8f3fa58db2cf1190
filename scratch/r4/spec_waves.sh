# dev: per-wave lifetimes of k_stream_spec on the current tree (timers build on the box)
R=$GRAFT_REPO_ROOT; cd $R
RL_HIP_FLAGS="-DRL_SPEC_TIMERS" python -m rustlight_amd.build --force > /dev/null 2>&1
python scratch/r4/wave_top.py 2>&1 | tail -45
