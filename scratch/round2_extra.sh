# round 2, extra evidence: kernel-trace stats of the ao / direct integrators (k_pixel_mc) and of the wavefront pipeline on the 508 k-triangle scene
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r2extra; mkdir -p $O
mkdir -p scratch/variants; rm -f scratch/variants/*.so; cp rustlight_amd/lib/librustlight_amd.so scratch/variants/libdefault.so
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_mc -o p -- python $R/scratch/mc_bench.py > $O/stats_mc.log 2>&1
if [ -z "$SKIP_WF" ]; then rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_wf_living -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --scene living_room --pipeline wavefront > $O/stats_wf_living.log 2>&1; fi
cd $R; rm -f scratch/variants/libdefault.so
if [ -z "$SKIP_WF" ]; then python bench.py --steps 2 --warmup 1 --no-cpu-baseline --scene living_room --pipeline wavefront > $O/bench_living_wavefront.json 2>/dev/null; fi
grep "^{" $O/bench_living_wavefront.json | python -c "import sys,json; o=json.loads(sys.stdin.read()); print('living wavefront', round(o['value'],1), round(o['ms_per_step'],2))"
head -8 $O/stats_mc/p_kernel_stats.csv | cut -c1-160; head -8 $O/stats_wf_living/p_kernel_stats.csv | cut -c1-160
find $O -name '*kernel_trace.csv' -size +2M -delete; find $O -name '*agent_info.csv' -delete; find $O -name '*domain_stats.csv' -delete
