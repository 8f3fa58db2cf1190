cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
for i in $(seq 1 40); do timeout 60 python scratch/r5/stall_repro.py own 2>&1 | grep -v amdgpu.ids | tail -3; done > gpurun_out/r5/stall_final_own.txt
for i in $(seq 1 40); do timeout 60 python scratch/r5/stall_repro.py torch 2>&1 | grep -v amdgpu.ids | tail -3; done > gpurun_out/r5/stall_final_torch.txt
for f in own torch; do echo "$f: $(grep -c STALL gpurun_out/r5/stall_final_$f.txt) long renders in $(grep -c worst gpurun_out/r5/stall_final_$f.txt) processes; worst-render distribution:"; grep worst gpurun_out/r5/stall_final_$f.txt | awk '{print int($4/20)*20}' | sort -n | uniq -c | tr '\n' ';'; echo; done
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
