# dev: kernel trace of two reference-order frames in flight — do the chain pass of one frame and the passes of the other really overlap?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/inflight_trace; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/t -o p -- python $R/scratch/r4/inflight.py reference 6 1 cbox 2 > $O/run.log 2>&1
grep "in flight" $O/run.log
python - <<PY
import csv, glob, json
rows = []
for f in glob.glob('$O/t/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'k_stream_spec' in n or 'k_path_fused' in n:
            rows.append(('k_stream_spec' if 'k_stream_spec' in n else 'k_path_fused', int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Stream_Id') or r.get('Queue_Id')))
rows.sort(key=lambda r: r[1])
rows = rows[-24:]           # the timed part: 6 frames x 2 kernels (after the warm-up frames), with slack
t0 = rows[0][1]
ev = [{'kernel': k, 'start_ms': round((a - t0) * 1e-6, 2), 'end_ms': round((b - t0) * 1e-6, 2), 'queue': q} for k, a, b, q in rows]
# union of the busy intervals and sum of the durations
iv = sorted((a, b) for _, a, b, _ in rows)
union, cs, ce = 0, iv[0][0], iv[0][1]
for a, b in iv[1:]:
    if a > ce: union += ce - cs; cs, ce = a, b
    else: ce = max(ce, b)
union += ce - cs
total = sum(b - a for _, a, b, _ in rows)
out = {'what': 'rocprofv3 --kernel-trace of scratch/r4/inflight.py reference 6 1 cbox 2 (two contexts, two host threads): the last 24 launches of k_stream_spec / k_path_fused',
       'sum_of_kernel_durations_ms': round(total * 1e-6, 1), 'time_with_a_kernel_running_ms': round(union * 1e-6, 1), 'overlap_factor': round(total / union, 3), 'launches': ev}
json.dump(out, open('$O/summary.json', 'w'), indent=1)
print({k: out[k] for k in ('sum_of_kernel_durations_ms', 'time_with_a_kernel_running_ms', 'overlap_factor')})
for e in ev[:10]: print(e)
PY
find $O -name '*.csv' -size +1M -delete
