python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for sc in cbox living_room cbox_medium; do python bench.py --steps 3 --scene $sc --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['image_mean'])"; done
