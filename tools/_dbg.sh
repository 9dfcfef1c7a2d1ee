python -m pytest tests -m gpu -q -x 2>&1 | tail -3
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --also-f32 0 --dice 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline'])"
