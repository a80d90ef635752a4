for n in 1024 4096 16384; do for k in rand text; do
python3 bench.py --mode inflate --kind $k --streams $n --steps 3 --warmup 1 --no-extra --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$k $n', d['value'], d['ms_per_step'])"
done; done
