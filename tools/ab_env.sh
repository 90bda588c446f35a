# A/B of one environment switch on one box, alternating: bash tools/ab_env.sh VAR [pytest -k expression]
V=$1; K=${2:-"c2_mul or test_multiply"}
mkdir -p gpurun_out/r02e
for round in 1 2 3; do
for on in 0 1; do
  if [ $on = 1 ]; then export $V=1; else unset $V; fi
  echo "== $V=${on} (round $round)"
  if [ $round = 1 ]; then python -m pytest tests/test_gpu_parity.py -x -q -k "$K" 2>&1 | tail -1; fi
  python bench.py --no-cpu --no-extras --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], {k:v['ms'] for k,v in d['roofline']['kernels'].items()})"
done
done 2>&1 | tee gpurun_out/r02e/ab_env_$V.txt
