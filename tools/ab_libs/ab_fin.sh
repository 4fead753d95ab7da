for i in 1 2 3; do
  for v in old new; do
    cp tools/ab_libs/$v.so margipose_amd/libmargipose_hip.so
    python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-inference --no-configs4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v', round(d['ms_per_step'],3))"
  done
done
cp tools/ab_libs/new.so margipose_amd/libmargipose_hip.so
