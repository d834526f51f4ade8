for i in 1 2 3; do
  for lib in prev new; do
    if [ $lib = prev ]; then export MF_LIB_PATH=$PWD/markovflow_amd/libmf_prev.so; else unset MF_LIB_PATH; fi
    python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('$lib', 'ms/step', round(j['ms_per_step'],4), 'min', round(j['ms_per_step_min'],4), 'median', round(j['ms_per_step_median'],4), 'kernel', round(j['roofline']['kernel_ms'],4), 'll', j['log_likelihood'])"
  done
done
