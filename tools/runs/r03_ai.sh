for cfg in "8 64" "32 16" "128 4" "100000 1"; do set -- $cfg
  EXA_PREPASS_SPLIT_DIV=$1 EXA_PREPASS_SPLIT_MIN=$2 EXA_HIP_VERBOSE=1 python bench.py --size 4096 --iso 0.5 --ao --spp 16 --steps 3 --warmup 1 --cpu-baseline off --pmc off > gpurun_out/r03_ai_c5_$1.json 2> gpurun_out/r03_ai_c5_$1.err
  echo "C5 div $1 min $2: $(python -c "import json; d=json.loads(open('gpurun_out/r03_ai_c5_$1.json').read().strip().splitlines()[-1]); print('%.1f ms/step' % d['ms_per_step'])") $(grep -h 'pre-pass costs' gpurun_out/r03_ai_c5_$1.err | tail -1)"
  EXA_PREPASS_SPLIT_DIV=$1 EXA_PREPASS_SPLIT_MIN=$2 EXA_HIP_VERBOSE=1 python bench.py --config c3_gear --iso 0.5 --steps 20 --cpu-baseline off --pmc off > gpurun_out/r03_ai_c3_$1.json 2> gpurun_out/r03_ai_c3_$1.err
  echo "C3 div $1 min $2: $(python -c "import json; d=json.loads(open('gpurun_out/r03_ai_c3_$1.json').read().strip().splitlines()[-1]); print('%.3f ms/step' % d['ms_per_step'])") $(grep -h 'pre-pass costs' gpurun_out/r03_ai_c3_$1.err | tail -1)"
done
