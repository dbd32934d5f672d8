for c in 32 256 1024 4096 256 32; do
  EXA_COST_CLASSES=$c python bench.py --steps 20 --cpu-baseline off --pmc off > gpurun_out/r03_x_cls$c.json 2>/dev/null
  python -c "
import json; d=json.loads(open('gpurun_out/r03_x_cls$c.json').read().strip().splitlines()[-1]); print('classes $c: %.3f ms/frame kernel %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
done
