for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --no-side-legs --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('K20',d['value'],d['ms_per_step']*20,d['roofline']['kernel_ms'])"; done
python bench.py --no-side-legs --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('K100',d['value'],d['ms_per_step']*100,d['roofline']['kernel_ms'])"
