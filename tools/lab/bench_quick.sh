python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('s20', d['value'], d['iterations']['first_ms'], d['iterations']['later_ms_per_step'], d['roofline']['avg_launch_ms'])"
python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('s6 ', d['value'], d['iterations']['first_ms'], d['iterations']['later_ms_per_step'], d['roofline']['avg_launch_ms'])"
