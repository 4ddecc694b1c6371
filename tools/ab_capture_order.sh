for ord in "7,5;7,inf" "4,5;7,inf" "2,5;7,inf" "1,2;7,inf" "3,2;4,3;7,inf" "5,5;7,inf" "7,3;7,inf" "9,5;7,inf"; do
  NR_CAPTURE_ORDER="$ord" python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$ord', d['value'], d['ms_per_step'])"
done
