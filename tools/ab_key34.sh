for k in 7 15; do
python3 bench.py --shard-of 8 --workload configs2 --steps 1 --warmup 1 --gen 64 --no-cpu-baseline --no-side --no-fp8 --tuning 34=$k 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d.get('configs2',d)
print('key34=$k decode ms/step', c.get('decode_ms_per_step_p50'), d.get('decode_ms_per_token_p50'))"
done
