for w in 0 150 0 150; do
  python bench.py --clock-warmup-ms $w --no-cpu-baseline --no-c5 --no-bassoc --no-batch 2>/dev/null > /tmp/l.json
  python -c "
import json; d=json.loads(open('/tmp/l.json').readline()); print('clock warmup $w', d['value'], d['ms_per_step'], d['steps'], d['warmup'])"
done
