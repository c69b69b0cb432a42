cd $GRAFT_REPO_ROOT
for b in 32 64 96 128 192 256; do for v in 0 2; do
python bench.py --variant resize_lds=$v --steps 10 --warmup 2 --no-cpu-baseline --no-end-to-end --no-density-sweep --no-extra-configs --batch $b > /tmp/o.json 2>/dev/null
python -c "
import json; d=json.load(open('/tmp/o.json')); print('batch $b lds=$v', d['value'], d['ms_per_step'], d['stages_ms']['resize'])"
done; done
