for v in 1 256 0 1 256 0; do echo -n "MODA_BWD256=$v: "; MODA_BWD256=$v timeout 400 python bench.py --mode train --precision bf16 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('eager_ms_per_step'), d['loss'])"; done
