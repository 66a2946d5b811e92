import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
recs = bench.secondary_records(0)
for r in recs:
    if 'improve(ADMM' in r.get('config', '') and 'configs[1]' in r.get('config', ''):
        print(json.dumps(r, indent=1, default=str))
