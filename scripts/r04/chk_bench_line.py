import json
d=json.loads(open("/tmp/b.json").read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["pmc_stale"], d["cpu_baseline"]["value"])
