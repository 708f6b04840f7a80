import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("cpu_baseline",{}).get("value"), d.get("strict",{}).get("value"), d.get("fp32_fast_mode",{}).get("value"), d["config"].get("mapping"))
