import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], d["roofline"]["traffic_source"][:40])
print("cpu", d["cpu_baseline"]["value"])
for k,v in d.get("extra",{}).items():
    if isinstance(v,dict): print(k, v.get("value"), v.get("error"), (v.get("host_fed_pipeline") or {}).get("sites_per_s"), (v.get("pool_form") or {}).get("value"))
