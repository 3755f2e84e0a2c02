import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{"metric"')][-1])
print(sys.argv[1], "main", d["value"], [(o["config"]["workload"][:14], o["value"]) for o in d.get("other_configs") or []], (d.get("config5_batch") or {}).get("value"))
