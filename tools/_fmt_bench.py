import json
import sys

for line in sys.stdin:
    line = line.rstrip()
    if line.startswith("{"):
        d = json.loads(line)
        r = d.get("roofline") or {}
        print("value", d["value"], "ms/step", d["ms_per_step"], "| kernel_ms", r.get("kernel_ms_mean"), "GB/s", r.get("achieved"), "frac", r.get("frac"))
        for k, v in (d.get("extras") or {}).items():
            print("  ", k, v)
        if d.get("cpu_baseline"):
            print("  cpu_baseline", d["cpu_baseline"])
    else:
        print(line)
