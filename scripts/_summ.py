import sys, json
bad = 0; tot = 0
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()[:200]); continue
    tot += 1; bad += len(d.get("events", []))
print("processes", tot, "events", bad)
