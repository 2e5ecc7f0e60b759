import sys, json
for line in sys.stdin:
    line=line.strip()
    if not line.startswith("{"): continue
    d=json.loads(line)
    if any(k in d["case"] for k in sys.argv[1:]): print("   %-75s %8.1f us" % (d["case"][:75], d["us"]))
