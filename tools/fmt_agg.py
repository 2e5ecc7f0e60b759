import sys,json
for l in sys.stdin:
    if not l.startswith("{"): continue
    r=json.loads(l); print(r["shape"], r["flow"], {k:v for k,v in r.items() if k.endswith("_us") or k.endswith("maxdiff")})
