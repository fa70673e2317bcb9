import csv, sys, collections, glob, os
root = sys.argv[1]
def load(sub):
    f = glob.glob(os.path.join(root, sub, "*", "*counter_collection.csv"))
    if not f: return {}
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].replace("void (anonymous namespace)::","").replace("(anonymous namespace)::","").split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key=(r["Dispatch_Id"])
        if (k,key) not in seen:
            seen.add((k,key)); cnt[k]+=1
    return agg, cnt
out = {}
for sub in ("sq","sq2","fetch","write"):
    res = load(sub)
    if not res: continue
    agg, cnt = res
    for k,v in agg.items():
        out.setdefault(k, {"n": cnt[k]}).update(v)
keys = sorted(out, key=lambda k: -out[k].get("SQ_WAVE_CYCLES",0))
for k in keys[:14]:
    v = out[k]; n = max(v["n"],1)
    print(k[:60], "n=%d" % n)
    for c in sorted(v):
        if c!="n": print("   %-28s %.4g per launch" % (c, v[c]/n))
