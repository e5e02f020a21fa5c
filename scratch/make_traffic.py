"""Builds profiles/<tag>_traffic.json from two rocprofv3 --pmc passes (rocpd databases) of bench.py:
   make_traffic.py <tag> <config> <batch> <fetch.db> <write.db>
HBM bytes per launch and kernel = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: rocprofv3 reports KiB, and on gfx950
FETCH_SIZE counts half of the bytes of 16-byte-per-lane streaming reads (MI355X_MICROARCH.md, HBM)."""
import hashlib, json, os, re, sqlite3, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
def short(n):
    m = re.match(r'(?:void )?(svk::\w+(?:<[^>]*>)?)', n)
    return m.group(1) if m else n[:48]
def means(db, counter):
    c = sqlite3.connect(db)
    q = "select kernel_name,count(*),avg(value) from counters_collection where counter_name=? group by kernel_name"
    return {short(n): (k, v) for n, k, v in c.execute(q, (counter,)) if "svk::" in n}
tag, config, batch, fdb, wdb = sys.argv[1:6]
import bench
f, w = means(fdb, "FETCH_SIZE"), means(wdb, "WRITE_SIZE")
kern = {}
for k in sorted(set(f) | set(w)):
    fk, wk = f.get(k, (0, 0.0)), w.get(k, (0, 0.0))
    kern[k] = {"dispatches_fetch_pass": fk[0], "dispatches_write_pass": wk[0], "fetch_size_kib": round(fk[1], 1),
               "write_size_kib": round(wk[1], 1), "traffic_bytes_per_launch": int(round((2 * fk[1] + wk[1]) * 1024))}
path = os.path.join(ROOT, "profiles", "%s_traffic.json" % tag)
doc = {}
if os.path.exists(path):
    doc = json.load(open(path))
doc["_comment"] = ("HBM traffic per launch and kernel from rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (two separate passes "
                   "of `python bench.py --no-cpu-baseline`), traffic = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 correction, "
                   "MI355X_MICROARCH.md HBM section). source_sha = sha256[:16] over stereo-vision_amd/csrc/{%s}: bench.py "
                   "reports these figures only when the tree's sources hash to it." % ",".join(bench.KERNEL_SOURCES))
doc["source_sha"] = bench.source_sha()
doc["%s_b%s" % (config, batch)] = {"kernels": kern}
json.dump(doc, open(path, "w"), indent=1)
print(json.dumps(doc["%s_b%s" % (config, batch)], indent=1))
