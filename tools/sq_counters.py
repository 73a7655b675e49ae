"""Average SQ counters per kernel from a rocprofv3 --pmc run directory (counter_collection.csv) -- e.g. the VALU / wait breakdown of the
geometry passes at 10 M points (profiles/r02*_geom_10m_sq_counters.txt).  usage: python tools/sq_counters.py DIR"""
import collections, csv, glob, os, sys
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        rows[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("average per launch; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count quad-cycles summed over waves (MI355X_MICROARCH.md)")
for k, d in sorted(rows.items()):
    name = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if not name.startswith("k_"):
        continue
    print("%-18s launches %3d  " % (name, len(next(iter(d.values())))) + "  ".join("%s %d" % (c.replace("SQ_", ""), round(sum(v) / len(v))) for c, v in sorted(d.items())))
