"""Per-shape averages of the 256-row GEMM launches in OVO_PROF_DUMP files (one line per profiled launch: kind M N K work ms flags), for A/B runs of the bench:
    OVO_PROF_DUMP=a.txt python bench.py ...; OVO_VIT_LNFOLD=0 OVO_PROF_DUMP=b.txt python bench.py ...; python tools/prof_dump_shapes.py a.txt b.txt"""
import sys, collections
for f in sys.argv[1:]:
    agg = collections.defaultdict(lambda: [0, 0.0])
    for line in open(f):
        k, a, b, c, wk, t, fl = line.split()
        if int(k) in (0, 3) and int(a) >= 2048:
            e = agg[(int(k), int(a), int(b), int(c), int(fl))]; e[0] += 1; e[1] += float(t)
    print(f)
    for key, (n, t) in sorted(agg.items(), key=lambda x: -x[1][1])[:14]:
        print("   kind %d M %6d N %5d K %5d flags %4d: %5d launches, avg %8.2f us" % (*key, n, 1e3 * t / n))
