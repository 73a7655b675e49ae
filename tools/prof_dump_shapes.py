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
