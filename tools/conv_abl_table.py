"""tools/conv_ab.sh output -> one row per shape, one column per library variant (best TF/s of the passes)."""
import collections, sys
d = collections.defaultdict(list)
order = []
for ln in open(sys.argv[1]):
    p = ln.split()
    if len(p) < 11:
        continue
    if p[0] not in order:
        order.append(p[0])
    d[(p[0], " ".join(p[2:7]))].append(float(p[9]))
shapes = []
for (v, s) in d:
    if s not in shapes:
        shapes.append(s)
print("shape".ljust(22), " ".join(v.rjust(9) for v in order))
for s in shapes:
    print(s.ljust(22), " ".join(("%.0f" % max(d[(v, s)])).rjust(9) if (v, s) in d else "-".rjust(9) for v in order))
