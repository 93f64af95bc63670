import re,collections,sys
rows=collections.OrderedDict()
for l in open(sys.argv[1]):
    m=re.search(r'buf(\d+) \[(.*?)\s*\] \S+\s+([\d.]+)',l)
    if m: rows.setdefault(m.group(2),{})[int(m.group(1))]=m.group(3)
for k,v in rows.items(): print(f'{k:40s}', '  '.join(v[b] for b in sorted(v)))
