import re,collections,sys
rows=collections.OrderedDict()
for l in open(sys.argv[1]):
    m=re.search(r'buf(\d+) \[(.*?)\s*\] old\s+([\d.na]+)\s+product\s+([\d.]+).*bits: (\w+)',l)
    if m: rows.setdefault(m.group(2),{})[int(m.group(1))]=m.group(3)+'/'+m.group(4)+('' if m.group(5)=='True' else '!')
for k,v in rows.items(): print(f'{k:28s}', '  '.join(v[b] for b in sorted(v)))
