"""Dev tool: instruction histogram of the loops of plan_step_kernel in a hipcc -S listing (block comments name the loop)."""
import re, sys, collections
lines = open(sys.argv[1]).read().split('\n')
start = [i for i, l in enumerate(lines) if l.startswith('_ZN10lipmpc_dev16plan_step_kernel')][0]
end = [i for i, l in enumerate(lines) if 's_endpgm' in l and i > start][0]
cur = None
loops = collections.defaultdict(list)
for l in lines[start:end]:
    m = re.match(r'\.LBB\d+_(\d+):\s*;\s*(.*)', l)
    if m:
        c = m.group(2)
        h = re.search(r'Header=BB\d+_(\d+) Depth=1', c)
        if 'Loop Header: Depth=1' in c:
            cur = m.group(1)
        elif h:
            cur = h.group(1)
        elif 'Parent Loop BB' in c:
            cur = re.search(r'Parent Loop BB\d+_(\d+)', c).group(1)
        else:
            cur = None
        continue
    if re.match(r'\.LBB', l):
        cur = None if 'Loop' not in l else cur
        continue
    if cur and re.match(r'\s+[a-z]', l) and not l.strip().startswith('.'):
        loops[cur].append(re.sub(r'_e32$|_e64$', '', l.split()[0]))
for k, v in loops.items():
    if len(v) > 300:
        print('loop BB_%s: %d instrs' % (k, len(v)), collections.Counter(v).most_common(int(sys.argv[2]) if len(sys.argv) > 2 else 14))
