"""Per-basic-block instruction mix of a kernel in the saved gfx950 assembly (build/sar_hip/*.s).
usage: python tools/asm_blocks.py <mangled-name-prefix> [min_block_size]"""
import re, sys
import glob
s = ''.join(open(f).read() for f in sorted(glob.glob('build/sar_hip/sar_*-hip-amdgcn-amd-amdhsa-gfx950.s')))
pref = sys.argv[1]; mn = int(sys.argv[2]) if len(sys.argv) > 2 else 15
m = re.search(r'^(%s[A-Za-z0-9_]*):' % re.escape(pref), s, re.M)
a = m.end(); b = s.index('.Lfunc_end', a)
blocks = []; cur = ['entry', []]
for l in s[a:b].split('\n'):
    l = l.strip()
    if re.match(r'\.LBB\d+_\d+:', l):
        blocks.append(cur); cur = [l.split(':')[0], []]; continue
    if not l or l.startswith(';') or l.startswith('.'): continue
    cur[1].append(l)
blocks.append(cur)
print(m.group(1), len(blocks), 'blocks')
for lab, ins in blocks:
    if len(ins) < mn: continue
    c = dict(valu=0, f64=0, salu=0, ds=0, glob=0, wait=0, br=0)
    for i in ins:
        op = i.split()[0]
        if op.startswith('v_'):
            c['valu'] += 1; c['f64'] += 'f64' in op
        elif op.startswith('s_waitcnt'): c['wait'] += 1
        elif op.startswith(('s_cbranch', 's_branch')): c['br'] += 1
        elif op.startswith('s_'): c['salu'] += 1
        elif op.startswith('ds_'): c['ds'] += 1
        elif op.startswith(('global_', 'buffer_', 'flat_')): c['glob'] += 1
    print('  ', lab, len(ins), c)
